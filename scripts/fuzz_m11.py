"""One-off soak: m=11 / m=14 random configurations on the GPU against the CPU oracle (minutes of CPU)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import nanopore_dna_storage_amd as pkg
from nanopore_dna_storage_amd import synth
from oracle import oracle as O
rng = np.random.default_rng(int(sys.argv[1])); ncase = int(sys.argv[2])
cfgs = [(11, 1, 40), (11, 2, 61), (11, 5, 100), (11, 5, 180), (11, 1, 90), (14, 1, 20), (14, 7, 58)]
bad = n = 0; t0 = time.time()
for it in range(ncase):
    m, r, msg_len = cfgs[rng.integers(len(cfgs))]
    try:
        pkg.code_info(m, r, msg_len)
    except pkg.LvaError:
        continue
    L = int(rng.choice([1, 2, 4, 8, 8])); md = int(rng.choice([20, 20, 10, 5])); margin = float(rng.choice([2.5, 3.0, 4.0, 6.0]))
    seed = int(rng.integers(1 << 30))
    reads = [synth.make_read(m, r, msg_len, seed + i, rc=bool(i & 1), margin=margin) for i in range(2)]
    with pkg.Decoder(m, r, msg_len, list_size=L, max_deviation=md, max_slots=2) as dec:
        got = dec.decode([x["post"] for x in reads], rc=[x["rc"] for x in reads])
        fx = dec.profile()["fixup_reason"]
    for x, g in zip(reads, got):
        wm, ws = O.OracleCode(m, r, msg_len, rc=x["rc"]).decode(x["post"], L, md, num_threads=32)
        ok = np.array_equal(g[0], wm) and np.array_equal(g[1].view(np.uint32), ws.view(np.uint32))
        n += 1
        if not ok:
            bad += 1; print("MISMATCH", m, r, msg_len, L, md, margin, seed)
    print(it, (m, r, msg_len, L, md, margin), "fixups", fx, "t=%.0fs" % (time.time() - t0), flush=True)
print("checked", n, "bad", bad)
