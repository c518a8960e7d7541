"""One-off soak of the big-list kernel (lva_step_big<LL,P>) against the CPU oracle: P = 2..4, L = 9..64."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import nanopore_dna_storage_amd as pkg
from nanopore_dna_storage_amd import synth
from oracle import oracle as O
rng = np.random.default_rng(int(sys.argv[1])); ncase = int(sys.argv[2])
bad = n = 0; t0 = time.time()
for it in range(ncase):
    m = int(rng.choice([6, 8, 8])); r = int(rng.choice([1, 3, 5] if m == 8 else [1, 2]))
    msg_len = int(rng.choice([100, 150, 180, 200, 240]))
    try:
        pkg.code_info(m, r, msg_len)
    except pkg.LvaError:
        continue
    L = int(rng.choice([9, 12, 16, 24, 32, 36, 44, 48, 52, 60, 64]))   # (from 32 on, multiples of 4 with three planes: the record layout)
    md = int(rng.choice([20, 10])); margin = float(rng.choice([2.5, 3.0, 4.0]))
    seed = int(rng.integers(1 << 30))
    reads = [synth.make_read(m, r, msg_len, seed + i, rc=bool(i & 1), margin=margin) for i in range(2)]
    with pkg.Decoder(m, r, msg_len, list_size=L, max_deviation=md, max_slots=2) as dec:
        got = dec.decode([x["post"] for x in reads], rc=[x["rc"] for x in reads])
        pr = dec.profile()
    for x, g in zip(reads, got):
        wm, ws = O.OracleCode(m, r, msg_len, rc=x["rc"]).decode(x["post"], L, md, num_threads=32)
        ok = np.array_equal(g[0], wm) and np.array_equal(g[1].view(np.uint32), ws.view(np.uint32))
        n += 1
        if not ok:
            bad += 1; print("MISMATCH", m, r, msg_len, L, md, margin, seed)
    print(it, (m, r, msg_len, L, md, margin), "P", (msg_len + m + 63) // 64, "kernel", pr["kernel"], "fixups", pr["fixup_reason"], "t=%.0fs" % (time.time() - t0), flush=True)
print("checked", n, "bad", bad)
