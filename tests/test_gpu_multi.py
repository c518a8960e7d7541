"""Multi-rank path on the GPU box: rank processes started by sharding.launch_ranks /
bench.py itself, the REAL decoder in every rank, lists gathered on rank 0 and compared with a
single-process decode.  On a 1-GPU box the ranks share GPU 0 (gloo); with RCCL each rank needs its own GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import nanopore_dna_storage_amd as pkg
from nanopore_dna_storage_amd import sharding, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_two_ranks_real_decoder_equal_single_process(tmp_path):
    m, r, ml, L, n = 6, 1, 60, 4, 9
    out = str(tmp_path / "g.npz")
    env = dict(os.environ, LVA_DIST_BACKEND="gloo")
    rc_ = sharding.launch_ranks(os.path.join(HERE, "_rank_worker.py"), [out, "gpu", str(n), str(m), str(r), str(ml), str(L)], 2, env=env)
    assert rc_ == 0
    z = np.load(out)
    assert int(z["world"]) == 2
    reads = [synth.make_read(m, r, ml, seed=7000 + i, rc=bool(i & 1), margin=3.0 if i % 3 == 0 else 6.0) for i in range(n)]
    with pkg.Decoder(m, r, ml, list_size=L, max_deviation=20) as dec:
        want = dec.decode([x["post"] for x in reads], rc=[x["rc"] for x in reads])
    c, mm, s = sharding.pack_results(want, L, ml)
    assert np.array_equal(z["counts"], c) and np.array_equal(z["msgs"], mm)
    assert np.array_equal(z["scores"].view(np.uint32), s.view(np.uint32))


def _bench(args, env_extra, tmp_path, tag):
    dump = str(tmp_path / (tag + ".npz"))
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--dump-lists", dump], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0]), np.load(dump)


SMALL = ["--steps", "1", "--warmup", "1", "--total-reads", "10", "--mem-conv", "8", "--rate", "3", "--msg-len", "164",
         "--list-size", "8", "--no-cpu-baseline", "--slots", "3"]


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher: two rank processes, n_gpus 2 in the JSON line, and the gathered
    lists equal those of `--gpus 1` (run through the same spawn path with one rank)."""
    j2, z2 = _bench(["--gpus", "2"] + SMALL, {"LVA_BENCH_BACKEND": "gloo"}, tmp_path, "two")
    j1, z1 = _bench(["--gpus", "1"] + SMALL, {"LVA_BENCH_SPAWN": "1"}, tmp_path, "one")
    assert j2["n_gpus"] == 2 and j1["n_gpus"] == 1
    assert j2["scaling"] == "strong" and j2["config"]["gathered_lists"] == 10
    for k in ("counts", "msgs"):
        assert np.array_equal(z1[k], z2[k])
    assert np.array_equal(z1["scores"].view(np.uint32), z2["scores"].view(np.uint32))
    assert int(z1["counts"].min()) >= 1
    for j in (j1, j2):
        assert j["roofline"]["achieved"] > 0 and j["roofline"]["pair"]["avg_launch_ms"] >= j["roofline"]["avg_launch_ms"]
        assert j["config"]["h2d"] == "included" and j["config"]["h2d_bytes_per_step"] > 0


def test_bench_weak_mode_cycles_distinct_batches(tmp_path):
    j, z = _bench(["--gpus", "1", "--steps", "3", "--warmup", "0", "--reads-per-step", "5", "--pool", "12", "--mem-conv", "6",
                   "--rate", "1", "--msg-len", "60", "--list-size", "4", "--no-cpu-baseline", "--slots", "2"], {}, tmp_path, "weak")
    assert j["scaling"] == "weak" and j["config"]["distinct_reads_per_gpu"] == 15 and j["config"]["reads_per_step_per_gpu"] == 5
    assert z["counts"].shape[0] == 5
