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
    # every rank reports its own rate, launch time and slot occupancy (a bad scaling point names its rank)
    pr = j2["config"]["per_rank"]
    assert [x["rank"] for x in pr] == [0, 1] and all(x["reads_s"] > 0 and x["avg_launch_ms"] > 0 and x["mean_active_slots"] > 0 for x in pr)
    assert j1["config"]["per_rank"] is None or len(j1["config"]["per_rank"]) == 1
    for k in ("counts", "msgs"):
        assert np.array_equal(z1[k], z2[k])
    assert np.array_equal(z1["scores"].view(np.uint32), z2["scores"].view(np.uint32))
    assert int(z1["counts"].min()) >= 1
    for j in (j1, j2):
        assert j["roofline"]["achieved"] > 0 and j["roofline"]["pair"]["avg_launch_ms"] >= j["roofline"]["avg_launch_ms"]
        assert j["config"]["h2d"] == "included" and j["config"]["h2d_bytes_per_step"] > 0


def test_rccl_branch_on_one_gpu(tmp_path):
    """LVA_FORCE_DIST=1: a single rank runs init_process_group("nccl", device_id=...), the one-GPU-per-rank all_gather,
    the max all_reduce of the step time and gather_results on cuda tensors -- every RCCL call of the N > 1 path, on
    the one GPU of this box; lists equal those of the plain run."""
    j1, z1 = _bench(["--gpus", "1"] + SMALL, {"LVA_FORCE_DIST": "1"}, tmp_path, "forced")
    j0, z0 = _bench(["--gpus", "1"] + SMALL, {}, tmp_path, "plain")
    assert j1["config"]["dist_backend"] == "nccl" and j1["config"]["world"] == 1 and j1["n_gpus"] == 1
    assert j0["config"]["dist_backend"] is None
    for k in ("counts", "msgs"):
        assert np.array_equal(z1[k], z0[k])
    assert np.array_equal(z1["scores"].view(np.uint32), z0["scores"].view(np.uint32))
    # the rank worker with the REAL decoder through the forced nccl group
    out = str(tmp_path / "g1.npz")
    env = dict(os.environ, LVA_FORCE_DIST="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LVA_DIST_BACKEND"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(HERE, "_rank_worker.py"), out, "gpu", "7", "6", "1", "60", "4"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    z = np.load(out)
    assert str(z["backend"]) == "nccl" and int(z["world"]) == 1
    reads = [synth.make_read(6, 1, 60, seed=7000 + i, rc=bool(i & 1), margin=3.0 if i % 3 == 0 else 6.0) for i in range(7)]
    with pkg.Decoder(6, 1, 60, list_size=4, max_deviation=20) as dec:
        want = dec.decode([x["post"] for x in reads], rc=[x["rc"] for x in reads])
    c, mm, s = sharding.pack_results(want, 4, 60)
    assert np.array_equal(z["counts"], c) and np.array_equal(z["msgs"], mm)
    assert np.array_equal(z["scores"].view(np.uint32), s.view(np.uint32))


def test_bench_weak_mode_cycles_distinct_batches(tmp_path):
    j, z = _bench(["--gpus", "1", "--steps", "3", "--warmup", "0", "--reads-per-step", "5", "--pool", "12", "--mem-conv", "6",
                   "--rate", "1", "--msg-len", "60", "--list-size", "4", "--no-cpu-baseline", "--slots", "2"], {}, tmp_path, "weak")
    assert j["scaling"] == "weak" and j["config"]["distinct_reads_per_gpu"] == 15 and j["config"]["reads_per_step_per_gpu"] == 5
    assert z["counts"].shape[0] == 5


def _manifest(tmp_path, n=5):
    """mixed manifest: windows known (6 columns) and untruncated posts (3 columns)"""
    sb, eb = "CACCTGTGCTGCGTCAGGCTGTGTC", "GCTGTCCGTTCCGCATTGACACGGC"
    rows, msgs = [], []
    for i in range(n):
        p = tmp_path / ("r%d.post" % i)
        if i % 2 == 0:
            rd = synth.make_read(6, 1, 60, 300 + i, rc=bool(i & 2), margin=6.0)
            pad = np.full((3, 40), -3.7, np.float32)
            np.concatenate([pad, rd["post"], pad]).tofile(p)
            rows.append("read%d\tref%d\t%s\t%d\t%d\t%d" % (i, i, p, 3, 3 + rd["post"].shape[0] - 1, int(rd["rc"])))
        else:
            rd = synth.make_barcoded_read(6, 1, 60, 300 + i, sb, eb, rc=bool(i & 2), margin=6.0, flank=(5, 14))
            rd["post"].tofile(p)
            rows.append("read%d\tref%d\t%s" % (i, i, p))
        msgs.append("".join(map(str, rd["msg"])))
    man = tmp_path / "manifest.tsv"
    man.write_text("\n".join(rows) + "\n")
    return man, msgs, ["--mem_conv", "6", "--msg_len", "60", "--rate_conv", "1", "--list_size", "4",
                       "--start_barcode", sb, "--end_barcode", eb]


def test_generate_decoded_lists_two_ranks_and_resume(tmp_path):
    """--gpus 2 --chunk 2: the driver starts its own rank processes and walks the manifest two reads at a time (three
    gathers for five reads); rank 0 gathers the lists and writes every file (temporary name + rename);
    --resume: list files that exist are left alone (the reference's per-read checkpoint, pick_new_reads.py:11-18)."""
    man, msgs, flags = _manifest(tmp_path)
    d1, d2 = tmp_path / "one", tmp_path / "two"
    d1.mkdir(); d2.mkdir()
    env = dict(os.environ, LVA_DIST_BACKEND="gloo", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    base = [sys.executable, "-m", "nanopore_dna_storage_amd.generate_decoded_lists", "--post_manifest", str(man)] + flags
    p1 = subprocess.run(base + ["--out_prefix", str(d1 / "list"), "--info_file", str(d1 / "info.txt")], env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p1.returncode == 0, p1.stderr[-2000:]
    p2 = subprocess.run(base + ["--out_prefix", str(d2 / "list"), "--info_file", str(d2 / "info.txt"), "--gpus", "2", "--chunk", "2"], env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p2.returncode == 0, p2.stderr[-2000:]
    assert (d1 / "info.txt").read_text() == (d2 / "info.txt").read_text()
    for i, msg in enumerate(msgs):
        a, b = (d1 / ("list_%d" % i)).read_text(), (d2 / ("list_%d" % i)).read_text()
        assert a == b and a.split()[0] == msg
    # the same manifest through a forced RCCL group of one: gather_results on cuda tensors + gather_object
    d3 = tmp_path / "forced"
    d3.mkdir()
    env3 = dict(env, LVA_FORCE_DIST="1")
    env3.pop("LVA_DIST_BACKEND")
    p4 = subprocess.run(base + ["--out_prefix", str(d3 / "list"), "--info_file", str(d3 / "info.txt")], env=env3,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p4.returncode == 0, p4.stderr[-2000:]
    assert "backend nccl" in p4.stdout
    for i in range(len(msgs)):
        assert (d1 / ("list_%d" % i)).read_text() == (d3 / ("list_%d" % i)).read_text()
    # resume: a marker in place of list_1 survives, a deleted list_2 comes back
    (d1 / "list_1").write_text("kept\n")
    keep2 = (d1 / "list_2").read_text()
    (d1 / "list_2").unlink()
    p3 = subprocess.run(base + ["--out_prefix", str(d1 / "list"), "--info_file", str(d1 / "info.txt"), "--resume"], env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p3.returncode == 0, p3.stderr[-2000:]
    assert (d1 / "list_1").read_text() == "kept\n" and (d1 / "list_2").read_text() == keep2


def test_reference_path_launcher_decodes(tmp_path):
    """viterbi/viterbi_nanopore.out, the reference's PATH_TO_CPP_EXEC (helper.py:19, simulator.py:30,
    generate_decoded_lists.py:37), called exactly as helper.py:305 calls it"""
    from golden_util import GOLDEN, load_case
    name = "m6_r1_L4_rc"
    m, post, lines = load_case(name)
    out_file = tmp_path / "dec"
    p = subprocess.run([os.path.join(ROOT, "viterbi", "viterbi_nanopore.out"), "-m", "decode", "-i", os.path.join(GOLDEN, name + ".post"),
                        "-o", str(out_file), "--mem-conv", "6", "--msg-len", "60", "-l", "4", "-t", "8", "-r", "1", "--rc",
                        "--max-deviation", "20"], stdout=subprocess.PIPE, text=True, timeout=600, cwd=str(tmp_path))
    assert p.returncode == 0 and p.stdout == "Reverse complement flag detected.\n"
    assert out_file.read_text() == "".join(ln + "\n" for ln in lines)
