"""Multi-process path on CPU: world_size 2, gloo backend.  The decode itself needs a GPU, so a
deterministic stand-in plays the per-rank decoder; what is tested is the sharding, the gather on
rank 0 and the input-order reassembly (nanopore_dna_storage_amd/sharding.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nanopore_dna_storage_amd import sharding

L, MSG = 4, 12


def fake_decode(posts, rc):
    out = []
    for p, r in zip(posts, rc):
        n = p.shape[0]
        if n < 5:
            out.append(-6)
            continue
        cnt = 1 + n % L
        msgs = ((np.arange(cnt * MSG).reshape(cnt, MSG) + n + int(r)) % 2).astype(np.uint8)
        scores = -np.arange(cnt, dtype=np.float32) - n
        out.append((msgs, scores))
    return out


def make_inputs():
    rng = np.random.default_rng(3)
    nblks = [int(x) for x in rng.integers(3, 60, size=11)]
    posts = [np.zeros((n, 40), np.float32) for n in nblks]
    rc = [bool(i & 1) for i in range(len(posts))]
    return posts, rc


def worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    posts, rc = make_inputs()
    res = sharding.decode_sharded(fake_decode, posts, rc, L, MSG, dist=dist)
    if rank == 0:
        q.put([(r if isinstance(r, int) else (r[0].tolist(), r[1].tolist())) for r in res])
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_shards_are_balanced_and_complete():
    nblks = [500, 480, 520, 510, 90, 505, 495, 530]
    shards = sharding.shard_reads(nblks, 3)
    assert sorted(np.concatenate(shards).tolist()) == list(range(8))
    work = [sum(nblks[i] for i in s) for s in shards]
    assert max(work) - min(work) <= max(nblks)


def test_gather_on_rank0_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    posts, rc = make_inputs()
    want = [(r if isinstance(r, int) else (r[0].tolist(), r[1].tolist())) for r in fake_decode(posts, rc)]
    assert got == want


def test_strided_and_contiguous_shards():
    for n, w in ((11, 3), (8, 8), (5, 8), (0, 2)):
        for fn in (sharding.shard_strided, sharding.shard_contiguous):
            sh = fn(n, w)
            assert len(sh) == w and sorted(np.concatenate(sh).tolist()) == list(range(n))
    assert sharding.shard_strided(7, 3)[1].tolist() == [1, 4]                 # pick_new_reads.py: lst[i::n]
    assert sharding.shard_contiguous(7, 3)[2].tolist() == [6]                 # generate_read_id_files.py: last file shorter


def test_launch_ranks_world2_gather_in_input_order(tmp_path):
    """sharding.launch_ranks starts two rank processes (torch.distributed.run, gloo); each decodes a strided shard
    with the stand-in decoder; rank 0's gathered list equals the single-process result."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import _rank_worker as W
    out = str(tmp_path / "g.npz")
    env = dict(os.environ, LVA_DIST_BACKEND="gloo")
    rc_ = sharding.launch_ranks(os.path.join(here, "_rank_worker.py"), [out, "fake", "13"], 2, env=env)
    assert rc_ == 0
    z = np.load(out)
    assert int(z["world"]) == 2
    posts, rc = W.fake_posts(13)
    c, m, s = sharding.pack_results(W.fake_decode(posts, rc), W.L, W.MSG)
    assert np.array_equal(z["counts"], c) and np.array_equal(z["msgs"], m) and np.array_equal(z["scores"], s)


def test_bench_parent_refuses_mismatched_group(tmp_path):
    """bench.py run as a rank of a group whose size differs from --gpus stops before touching the GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode != 0 and "launcher started 1 ranks" in p.stderr


def test_forced_group_of_one_runs_the_collectives(tmp_path):
    """LVA_FORCE_DIST=1: one rank, started WITHOUT a launcher, still creates its process group (gloo here, nccl on
    the GPU box) and goes through gather_results' collective branch; result equals the plain single-process one."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import _rank_worker as W
    out = str(tmp_path / "g1.npz")
    env = dict(os.environ, LVA_DIST_BACKEND="gloo", LVA_FORCE_DIST="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(here, "_rank_worker.py"), out, "fake", "9"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    z = np.load(out)
    assert int(z["world"]) == 1 and int(z["grouped"]) == 1
    posts, rc = W.fake_posts(9)
    c, m, s = sharding.pack_results(W.fake_decode(posts, rc), W.L, W.MSG)
    assert np.array_equal(z["counts"], c) and np.array_equal(z["msgs"], m) and np.array_equal(z["scores"], s)


def test_launch_ranks_world8_uneven_shards_and_error_codes(tmp_path):
    """Eight ranks (the node the driver's scaling run uses; gloo here): 67 = 8 * 8 + 3 reads in strided shards of 9 and 8, reads
    too short to decode (error codes instead of lists) in several shards; rank 0's gathered list equals the single-process result
    in input order (util/extra/merge_lists.py:11-21)."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import _rank_worker as W
    n, world = 67, 8
    posts, rc = W.fake_posts(n)
    want = W.fake_decode(posts, rc)
    shards = sharding.shard_strided(n, world)
    assert sorted(len(s) for s in shards) == [8] * 5 + [9] * 3
    bad_ranks = {r for r, s in enumerate(shards) for i in s if isinstance(want[int(i)], int)}
    assert len(bad_ranks) >= 2, "the fixture should put error-coded reads into several ranks"
    out = str(tmp_path / "g8.npz")
    env = dict(os.environ, LVA_DIST_BACKEND="gloo")
    assert sharding.launch_ranks(os.path.join(here, "_rank_worker.py"), [out, "fake", str(n)], world, env=env) == 0
    z = np.load(out)
    assert int(z["world"]) == world
    c, m, s = sharding.pack_results(want, W.L, W.MSG)
    assert np.array_equal(z["counts"], c) and np.array_equal(z["msgs"], m) and np.array_equal(z["scores"], s)
    assert (z["counts"] < 0).sum() == sum(isinstance(w, int) for w in want) > 0


def test_bench_eight_ranks_strong_scaling_arguments(tmp_path):
    """`bench.py --gpus 8 --total-reads 100000` -- the driver's configs[2] run -- with a stand-in decoder (LVA_BENCH_STUB, honoured
    only with LVA_TESTING): self-launch of eight ranks, strided shards of 12 500 reads, the clock's all_reduce, the per_rank
    all_gather, the gather of 100 000 lists on rank 0 and the JSON line, without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LVA_BENCH_BACKEND="gloo", LVA_BENCH_STUB="1", LVA_TESTING="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    dump = str(tmp_path / "lists.npz")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--total-reads", "100000", "--steps", "1", "--warmup", "0",
                        "--list-size", "4", "--msg-len", "12", "--no-cross-check", "--dump-lists", dump],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    j = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 8 and j["scaling"] == "strong" and j["config"]["gathered_lists"] == 100000
    assert len(j["config"]["per_rank"]) == 8 and [r["rank"] for r in j["config"]["per_rank"]] == list(range(8))
    assert j["config"]["reads_per_step_per_gpu"] == 12500 and j["config"]["dist_backend"] == "gloo"
    z = np.load(dump)
    n = np.array([3 + (gi * 7919) % 57 for gi in range(100000)])                   # the stand-in reads' block counts
    assert np.array_equal(z["counts"], np.where(n < 5, -6, 1 + n % 4))             # every list in global read order, error codes too


def test_ranks_that_disagree_all_leave(tmp_path):
    """The run's agreement step (sharding.assert_same_configuration: rank 0 broadcasts configuration, library build and a checksum
    of the code tables; a MIN all-reduce spreads the verdict): eight ranks that agree run to the end; with ONE rank started on
    another list size every rank exits non-zero -- nobody hangs in the first gather (util/extra/merge_lists.py:11-21 assumes
    identical workers; this checks it)."""
    import sys
    import time
    here = os.path.dirname(os.path.abspath(__file__))
    out = str(tmp_path / "agree.npz")
    env = dict(os.environ, LVA_DIST_BACKEND="gloo", LVA_TEST_AGREE="1")
    assert sharding.launch_ranks(os.path.join(here, "_rank_worker.py"), [out, "fake", "24"], 8, env=env) == 0
    assert int(np.load(out)["world"]) == 8
    out2 = str(tmp_path / "disagree.npz")
    t0 = time.time()
    rc, _ = sharding.launch_ranks(os.path.join(here, "_rank_worker.py"), [out2, "fake", "24"], 8, env=dict(env, LVA_TEST_ODD_RANK="5"), capture=True)
    assert rc != 0 and not os.path.exists(out2)
    assert time.time() - t0 < 120, "the ranks should leave at once, not wait for a collective's time-out"
    rec = sharding.configuration_record(6, 1, 12, 4, 20)
    assert len(rec) == 8 and rec != sharding.configuration_record(6, 1, 12, 8, 20) and rec != sharding.configuration_record(8, 1, 12, 4, 20)
