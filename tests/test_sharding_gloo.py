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
