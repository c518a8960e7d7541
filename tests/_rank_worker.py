"""Rank program of the multi-process tests: started by sharding.launch_ranks (torch.distributed.run),
joins the group with sharding.init_rank, decodes its shard and gathers on rank 0.

    _rank_worker.py OUT.npz fake|gpu  N_READS  [m r msg_len L]

"fake": a deterministic stand-in decoder (CPU container, gloo).  "gpu": the real Decoder (GPU box; with
LVA_DIST_BACKEND=gloo several ranks share GPU 0)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanopore_dna_storage_amd import sharding  # noqa: E402

L, MSG = 4, 12


def fake_posts(n):
    rng = np.random.default_rng(3)
    nblks = [int(x) for x in rng.integers(3, 60, size=n)]
    return [np.zeros((k, 40), np.float32) for k in nblks], [bool(i & 1) for i in range(n)]


def fake_decode(posts, rc):
    out = []
    for p, r in zip(posts, rc):
        n = p.shape[0]
        if n < 5:
            out.append(-6)
            continue
        cnt = 1 + n % L
        msgs = ((np.arange(cnt * MSG).reshape(cnt, MSG) + n + int(r)) % 2).astype(np.uint8)
        out.append((msgs, -np.arange(cnt, dtype=np.float32) - n))
    return out


def main():
    out_path, mode, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    dist, rank, world, devno, coll_dev = sharding.init_rank()
    if mode == "fake":
        if dist is not None and os.environ.get("LVA_TEST_AGREE") == "1":
            # the run's agreement step; LVA_TEST_ODD_RANK=r: rank r was started with another list size
            odd = int(os.environ.get("LVA_TEST_ODD_RANK", "-1"))
            rec = sharding.configuration_record(6, 1, MSG, L + (1 if rank == odd else 0), 20)
            sharding.assert_same_configuration(dist, rec, device=coll_dev)
        posts, rc = fake_posts(n)
        res = sharding.decode_sharded(fake_decode, posts, rc, L, MSG, dist=dist, device=coll_dev,
                                      shards=sharding.shard_strided(n, world))
        ls, ml = L, MSG
    else:
        m, r, ml, ls = (int(x) for x in sys.argv[4:8])
        import nanopore_dna_storage_amd as pkg
        from nanopore_dna_storage_amd import synth
        shards = sharding.shard_strided(n, world)
        # every rank builds only its own reads (the others stay None)
        posts, rc = [None] * n, [False] * n
        for i in shards[rank]:
            x = synth.make_read(m, r, ml, seed=7000 + int(i), rc=bool(i & 1), margin=3.0 if i % 3 == 0 else 6.0)
            posts[int(i)], rc[int(i)] = x["post"], x["rc"]
        with pkg.Decoder(m, r, ml, list_size=ls, max_deviation=20, device=devno, max_slots=3) as dec:
            res = sharding.decode_sharded(dec.decode, posts, rc, ls, ml, dist=dist, device=coll_dev, shards=shards)
    if rank == 0:
        c, mm, s = sharding.pack_results(res, ls, ml)
        np.savez(out_path, counts=c, msgs=mm, scores=s, world=world, grouped=int(dist is not None),
                 backend=(dist.get_backend() if dist is not None else ""))
    else:
        assert res is None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
