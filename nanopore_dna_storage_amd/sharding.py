"""Multi-GPU sharding of independent reads.

Reads are independent units -- the reference itself scales by running separate processes on
disjoint read-id files (util/extra/generate_read_id_files.py:23-36, merge_lists.py:11-21).
One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in CPU tests): reads are dealt to ranks longest-first so every rank gets the same work, each
rank decodes its shard with no data-path collective, and the decoded lists (about 230 B per
read) are gathered on rank 0 -- the path's only exchange step.
"""
import numpy as np


def shard_reads(nblks, world):
    """-> list of index arrays, one per rank.  Longest reads first, dealt round-robin in a
    serpentine order so per-rank sums of nblk (the work) are balanced."""
    order = np.argsort(-np.asarray(nblks, dtype=np.int64), kind="stable")
    shards = [[] for _ in range(world)]
    for n, idx in enumerate(order):
        rnd, pos = divmod(n, world)
        shards[pos if rnd % 2 == 0 else world - 1 - pos].append(int(idx))
    return [np.asarray(sorted(s), dtype=np.int64) for s in shards]


def pack_results(results, list_size, msg_len):
    """list of (msgs, scores) | error code -> (counts int32[n], msgs uint8[n, L, msg_len], scores f32[n, L])"""
    n = len(results)
    counts = np.zeros(n, np.int32)
    msgs = np.zeros((n, list_size, msg_len), np.uint8)
    scores = np.zeros((n, list_size), np.float32)
    for i, r in enumerate(results):
        if isinstance(r, int):
            counts[i] = r
        else:
            counts[i] = len(r[0])
            msgs[i, :len(r[0])] = r[0]
            scores[i, :len(r[0])] = r[1]
    return counts, msgs, scores


def unpack_results(counts, msgs, scores):
    out = []
    for i, c in enumerate(counts):
        c = int(c)
        out.append((msgs[i, :c].copy(), scores[i, :c].copy()) if c >= 0 else c)
    return out


def decode_sharded(decode_fn, posts, rc, list_size, msg_len, dist=None, device=None):
    """Decode `posts` across the ranks of an initialised torch.distributed group.
    decode_fn(posts_subset, rc_subset) -> list of results (this rank's Decoder.decode).
    Every rank passes the same posts/rc (or at least the same lengths); rank 0 gets the full
    result list in input order, other ranks get None."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return decode_fn(posts, rc)
    world, rank = dist.get_world_size(), dist.get_rank()
    shards = shard_reads([p.shape[0] for p in posts], world)
    mine = shards[rank]
    res = decode_fn([posts[i] for i in mine], [rc[i] for i in mine])
    counts, msgs, scores = pack_results(res, list_size, msg_len)
    # pad every shard to the largest so that a plain gather works
    cap = max(len(s) for s in shards)

    def pad(a):
        out = np.zeros((cap,) + a.shape[1:], a.dtype)
        out[:a.shape[0]] = a
        return torch.from_numpy(out).to(device) if device is not None else torch.from_numpy(out)

    tensors = [pad(counts), pad(msgs), pad(scores)]
    gathered = []
    for t in tensors:
        buf = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, buf, dst=0)
        gathered.append(buf)
    if rank != 0:
        return None
    out = [None] * len(posts)
    for r in range(world):
        c = gathered[0][r].cpu().numpy()[:len(shards[r])]
        m = gathered[1][r].cpu().numpy()[:len(shards[r])]
        s = gathered[2][r].cpu().numpy()[:len(shards[r])]
        for idx, item in zip(shards[r], unpack_results(c, m, s)):
            out[idx] = item
    return out
