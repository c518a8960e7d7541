"""Multi-GPU sharding of independent reads.

Reads are independent units -- the reference itself scales by running separate driver processes on
disjoint read-id files and merging their lists afterwards:
  * util/extra/generate_read_id_files.py:23-36   contiguous chunks of the read list, one file per worker
  * util/extra/pick_new_reads.py:8-9,21          strided chunks (lst[i::n]) of the reads not done yet
  * util/extra/merge_lists.py:11-21              worker after worker, list_<i>_<j> -> list_<done+j>
Here: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in CPU tests and when several ranks share one GPU), each rank decodes its shard with no data-path
collective, and the decoded lists (about 230 B per read) are gathered on rank 0 in input order --
the path's only exchange step.  `launch_ranks` starts the rank processes (the reference's "run N
copies of the driver") from a parent that has not touched the GPU.
"""
import os
import socket
import subprocess
import sys

import numpy as np


# ---------------------------------------------------------------------------------------------
# who decodes what
# ---------------------------------------------------------------------------------------------
def shard_reads(nblks, world):
    """-> list of index arrays, one per rank.  Longest reads first, dealt round-robin in a
    serpentine order so per-rank sums of nblk (the work) are balanced."""
    order = np.argsort(-np.asarray(nblks, dtype=np.int64), kind="stable")
    shards = [[] for _ in range(world)]
    for n, idx in enumerate(order):
        rnd, pos = divmod(n, world)
        shards[pos if rnd % 2 == 0 else world - 1 - pos].append(int(idx))
    return [np.asarray(sorted(s), dtype=np.int64) for s in shards]


def shard_strided(n, world):
    """pick_new_reads.py's chunkify: rank r takes reads r, r+world, r+2*world, ...  Needs no knowledge of
    the read lengths, so every rank can build (or load) only its own reads."""
    return [np.arange(r, n, world, dtype=np.int64) for r in range(world)]


def shard_contiguous(n, world):
    """generate_read_id_files.py: consecutive blocks of ceil(n/world) reads, the last one shorter."""
    per = -(-n // world) if world else 0
    return [np.arange(min(r * per, n), min((r + 1) * per, n), dtype=np.int64) for r in range(world)]


# ---------------------------------------------------------------------------------------------
# results <-> fixed-shape arrays (what travels in the gather)
# ---------------------------------------------------------------------------------------------
def pack_results(results, list_size, msg_len):
    """list of (msgs, scores) | error code -> (counts int32[n], msgs uint8[n, L, msg_len], scores f32[n, L])"""
    n = len(results)
    counts = np.zeros(n, np.int32)
    msgs = np.zeros((n, list_size, msg_len), np.uint8)
    scores = np.zeros((n, list_size), np.float32)
    for i, r in enumerate(results):
        if isinstance(r, (int, np.integer)):
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


def gather_results(local_results, shards, list_size, msg_len, dist=None, device=None):
    """The exchange step (merge_lists.py:11-21): every rank contributes the results of ITS shard
    (`local_results[j]` belongs to global read `shards[rank][j]`); rank 0 returns the full list in
    global read order, the other ranks return None.  `shards` = the same list of index arrays on
    every rank.  `device`: where the collective's tensors live ("cuda" for nccl/RCCL, None = CPU for gloo)."""
    import torch
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force_dist()):
        out = [None] * sum(len(s) for s in shards)
        for idx, item in zip(shards[0], local_results):
            out[int(idx)] = item
        return out
    world, rank = dist.get_world_size(), dist.get_rank()
    assert len(shards) == world and len(local_results) == len(shards[rank])
    counts, msgs, scores = pack_results(local_results, list_size, msg_len)
    cap = max(max(len(s) for s in shards), 1)         # pad every shard to the largest: a plain gather works

    def pad(a):
        out = np.zeros((cap,) + a.shape[1:], a.dtype)
        out[:a.shape[0]] = a
        t = torch.from_numpy(out)
        return t.to(device) if device is not None else t

    gathered = []
    for t in (pad(counts), pad(msgs), pad(scores)):
        buf = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, buf, dst=0)
        gathered.append(buf)
    if rank != 0:
        return None
    out = [None] * sum(len(s) for s in shards)
    for r in range(world):
        k = len(shards[r])
        c = gathered[0][r].cpu().numpy()[:k]
        m = gathered[1][r].cpu().numpy()[:k]
        s = gathered[2][r].cpu().numpy()[:k]
        for idx, item in zip(shards[r], unpack_results(c, m, s)):
            out[int(idx)] = item
    return out


def decode_sharded(decode_fn, posts, rc, list_size, msg_len, dist=None, device=None, shards=None):
    """Decode `posts` across the ranks of an initialised torch.distributed group.
    decode_fn(posts_subset, rc_subset) -> list of results (this rank's Decoder.decode).
    Every rank passes the same posts/rc (entries of other ranks' shards may be None when `shards` is
    given); rank 0 gets the full result list in input order, other ranks get None."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force_dist()):
        return decode_fn(posts, rc)
    world, rank = dist.get_world_size(), dist.get_rank()
    if shards is None:
        shards = shard_reads([p.shape[0] for p in posts], world)
    mine = shards[rank]
    res = decode_fn([posts[i] for i in mine], [rc[i] for i in mine])
    return gather_results(res, shards, list_size, msg_len, dist=dist, device=device)


# ---------------------------------------------------------------------------------------------
# starting the ranks
# ---------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def force_dist():
    """LVA_FORCE_DIST=1: a single rank still creates its process group and runs every collective of the multi-rank
    path (group of one) -- how the RCCL branch is exercised on a 1-GPU box."""
    return os.environ.get("LVA_FORCE_DIST", "") not in ("", "0")


def assert_one_gpu_per_rank(dist, device="cuda"):
    """every rank of an RCCL group must own its GPU: all_gather of (host, device ordinal) pairs, all distinct"""
    import zlib
    import torch
    world = dist.get_world_size()
    me = torch.tensor([zlib.crc32(socket.gethostname().encode()), torch.cuda.current_device()], dtype=torch.int64, device=device)
    got = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(got, me)
    pairs = {(int(x[0].item()), int(x[1].item())) for x in got}
    assert len(pairs) == world, "ranks share a GPU: %r" % (sorted(pairs),)
    return sorted(pairs)


def configuration_record(mem_conv, rate, msg_len, list_size, max_deviation, sync_marker=None, sync_period=0, kernel=0):
    """What every rank of a run must agree on, as eight int64: the decoder's configuration, the ABI version and build id of the
    library this process loaded, and a checksum of the code tables that library builds from the configuration (both
    orientations: positions, block types, valid-state masks, predecessor tables -- what the kernels index)."""
    import ctypes
    import zlib
    from . import _lib
    L = _lib.load_library()
    sm = sync_marker.encode() if sync_marker else None
    info = _lib.CodeInfoStruct()
    crc = 0
    for rc in (0, 1):
        st = L.lva_code_describe(mem_conv, rate, msg_len, rc, sm, sync_period, ctypes.byref(info))
        if st != 0:
            raise _lib.LvaError(st)
        npos, nconv = info.nstate_pos, info.nstate_conv
        pos2msg = np.zeros(npos, np.uint32); ptype = np.zeros(npos, np.uint8)
        vmask = np.zeros(npos, np.uint32); vval = np.zeros(npos, np.uint32); pred = np.zeros(4 * nconv, np.uint16)
        st = L.lva_code_tables(mem_conv, rate, msg_len, rc, sm, sync_period, pos2msg.ctypes.data, ptype.ctypes.data, vmask.ctypes.data,
                               vval.ctypes.data, pred.ctypes.data)
        if st != 0:
            raise _lib.LvaError(st)
        for a in (pos2msg, ptype, vmask, vval, pred):
            crc = zlib.crc32(a.tobytes(), crc)
    md = int(max_deviation) & 0xFFFFFFFF
    return [int(mem_conv), int(rate), int(msg_len), int(list_size) | (int(kernel) << 32), md | (int(sync_period) << 32),
            int(L.lva_abi_version()), zlib.crc32(_lib.build_id().encode()) | (zlib.crc32(sm or b"") << 32) & 0x7FFFFFFFFFFFFFFF, crc]


def assert_same_configuration(dist, record, device=None):
    """The run's one agreement step (the reference ASSUMES identical workers: util/extra/merge_lists.py:11-21 concatenates their
    lists unchecked): rank 0 broadcasts its configuration_record, every rank compares it with its own -- a stale liblva_hip.so or
    another list size on one rank would otherwise only show as a wrong or slow shard -- and a MIN all-reduce of the verdicts
    makes EVERY rank leave with the same SystemExit instead of some of them hanging in the first gather.  Microseconds, before
    the timed region.  `device`: where the collective's tensors live ("cuda" for nccl/RCCL, None = CPU for gloo)."""
    import torch
    if dist is None or not dist.is_initialized():
        return
    mine = torch.tensor([int(x) for x in record], dtype=torch.int64, device=device or "cpu")
    ref = mine.clone()
    dist.broadcast(ref, src=0)
    same = bool(torch.equal(ref.cpu(), mine.cpu()))
    ok = torch.tensor([1 if same else 0], dtype=torch.int64, device=device or "cpu")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) != 1:
        names = ["mem_conv", "rate", "msg_len", "list_size|kernel", "max_deviation|sync_period", "abi_version", "build_id|sync_marker", "code_tables"]
        diff = [n for n, a, b in zip(names, ref.cpu().tolist(), mine.cpu().tolist()) if a != b]
        raise SystemExit("rank %d: the ranks of this run do not agree on the decoder (%s): same flags and the same liblva_hip.so on every rank, please"
                         % (dist.get_rank(), ("this rank differs from rank 0 in " + ", ".join(diff)) if diff else "another rank differs from rank 0"))


def launch_ranks(script, argv, nproc, env=None, capture=False, module=False):
    """Run `script argv...` (`python -m script argv...` when module=True) as `nproc` ranks of one node through torch.distributed.run (one process
    per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) and wait for them.

    The caller must not have initialised the GPU: the ranks are fresh child processes started with
    subprocess (never an exec of the current process).  -> the launcher's exit code (and its stdout when
    capture=True)."""
    # --standalone: torchrun picks the rendezvous port itself (no bind-close-reuse race between concurrent launches)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(int(nproc))] + (["-m"] if module else []) + [script] + list(argv)
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this driver
    e.setdefault("OMP_NUM_THREADS", "1")
    if capture:
        p = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, text=True)
        return p.returncode, p.stdout
    return subprocess.run(cmd, env=e).returncode


def init_rank(backend=None):
    """In a rank process started by torch.distributed.run: bind this rank to its GPU and join the group.
    -> (dist module or None, rank, world, device ordinal, collective tensor device).
    backend "nccl" (default; = RCCL) needs one GPU per rank; "gloo" lets several ranks share a GPU
    (tests on a 1-GPU box)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and not force_dist():
        return None, 0, 1, local, None
    if world == 1:                                        # LVA_FORCE_DIST: a group of one, started without a launcher
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
    import torch
    import torch.distributed as dist
    backend = backend or os.environ.get("LVA_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()                      # (does not initialise the GPU)
    if backend == "nccl":
        if ndev < world and int(os.environ.get("LOCAL_WORLD_SIZE", world)) > ndev:
            raise RuntimeError("%d ranks on this node but only %d GPUs: one GPU per rank is required with RCCL "
                               "(set LVA_DIST_BACKEND=gloo to share a GPU in tests)" % (world, ndev))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        return dist, rank, world, local, "cuda"
    dist.init_process_group(backend)
    return dist, rank, world, local % max(ndev, 1), None
