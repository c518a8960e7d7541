#!/usr/bin/env python3
"""bench.py -- reads decoded/sec on the north-star configuration.

Metric: reads decoded per second END TO END for the decode path -- host->device copy of the posterior
matrices, every trellis step, final selection, device->host copy of the lists -- on BASELINE.json
configs[1]'s shape: mem_conv=11, rate=5 (5/6), list_size=8, msg_len=180, max_deviation=20, forward and
reverse-complement reads mixed.  It replaces the per-read loop of the reference's
generate_decoded_lists.py:50-98 (one decoder subprocess per read).

One "step" = one call of the hot path's C-ABI entry lva_decode_batch on one batch of host-resident
synthetic reads per GPU (--reads-per-step, default 2x the decoder's read slots, so slots are refilled
inside the timed region); consecutive steps take consecutive batches out of a pool of --pool distinct
reads per GPU (default 512).  `--resident` keeps the posteriors in HBM instead (lva_decode_batch_device).

N GPUs: `python bench.py --gpus N` starts N rank processes itself (torch.distributed.run, one per GPU,
before anything touches the GPU in the parent) unless it already runs as a rank (WORLD_SIZE set by the
driver's launcher).  Reads are independent: every rank decodes its own shard with no data-path
collective; the decoded lists are gathered on rank 0 through sharding.gather_results (RCCL) after the
timed region.  Weak scaling by default (fixed reads per GPU); `--total-reads T` fixes the job instead
(strong scaling, configs[2]'s shape: rank r takes reads r, r+N, ...).

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline      algorithmic bytes (SURVEY 8d formula) / HIP-event time of the dominant kernel, measured live
                on the decoder's stream around every launch, against the 8 TB/s HBM peak
  cpu_baseline  the unmodified reference binary (oracle/_ref) on this host, all (<=16) OpenMP threads, on a
                bounded sample of the same reads; cpu_baseline_single_thread: the same with -t 1
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M, RATE, MSG_LEN, LIST, MAXDEV = 11, 5, 180, 8, 20


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads-per-step", type=int, default=0, help="reads per GPU per step (0 = 2 x read slots)")
    ap.add_argument("--pool", type=int, default=512, help="distinct synthetic reads per GPU the steps cycle through")
    ap.add_argument("--total-reads", type=int, default=0, help="strong scaling: reads per step over ALL GPUs")
    ap.add_argument("--resident", action="store_true", help="posteriors resident in HBM before the timed region")
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--mem-conv", type=int, default=M)
    ap.add_argument("--rate", type=int, default=RATE)
    ap.add_argument("--msg-len", type=int, default=MSG_LEN)
    ap.add_argument("--list-size", type=int, default=LIST)
    ap.add_argument("--max-deviation", type=int, default=MAXDEV)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--cpu-reads", type=int, default=3, help="reads of the multi-thread reference sample")
    ap.add_argument("--no-launch-events", action="store_true", help="no per-launch HIP events (roofline from the span)")
    ap.add_argument("--no-cross-check", action="store_true", help="skip the exact-kernel check of the last timed batch")
    ap.add_argument("--cross-check-reads", type=int, default=0, help="check only the first K reads of the last timed batch (0 = all)")
    ap.add_argument("--dump-lists", type=str, default="", help="rank 0 writes the gathered lists of the last step (npz)")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the short driver-timed runs of configs[3] and configs[4]")
    ap.add_argument("--extra-reads", type=int, default=32, help="reads per extra configuration")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# CPU leg: the unmodified reference decoder (test infrastructure under oracle/), outside the timed region
# ---------------------------------------------------------------------------------------------
def _ref_cmd(O, a, post_path, out_path, rc, threads):
    cmd = [O.REF_BIN, "-m", "decode", "-i", post_path, "-o", out_path, "--msg-len", str(a.msg_len), "--mem-conv",
           str(a.mem_conv), "-r", str(a.rate), "-l", str(a.list_size), "-t", str(threads),
           "--max-deviation", str(a.max_deviation)]
    if rc:
        cmd.append("--rc")
    return cmd


class RefJob:
    """one reference decode as a child process (so that the single-thread sample can run beside the GPU leg)"""

    def __init__(self, O, a, post, rc, threads, tmpdir):
        self.post_path = os.path.join(tmpdir, "ref_%d_%d.post" % (threads, id(self)))
        self.out_path = self.post_path + ".out"
        post.tofile(self.post_path)
        self.t0 = time.time()
        self.p = subprocess.Popen(_ref_cmd(O, a, self.post_path, self.out_path, rc, threads), stdout=subprocess.DEVNULL)
        self.wall = None

    def wait(self, timeout=None):
        rc_ = self.p.wait(timeout=timeout)
        self.wall = time.time() - self.t0
        lines = open(self.out_path).read().split("\n")[:-1] if (rc_ == 0 and os.path.exists(self.out_path)) else None
        for f in (self.post_path, self.out_path):
            if os.path.exists(f):
                os.remove(f)
        return rc_, lines


def as_lines(res):
    return ["".join("1" if b else "0" for b in row) for row in res[0]]


class StubDecoder:
    """LVA_BENCH_STUB=1 together with LVA_TESTING=1 (tests/test_sharding_gloo.py): a stand-in for the GPU decoder so that the
    8-rank argument / shard / gather / JSON path of this script runs in a container without GPUs.  Never measures anything."""

    def __init__(self, list_size, msg_len, slots):
        self.L, self.ml, self.slots, self.p = list_size, msg_len, slots or 128, None

    def profile(self):
        return self.p or dict(slots=self.slots, kernel=1)

    def set_launch_events(self, on):
        pass

    def decode_packed(self, flat, off, rc):
        import numpy as np
        out = []
        for i in range(len(off) - 1):
            n = int(off[i + 1] - off[i])
            if n < 5:
                out.append(-6)                            # LVA_ERR_POST_TOO_SHORT
                continue
            cnt = 1 + n % self.L
            msgs = ((np.arange(cnt * self.ml).reshape(cnt, self.ml) + n + int(rc[i])) % 2).astype(np.uint8)
            out.append((msgs, -np.arange(cnt, dtype=np.float32) - n))
        nb = int(off[-1] - off[0])
        self.p = dict(slots=self.slots, kernel=1, step_kernel_ms=1.0, dominant_kernel_ms=1.0, step_pair_ms=1.0, algorithmic_bytes=1e6 * nb,
                      working_bytes=9e5 * nb, step_launches=nb, timed_launches=nb, read_steps=nb, fixup_states=0, h2d_ms=0.0,
                      h2d_bytes=160 * nb, total_ms=1.0, fixup_reason=[0, 0, 0, 0])
        return out

    def close(self):
        pass


def run_extra_config(pkg, synth, np, name, m, r, L, golden, n_reads, devno, build_id):
    """One short, driver-timed run of another BASELINE.json configuration at the decoder's DEFAULT slot count: n_reads reads in one
    lva_decode_batch call (the first of them golden fixtures of the reference binary, checked), HIP events around every launch."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_util as G
    posts, rcs, want = [], [], []
    for g in golden:
        meta, post, lines = G.load_case(g)
        posts.append(post); rcs.append(bool(meta["rc"])); want.append(lines)
    for i in range(max(0, n_reads - len(golden))):
        x = synth.make_read(m, r, MSG_LEN, seed=5000 + i, rc=bool(i & 1), margin=3.0 if i % 5 == 0 else 6.0)
        posts.append(x["post"]); rcs.append(x["rc"])
    with pkg.Decoder(m, r, MSG_LEN, list_size=L, max_deviation=MAXDEV, device=devno) as d2:
        d2.decode(posts[:2], rc=rcs[:2])                      # (first call: allocations, code upload)
        d2.set_launch_events(True)
        t0 = time.perf_counter()
        out = d2.decode(posts, rc=rcs)
        dt = time.perf_counter() - t0
        p = d2.profile()
    for i, w in enumerate(want):
        assert as_lines(out[i]) == w, "%s: list of golden read %s differs from the reference's" % (name, golden[i])
    dom = p["dominant_kernel_ms"]
    return {"workload": "%s: mem_conv=%d rate=%d list_size=%d msg_len=%d max_deviation=%d, %d reads in one lva_decode_batch call, %d read slots (the decoder's default)"
                        % (name, m, r, L, MSG_LEN, MAXDEV, len(posts), p["slots"]),
            "reads_s": len(posts) / dt, "avg_launch_ms": dom / max(p["timed_launches"], 1), "launches": p["step_launches"],
            "mean_active_slots": p["read_steps"] / max(p["step_launches"], 1),
            "achieved": (p["algorithmic_bytes"] / 1e9) / (dom / 1e3) if dom > 0 else None,
            "frac": (p["algorithmic_bytes"] / 1e9) / (dom / 1e3) / 8000.0 if dom > 0 else None,
            "pair_avg_launch_ms": p["step_pair_ms"] / max(p["timed_launches"], 1),
            "golden_checked": list(golden), "kernel_mode": p["kernel"], "slots": p["slots"], "build_id": build_id}


def main():
    a = parse()
    in_group = "WORLD_SIZE" in os.environ
    # ---- self-launch: N ranks as fresh child processes; this parent never touches the GPU ----
    if a.gpus > 1 and not in_group:
        from nanopore_dna_storage_amd import sharding
        sys.exit(sharding.launch_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus))
    if os.environ.get("LVA_BENCH_SPAWN") == "1" and not in_group:        # tests: the spawn path with one rank
        from nanopore_dna_storage_amd import sharding
        sys.exit(sharding.launch_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus))

    import numpy as np
    from nanopore_dna_storage_amd import sharding
    backend = os.environ.get("LVA_BENCH_BACKEND") or os.environ.get("LVA_DIST_BACKEND") or "nccl"
    dist, rank, world, devno, coll_dev = sharding.init_rank(backend)
    if in_group and world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (a.gpus, world))

    import nanopore_dna_storage_amd as pkg
    from nanopore_dna_storage_amd import synth, _lib

    L = pkg.load_library()
    build_id = _lib.build_id()
    stub = os.environ.get("LVA_BENCH_STUB") == "1" and os.environ.get("LVA_TESTING") == "1"
    dec = StubDecoder(a.list_size, a.msg_len, a.slots) if stub else pkg.Decoder(
        a.mem_conv, a.rate, a.msg_len, list_size=a.list_size, max_deviation=a.max_deviation, device=devno, max_slots=a.slots, kernel=a.kernel)
    slots = dec.profile()["slots"]
    if dist is not None and backend == "nccl":
        sharding.assert_one_gpu_per_rank(dist)
    if dist is not None:
        # the run's one agreement step: same flags, same library build, same code tables on every rank (or every rank leaves)
        sharding.assert_same_configuration(dist, sharding.configuration_record(a.mem_conv, a.rate, a.msg_len, a.list_size, a.max_deviation,
                                                                               kernel=a.kernel), device=coll_dev)

    # ---- the reads of this rank ------------------------------------------------------------------
    strong = a.total_reads > 0
    if strong:
        shards = sharding.shard_strided(a.total_reads, world)
        mine = shards[rank]                       # global read indices of this rank, the same every step
        per_step = len(mine)
        pool_idx = [int(i) for i in mine]
        nbatch = 1
    else:
        per_step = a.reads_per_step or 2 * slots
        nbatch = max(1, -(-max(a.pool, 1) // per_step))     # batches in the pool
        pool_idx = [rank * nbatch * per_step + i for i in range(nbatch * per_step)]
        shards = [np.arange(r * per_step, (r + 1) * per_step, dtype=np.int64) for r in range(world)]

    def make(gi):      # global read index -> read; every 5th read noisy (margin 3), odd reads reverse-complemented
        if stub:       # (a few blocks of zeros, some of them too short to decode: error codes travel through the gather too)
            return dict(post=np.zeros((3 + (gi * 7919) % 57, 40), np.float32), rc=bool(gi & 1))
        return synth.make_read(a.mem_conv, a.rate, a.msg_len, seed=1000 + gi, rc=bool(gi & 1),
                               margin=3.0 if gi % 5 == 0 else 6.0)

    reads = [make(gi) for gi in pool_idx]
    batches = []
    for b in range(nbatch):
        rs = reads[b * per_step:(b + 1) * per_step]
        flat, off = pkg.Decoder.pack([x["post"] for x in rs])
        batches.append(dict(flat=flat, off=off, rc=np.array([x["rc"] for x in rs], np.uint8), reads=rs))
    if a.resident:
        for bt in batches:
            bt["dev"], _ = dec.upload([x["post"] for x in bt["reads"]])

    # ---- CPU leg: set up here, run AFTER the timed region (nothing but the decoder's host thread works during it) ----
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import oracle as O
        import tempfile
        cpu = dict(O=O, tmp=tempfile.mkdtemp(prefix="lva_bench_"))
        cpu["i1"] = 5 if len(batches[0]["reads"]) > 5 else 0      # read 5 of the pool: reverse complement, noisy (margin 3)

    def run(bt):
        if a.resident:
            return dec.decode_resident(bt["dev"], bt["off"], bt["rc"])
        return dec.decode_packed(bt["flat"], bt["off"], bt["rc"])

    def barrier():
        if dist is not None:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            dist.barrier()

    step_no = 0
    outs = {}
    dec_closed = False
    for _ in range(a.warmup):
        outs[step_no % nbatch] = run(batches[step_no % nbatch]); step_no += 1
    dec.set_launch_events(not a.no_launch_events)
    barrier()
    acc = dict(span_ms=0.0, dom_ms=0.0, pair_ms=0.0, alg=0.0, moved=0.0, launches=0, tl=0, read_steps=0, fix=0, h2d_ms=0.0, h2d_b=0,
               total_ms=0.0)
    fixr = [0, 0, 0, 0]
    step_ms = []                                    # this rank's wall time of every timed step (drift over a long run shows here)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        b = step_no % nbatch
        ts = time.perf_counter()
        outs[b] = run(batches[b]); step_no += 1     # returns after the stream is drained and the lists are on the host
        step_ms.append(1e3 * (time.perf_counter() - ts))
        p = dec.profile()
        acc["span_ms"] += p["step_kernel_ms"]; acc["dom_ms"] += p["dominant_kernel_ms"]; acc["pair_ms"] += p["step_pair_ms"]
        acc["alg"] += p["algorithmic_bytes"]; acc["moved"] += p["working_bytes"]; acc["launches"] += p["step_launches"]; acc["tl"] += p["timed_launches"]
        acc["read_steps"] += p["read_steps"]; acc["fix"] += p["fixup_states"]
        acc["h2d_ms"] += p["h2d_ms"]; acc["h2d_b"] += p["h2d_bytes"]; acc["total_ms"] += p["total_ms"]
        fixr = [x + y for x, y in zip(fixr, p["fixup_reason"])]
    barrier()
    dt = time.perf_counter() - t0
    dt_own = dt                                     # this rank's clock (dt becomes the maximum over the ranks below)
    last_b = (step_no - 1) % nbatch
    gathered = None
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev or "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # per-rank figures, so that a bad scaling point names its rank: reads/s of the rank's own clock, its dominant kernel's
    # mean launch time, its mean number of active read slots per launch (one all_gather of three doubles, after the timed region)
    per_rank = None
    if dist is not None:
        import torch
        mine3 = torch.tensor([per_step * a.steps / dt_own,
                              (acc["dom_ms"] if acc["tl"] > 0 else acc["span_ms"]) / max(acc["launches"], 1),
                              acc["read_steps"] / max(acc["launches"], 1)], dtype=torch.float64, device=coll_dev or "cpu")
        allr = [torch.zeros_like(mine3) for _ in range(world)]
        dist.all_gather(allr, mine3)
        per_rank = [dict(rank=r, reads_s=float(v[0]), avg_launch_ms=float(v[1]), mean_active_slots=float(v[2]))
                    for r, v in enumerate(x.cpu() for x in allr)]
    # the path's only exchange step: the decoded lists of the last step, gathered on rank 0 in global read order
    gathered = sharding.gather_results(outs[last_b], shards, a.list_size, a.msg_len, dist=dist, device=coll_dev)

    if rank == 0:
        n_global = sum(len(s) for s in shards)
        assert gathered is not None and len(gathered) == n_global and all(g is not None for g in gathered)
        if a.dump_lists:
            c, m_, s_ = sharding.pack_results(gathered, a.list_size, a.msg_len)
            np.savez(a.dump_lists, counts=c, msgs=m_, scores=s_)
        total_reads = n_global * a.steps
        nblk_mean = float(np.mean([x["post"].shape[0] for x in reads]))
        distinct = min(len(pool_idx), per_step * (a.steps + a.warmup)) if not strong else per_step
        prof = dec.profile()
        res = {
            "metric": "reads decoded/sec (m=%d, r=%d/%d, L=%d, msg_len=%d)" % (a.mem_conv, a.rate, a.rate + 1, a.list_size, a.msg_len),
            "value": total_reads / dt, "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / max(a.steps, 1), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1] shape: mem_conv=%d rate=%d list_size=%d msg_len=%d max_deviation=%d; per step "
                                   "%d reads per GPU (mean nblk %.0f, fwd/rc mixed, 1 in 5 noisy) through %s, %d read slots; "
                                   "%d distinct reads per GPU"
                                   % (a.mem_conv, a.rate, a.list_size, a.msg_len, a.max_deviation, per_step, nblk_mean,
                                      "lva_decode_batch_device (posteriors resident in HBM)" if a.resident
                                      else "lva_decode_batch (host buffers: H2D inside the timed region)", slots, distinct),
                       "reads_per_step_per_gpu": per_step, "distinct_reads_per_gpu": distinct, "slots": slots,
                       "h2d": "excluded (resident)" if a.resident else "included",
                       "h2d_ms_per_step": acc["h2d_ms"] / max(a.steps, 1), "h2d_bytes_per_step": acc["h2d_b"] / max(a.steps, 1),
                       "kernel": prof["kernel"], "fixup_states": acc["fix"], "fixup_reason": fixr,
                       "gathered_lists": n_global, "mean_active_slots": acc["read_steps"] / max(acc["launches"], 1),
                       "dist_backend": dist.get_backend() if dist is not None else None, "world": world,
                       "per_rank": per_rank, "step_ms_rank0": [round(x, 1) for x in step_ms]},
            # the library that was measured: lva_version() carries a hash of its source files (csrc/Makefile); counter files
            # under profiles/ name the build they were taken on, and `traffic` below is only filled from a file of THIS build
            "library": {"version": L.lva_version().decode(), "build_id": build_id},
        }
        use_events = acc["tl"] > 0
        dom_ms = acc["dom_ms"] if use_events else acc["span_ms"]
        achieved = (acc["alg"] / 1e9) / (dom_ms / 1e3) if dom_ms > 0 else 0.0
        kname = {2: "lva_step_fast<%d,P>" % a.list_size if a.list_size in (2, 4, 8) else ("lva_step_acs<P>" if a.list_size == 1 else "lva_step_big<LL,P> / lva_step_big_rec<LL>"),
                 4: "lva_step_lazy<%d,P,true> | lva_step_lazy<%d,P,false> (the anchor-step and the odd-step instance: the slots are "
                    "phase-aligned, so even launches run the first over all slots and odd launches the second; avg_launch_ms is the "
                    "mean over both kinds of launch)" % (a.list_size, a.list_size),
                 3: "lva_step_wave", 1: "lva_step_exact"}.get(prof["kernel"], "?")
        # HBM bytes per launch from this round's PMC profile (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
        # passes, scripts/pmc_mem.sh), per read-step, scaled to this run's mean number of active slots per launch
        traffic, tsrc, limiter = None, None, None
        try:
            for name in ("r6_traffic.json", "r5_traffic.json", "r4_traffic.json", "r3_traffic.json", "r2_traffic.json", "r1_traffic.json"):
                pth = os.path.join(ROOT, "profiles", name)
                if os.path.exists(pth):
                    tj = json.load(open(pth))
                    if ((a.mem_conv, a.rate, a.list_size, a.msg_len, a.max_deviation) == (11, 5, 8, 180, 20) and acc["launches"]
                            and tj.get("kernel_mode", 2) == prof["kernel"]):
                        if tj.get("build_id") == build_id:
                            per_rs = (tj["fetch_correction"] * tj["fetch_size_kb_per_launch"] + tj["write_size_kb_per_launch"]) * 1024.0 / tj["slots"]
                            traffic = per_rs * (acc["read_steps"] / acc["launches"])
                            tsrc = "profiles/" + name + " (PMC passes over the same library build, scaled per read-step to this run's active slots)"
                            limiter = tj.get("limiter")
                        else:
                            tsrc = ("profiles/%s was measured on library build %s, this run is build %s: no traffic figure for this run"
                                    % (name, tj.get("build_id", "(unnamed, before round 5)"), build_id))
                    break
        except Exception:
            traffic = None
        moved = (acc["moved"] / 1e9) / (dom_ms / 1e3) if dom_ms > 0 else 0.0
        res["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
            # the same with the bytes of the band the kernels actually work on (lva_band_table's working band): what is MOVED per
            # second.  `frac` keeps SURVEY 8(d)'s definition (the reference's band) so that rounds stay comparable.
            "achieved_moved": moved, "frac_moved": moved / 8000.0,
            "working_band_bytes_per_launch": acc["moved"] / max(acc["launches"], 1),
            "traffic": traffic, "traffic_source": tsrc,
            # what the counters say holds the kernel back (the HBM roofline above stays the yardstick of the contract):
            # from the committed PMC profile of the same kernels, not measured in this run
            "limiter": limiter,
            "kernel": kname + " (one trellis step of every active read slot)",
            "basis": ("algorithmic bytes / sum of per-launch HIP-event times of the dominant kernel alone" if use_events
                      else "algorithmic bytes / HIP-event span first..last step launch"),
            "algorithmic_bytes_definition": "SURVEY 8(d): sum over time steps of the structurally reachable in-band states of the "
                                            "REFERENCE's band; since round 4 the kernels skip the cells of that band whose lists cannot reach the "
                                            "output (positions > t + 1 and < nstate_pos - nblk + t: about 5 % of them at this shape), so "
                                            "`achieved` counts bytes for those cells that are no longer moved -- `achieved_moved` / `frac_moved` "
                                            "count the working band's cells only",
            "launches": acc["launches"], "avg_launch_ms": dom_ms / max(acc["launches"], 1),
            "algorithmic_bytes_per_launch": acc["alg"] / max(acc["launches"], 1),
            "pair": {"kernel": kname + " + fix-up pass", "avg_launch_ms": acc["pair_ms"] / max(acc["tl"], 1),
                     "achieved": (acc["alg"] / 1e9) / (acc["pair_ms"] / 1e3) if acc["pair_ms"] > 0 else None} if use_events else None,
            "span": {"ms_per_launch": acc["span_ms"] / max(acc["launches"], 1),
                     "achieved": (acc["alg"] / 1e9) / (acc["span_ms"] / 1e3) if acc["span_ms"] > 0 else None,
                     "note": "first..last step launch incl. slot init / final gather launches"},
            "end_to_end_achieved": (acc["alg"] / 1e9) / dt,
        }
        # ---- every list of one timed batch against an independent exact kernel (mode 1: one thread per target, the
        #      reference merge verbatim, no fingerprints, no lazy messages), outside the timed region ----
        if not a.no_cross_check and prof["kernel"] != 1:
            t1 = time.time()
            # (a second decoder beside the first: both keep their read slots in HBM -- 2 x 44 GB at the benchmark shape)
            with pkg.Decoder(a.mem_conv, a.rate, a.msg_len, list_size=a.list_size, max_deviation=a.max_deviation,
                             device=devno, kernel=1) as dec1:
                bt = batches[last_b]
                k = min(a.cross_check_reads, len(bt["reads"])) if a.cross_check_reads > 0 else len(bt["reads"])
                want = dec1.decode([x["post"] for x in bt["reads"][:k]], rc=[x["rc"] for x in bt["reads"][:k]])

            def is_code(x):
                return isinstance(x, (int, np.integer))

            def same(g, w):
                if is_code(g) or is_code(w):
                    return is_code(g) and is_code(w) and int(g) == int(w)      # the same per-read error code on both sides
                return np.array_equal(g[0], w[0]) and np.array_equal(g[1].view(np.uint32), w[1].view(np.uint32))

            bad = [i for i, (g, w) in enumerate(zip(outs[last_b][:k], want)) if not same(g, w)]
            assert not bad, "lists of the timed batch differ from the exact kernel's for reads %r" % bad[:8]
            res["config"]["cross_checked_reads"] = len(want)
            res["config"]["cross_check"] = "%s %d lists + scores of the last timed batch == kernel mode 1 (lva_step_exact), %.1f s" % (
                "all" if k == len(bt["reads"]) else "the first", len(want), time.time() - t1)
        # ---- configs[3] and configs[4] of BASELINE.json, one short driver-timed run each at the decoder's default slot count (the
        #      headline decoder is closed first: every decoder keeps its read slots resident in HBM) ----
        if (world == 1 and not stub and not a.no_extra_configs
                and (a.mem_conv, a.rate, a.list_size, a.msg_len, a.max_deviation) == (M, RATE, LIST, MSG_LEN, MAXDEV)):
            if a.resident:
                for bt in batches:
                    dec.free(bt["dev"])
                a.resident = False
            dec.close()
            dec_closed = True
            res["extra_configs"] = [
                run_extra_config(pkg, synth, np, "configs[3]", 14, 7, 8, ["m14_r7_L8", "m14_r7_L8_rc"], a.extra_reads, devno, build_id),
                run_extra_config(pkg, synth, np, "configs[4]", 11, 5, 64, ["m11_r5_L64", "m11_r5_L64_rc"], a.extra_reads, devno, build_id)]
            # line traffic of the big-list kernel per launch (committed counters of the SAME library build, per read-step, scaled to
            # this run's active slots): FETCH_SIZE x 2 + WRITE_SIZE, and TCC_MISS x 128 B
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "r6_traffic.json")))
                e4, k4 = res["extra_configs"][1], tj["kernels"]["big_rec_64"]
                if tj.get("build_id") == build_id:
                    e4["traffic"] = k4["fetch_x2_plus_write"] / k4["slots"] * e4["mean_active_slots"]
                    e4["traffic_tcc_miss_x128"] = k4["tcc_miss_x128"] / k4["slots"] * e4["mean_active_slots"]
                    e4["traffic_over_algorithmic"] = [k4["ratio_fetch_x2_plus_write"], k4["ratio_tcc_miss_x128"]]
                    e4["traffic_source"] = "profiles/r6_traffic.json (PMC passes over the same library build)"
                else:
                    e4["traffic"] = None
                    e4["traffic_source"] = "profiles/r6_traffic.json was measured on build %s, this run is build %s" % (tj.get("build_id"), build_id)
            except Exception:
                pass
        if cpu is not None:
            O = cpu["O"]
            # the -t 1 sample runs beside the -t N samples: N is capped so that N + 1 threads never exceed the host's CPUs
            ncpu = os.cpu_count() or 1
            cores = a.cpu_threads or max(1, min(ncpu - 1, 16))
            checked = 0
            if O.have_ref():
                r1 = batches[0]["reads"][cpu["i1"]]
                cpu["job1"] = RefJob(O, a, r1["post"], r1["rc"], 1, cpu["tmp"])     # -t 1, beside the -t N sample below
                # part 2: all cores (OpenMP -t), reads 0 (fwd, noisy), 1 (rc), 2 (fwd) of the pool, one after another
                idxs = list(range(min(a.cpu_reads, len(batches[0]["reads"]))))
                t_multi = 0.0
                for i in idxs:
                    r = batches[0]["reads"][i]
                    job = RefJob(O, a, r["post"], r["rc"], cores, cpu["tmp"])
                    rc_, lines = job.wait()
                    t_multi += job.wall
                    assert rc_ == 0 and lines == as_lines(outs[0][i]), "GPU list of read %d differs from the reference" % i
                    checked += 1
                res["cpu_baseline"] = dict(value=len(idxs) / t_multi, unit="reads/s", cores=cores, kind="reference",
                                           sample="%d reads of the benchmark pool (fwd noisy, rc, fwd), -t %d, one after another; wall %.1f s"
                                                  % (len(idxs), cores, t_multi), host_cpus=ncpu,
                                           concurrent="one -t 1 reference job ran beside these samples (%d + 1 busy threads on %d CPUs)" % (cores, ncpu))
                rc_, lines = cpu["job1"].wait(timeout=900)
                i1 = cpu["i1"]
                assert rc_ == 0 and lines == as_lines(outs[0][i1]), "GPU list of read %d differs from the reference" % i1
                checked += 1
                res["cpu_baseline_single_thread"] = dict(
                    value=1.0 / cpu["job1"].wall, unit="reads/s", cores=1, kind="reference",
                    sample="1 read (rc, noisy, nblk=%d), -t 1, after the timed region (beside the -t N sample; run to run 85-135 s); wall %.1f s"
                           % (batches[0]["reads"][i1]["post"].shape[0], cpu["job1"].wall))
            else:
                # the reference binary did not travel: time the plain-C restatement instead, and say so loudly
                code = O.OracleCode(a.mem_conv, a.rate, a.msg_len, rc=bool(batches[0]["reads"][0]["rc"]))
                t1 = time.time()
                lst, _ = code.decode(batches[0]["reads"][0]["post"], a.list_size, a.max_deviation, num_threads=cores)
                wall = time.time() - t1
                assert ["".join(map(str, x)) for x in lst] == as_lines(outs[0][0]), "GPU list differs from the oracle"
                checked += 1
                res["cpu_baseline"] = dict(value=1.0 / wall, unit="reads/s", cores=cores, kind="port",
                                           warning="oracle/_ref/viterbi_nanopore.out is absent on this host: this is the C restatement, NOT the reference binary",
                                           sample="1 read, %d threads; wall %.1f s" % (cores, wall))
            res["config"]["reference_checked_reads"] = checked
        print(json.dumps(res), flush=True)
    if not dec_closed:
        if a.resident:
            for bt in batches:
                dec.free(bt["dev"])
        dec.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
