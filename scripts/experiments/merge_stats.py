#!/usr/bin/env python3
"""Merge statistics of the list decoder (kernel design input): builds scripts/experiments/merge_stats_oracle.c -- an INSTRUMENTED
COPY of the CPU oracle (the pinned oracle/lva_oracle.c itself carries no bookkeeping) -- with -DLVA_ORACLE_STATS into /tmp
and prints, for one synthetic read, how many heap pops a target
needs and how deep each candidate list is consumed.

    python scripts/merge_stats.py M RATE MSG_LEN L [margin] [threads]
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nanopore_dna_storage_amd import synth  # noqa: E402

m, r, ml, L = (int(x) for x in sys.argv[1:5])
margin = float(sys.argv[5]) if len(sys.argv) > 5 else 4.0
thr = int(sys.argv[6]) if len(sys.argv) > 6 else 4
so = "/tmp/liblva_oracle_stats.so"
subprocess.run(["gcc", "-O2", "-std=c11", "-fopenmp", "-fPIC", "-shared", "-fno-fast-math", "-ffp-contract=off",
                "-DLVA_ORACLE_STATS", "-I", os.path.join(ROOT, "oracle"), "-o", so, os.path.join(ROOT, "scripts", "experiments", "merge_stats_oracle.c"),
                os.path.join(ROOT, "oracle", "basecall_oracle.c"), "-lm"], check=True)
Lb = ctypes.CDLL(so)
Lb.lva_oracle_code_new.restype = ctypes.c_void_p
Lb.lva_oracle_code_new.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_char_p, ctypes.c_uint32,
                                   ctypes.POINTER(ctypes.c_int)]
Lb.lva_oracle_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int,
                                 ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
st = ctypes.c_int(0)
h = Lb.lva_oracle_code_new(m, r, ml, 0, None, 0, ctypes.byref(st))
rd = synth.make_read(m, r, ml, seed=4242, margin=margin)
post = np.ascontiguousarray(rd["post"])
msgs = np.zeros((L, ml), np.uint8); sc = np.zeros(L, np.float32); cnt = ctypes.c_uint32(0)
Lb.lva_oracle_stats_reset()
Lb.lva_oracle_decode(h, post.ctypes.data, post.shape[0], L, 20, thr, 0, 1, msgs.ctypes.data, sc.ctypes.data, ctypes.byref(cnt))


def arr(name, n):
    return np.array((ctypes.c_uint64 * n).in_dll(Lb, name), dtype=np.float64)


T = float(ctypes.c_uint64.in_dll(Lb, "lva_stats_targets").value)
pops = arr("lva_stats_pops", 8 * 65 + 1); acc = arr("lva_stats_accepted", 66)
stay = arr("lva_stats_stay_depth", 66); src = arr("lva_stats_src_depth", 66)
below = arr("lva_stats_src_pops_below", 66); within = arr("lva_stats_targets_src_within", 66)
sp = float(ctypes.c_uint64.in_dll(Lb, "lva_stats_src_pops_total").value)
yp = float(ctypes.c_uint64.in_dll(Lb, "lva_stats_stay_pops_total").value)
print("m=%d r=%d msg_len=%d L=%d margin=%.1f nblk=%d: targets with a finite head %.3g" % (m, r, ml, L, margin, post.shape[0], T))
print("mean pops %.2f  mean accepted %.2f  full lists %.1f%%" % ((pops * np.arange(len(pops))).sum() / T,
                                                               (acc * np.arange(66)).sum() / T, 100 * acc[L] / T))
print("pops from the stay list %.1f%%, from source lists %.1f%%" % (100 * yp / (yp + sp), 100 * sp / (yp + sp)))
q = np.cumsum(stay) / T
print("stay-list depth: mean %.2f  quantiles 50/90/99: %d/%d/%d" % ((stay * np.arange(66)).sum() / T,
      np.searchsorted(q, .5), np.searchsorted(q, .9), np.searchsorted(q, .99)))
rk = np.array((ctypes.c_uint64 * (8 * 66)).in_dll(Lb, "lva_stats_src_rank_depth"), dtype=np.float64).reshape(8, 66)
for a in range(7):
    if rk[a].sum():
        print("  source list rank %d: mean depth %.2f, zero %.1f%%" % (a, (rk[a] * np.arange(66)).sum() / rk[a].sum(), 100 * rk[a][0] / rk[a].sum()))
for K in (1, 2, 4, 8, 12, 16, 24, 32):
    if K <= L:
        print("K=%2d: %.1f%% of source-list pops have index < K; %.1f%% of targets never go beyond K in any source list"
              % (K, 100 * below[K] / max(sp, 1), 100 * within[K] / T))

# pops per target, the iterations a 64-lane wavefront runs for them, and what skipping source-vs-source duplicates would change
tot = pops.sum()
print("pops per target: " + " ".join("%d:%.3f" % (i, pops[i] / tot) for i in range(min(len(pops), L + 12)) if pops[i] / tot >= 0.0005))


def wave_max(hist, n=64):
    cum = np.cumsum(hist) / hist.sum()
    pm = np.diff(np.concatenate([[0.0], cum ** n]))
    return (pm * np.arange(len(pm))).sum()


dk = arr("lva_stats_dup_kind", 3); noss = arr("lva_stats_pops_noss", 8 * 65 + 1)
print("mean pops %.2f, expected maximum over 64 independent targets %.2f" % ((pops * np.arange(len(pops))).sum() / tot, wave_max(pops)))
print("duplicate pops per target %.2f: source vs source %.1f%%, popped from the stay list %.1f%%, source matching a stay entry %.1f%%" % (
    dk.sum() / tot, 100 * dk[0] / dk.sum(), 100 * dk[1] / dk.sum(), 100 * dk[2] / dk.sum()))
print("without the source-vs-source duplicates: mean pops %.2f, expected maximum over 64 targets %.2f" % (
    (noss * np.arange(len(noss))).sum() / noss.sum(), wave_max(noss)))
dl = arr("lva_stats_dup_lineage", 3); nolin = arr("lva_stats_pops_nolin", 8 * 65 + 1)
print("duplicates that are the same path seen twice (the stay entry was made from that very source entry): %.2f%%; different paths, same message: %.2f%%" % (
    100 * dl[0] / dl[:2].sum(), 100 * dl[1] / dl[:2].sum()))
print("popped candidates whose identity named an accepted entry as twin although the messages differ: %d (must be 0)" % dl[2])
print("if same-path duplicates were dropped without a pop: mean pops %.2f, expected maximum over 64 targets %.2f" % (
    (nolin * np.arange(len(nolin))).sum() / nolin.sum(), wave_max(nolin)))
tw = arr("lva_stats_tw", 3); notw = arr("lva_stats_pops_notw", 8 * 65 + 1)
print("with one twin word per entry + one forward map per list and step (what a GPU lane would have): %.2f%% of the duplicate pops named, "
      "%d named wrongly (must be 0)" % (100 * tw[0] / tw[2], tw[1]))
print("without those pops: mean pops %.2f, expected maximum over 64 targets %.2f" % ((notw * np.arange(len(notw))).sum() / notw.sum(), wave_max(notw)))
ti = arr("lva_stats_ties", 4)
print("targets with a tie (equal score still in the heap at a pop): %.3g = %.2f%% of the targets; every tie between two copies of ONE message in %.1f%% of them "
      "(tie events %.3g, benign %.1f%%)" % (ti[0], 100 * ti[0] / T, 100 * ti[1] / max(ti[0], 1), ti[2], 100 * ti[3] / max(ti[2], 1)))
