#!/usr/bin/env python3
"""Checks the rule by which the lazy kernels call a duplicate PROVEN without comparing messages on the CPU:
builds scripts/experiments/proof_check_oracle.c (an instrumented copy of the oracle) into /tmp, decodes synthetic reads and prints
  pairs      (stay entry, source entry) pairs of all targets on which the rule fires -- whatever their fingerprints
  violations of those, pairs whose messages differ            -- MUST BE 0
  dups       duplicate rejections of the reference merge; how many pair a stay entry with a source entry; how many the rule covers

    python scripts/proof_check.py M RATE MSG_LEN L [n_reads] [max_deviation]
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
n = int(sys.argv[5]) if len(sys.argv) > 5 else 3
md = int(sys.argv[6]) if len(sys.argv) > 6 else 20
so = "/tmp/liblva_proof_check.so"
subprocess.run(["gcc", "-O2", "-std=c11", "-fopenmp", "-fPIC", "-shared", "-fno-fast-math", "-ffp-contract=off", "-I", os.path.join(ROOT, "oracle"),
                "-o", so, os.path.join(ROOT, "scripts", "experiments", "proof_check_oracle.c"), os.path.join(ROOT, "oracle", "basecall_oracle.c"), "-lm"], check=True)
Lb = ctypes.CDLL(so)
Lb.lva_oracle_code_new.restype = ctypes.c_void_p
Lb.lva_oracle_code_new.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_char_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_int)]
Lb.lva_oracle_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int,
                                 ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
tot = np.zeros(6, np.uint64)
for i in range(n):
    rc = bool(i & 1)
    st = ctypes.c_int(0)
    h = Lb.lva_oracle_code_new(m, r, ml, int(rc), None, 0, ctypes.byref(st))
    kw = dict(margin=3.0) if i % 2 == 0 else dict(margin=5.0, sub=0.01, dele=0.02, ins=0.01)      # noisy reads and reads with indels (band edge)
    rd = synth.make_read(m, r, ml, seed=9100 + i, rc=rc, **kw)
    post = np.ascontiguousarray(rd["post"])
    msgs = np.zeros((L, ml), np.uint8); sc = np.zeros(L, np.float32); cnt = ctypes.c_uint32(0)
    Lb.lva_proof_check_reset()
    Lb.lva_oracle_decode(h, post.ctypes.data, post.shape[0], L, md, 8, 0, 1, msgs.ctypes.data, sc.ctypes.data, ctypes.byref(cnt))
    o = (ctypes.c_ulonglong * 6)()
    Lb.lva_proof_check_get(o)
    v = np.array(list(o), np.uint64)
    tot += v
    print("read %d (%s): pairs %d violations %d | dups %d, stay-source %d (%.1f %%), proven %d (%.1f %% of all dups), stale rows skipped %d"
          % (i, "rc" if rc else "fwd", v[0], v[1], v[2], v[4], 100.0 * v[4] / max(v[2], 1), v[3], 100.0 * v[3] / max(v[2], 1), v[5]))
print("TOTAL pairs %d VIOLATIONS %d (must be 0) | dups %d proven %.1f %%" % (tot[0], tot[1], tot[2], 100.0 * tot[3] / max(tot[2], 1)))
sys.exit(1 if tot[1] else 0)
