#!/usr/bin/env python3
"""Pin the CPU oracle beyond the 47 committed fixtures: random configurations through the UNMODIFIED reference binary
(oracle/_ref/viterbi_nanopore.out, built by oracle/Makefile from /root/reference -- build container only, no GPU) and through
oracle/lva_oracle.c, message lists compared line for line (the reference writes messages only; its exit code must agree too).

    python scripts/fuzz_oracle_vs_reference.py SEED N [--threads 8]

Draws: m in {6, 8} (and 11 for one case in 25), every rate the code has, list sizes 1..70, bands down to max_deviation 1 and
the unbanded default, both orientations, sync markers, substitutions / insertions / deletions, score ties (quantised posteriors),
NaN and +inf posteriors, reads truncated to barely more blocks than trellis positions.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanopore_dna_storage_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

RATES = {6: (1, 3, 5), 8: (1, 2, 3, 4, 5), 11: (1, 2, 5)}


def draw(rng):
    m = 11 if rng.random() < 0.04 else int(rng.choice([6, 6, 8]))
    r = int(rng.choice(RATES[m]))
    msg_len = int(rng.integers(12, 40)) if m == 11 else int(rng.integers(12, 200 if rng.random() < 0.2 else 70))
    L = int(rng.choice([1, 2, 3, 4, 5, 8, 8, 9, 12, 16, 24, 33, 64, 70])) if m < 11 else int(rng.choice([1, 2, 4, 8]))
    md = None if rng.random() < 0.1 else int(rng.choice([1, 2, 3, 6, 10, 20]))
    kw = {}
    if rng.random() < 0.25:
        kw["quantum"] = float(rng.choice([0.25, 0.5, 1.0]))
    if rng.random() < 0.2:
        kw.update(sub=0.02, dele=0.02, ins=0.01)
    sync = {}
    if rng.random() < 0.15 and msg_len >= 24:
        sync = dict(sync_marker="110", sync_period=int(rng.integers(7, 12)))
    return m, r, msg_len, L, md, bool(rng.random() < 0.5), float(rng.choice([2.0, 2.5, 3.0, 4.0, 6.0])), kw, sync


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("seed", type=int)
    ap.add_argument("n", type=int)
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    assert O.have_ref(), "build oracle/_ref first (make -C oracle ref): needs /root/reference"
    rng = np.random.default_rng(a.seed)
    bad = 0
    t0 = time.time()
    for i in range(a.n):
        while True:                                    # (some message lengths give an odd output length: the encoder refuses them)
            m, r, msg_len, L, md, rc, margin, kw, sync = draw(rng)
            try:
                rd = synth.make_read(m, r, msg_len, int(rng.integers(1 << 30)), rc=rc, margin=margin, **kw)
                break
            except Exception:
                continue
        post = rd["post"]
        what = ""
        u = rng.random()
        if u < 0.08:                                   # NaN / +inf posteriors: whatever the reference does with them
            v = rng.random(post.shape)
            post = post.copy()
            post[v < 0.01] = np.nan
            post[(v >= 0.01) & (v < 0.02)] = np.inf
            what = " nan/inf"
        elif u < 0.16:                                 # barely more blocks than trellis positions (:600-601) -- or fewer: the abort
            npos = O.OracleCode(m, r, msg_len, rc=rc, **sync).nstate_pos
            keep = npos + int(rng.integers(-1, 6))
            if 1 <= keep < post.shape[0]:
                post = post[:keep].copy()
                what = " nblk=npos%+d" % (keep - npos)
        code, lines = O.ref_decode(m, r, msg_len, post, L, md, rc=rc, num_threads=a.threads, **sync)
        try:
            msgs, _ = O.OracleCode(m, r, msg_len, rc=rc, **sync).decode(post, L, md, num_threads=a.threads)
            got = ["".join(map(str, x)) for x in msgs]
            ocode = 0
        except Exception as e:                         # the oracle's restatement of the reference's refusals
            got, ocode = [], 1
            what += " (oracle: %s)" % type(e).__name__
        ok = (got == lines) and ((code == 0) == (ocode == 0))
        bad += not ok
        print("%s m=%d r=%d msg_len=%d L=%d md=%s rc=%d margin=%.1f %s%s%s nblk=%d lines=%d exit=%d" % (
            "ok      " if ok else "MISMATCH", m, r, msg_len, L, md, rc, margin, kw or "", sync or "", what, post.shape[0], len(lines), code), flush=True)
    print("checked %d bad %d (%.0f s)" % (a.n, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
