#!/usr/bin/env python3
"""Generate the golden fixtures of tests/golden/ by running the UNMODIFIED reference decoder.

Run in the build container only (needs oracle/_ref/viterbi_nanopore.out, which oracle/Makefile
compiles from /root/reference with the flags of the reference's install.sh:9):

    python tests/golden/make_golden.py [--only NAME ...] [--threads 8]

For every case it writes
    <name>.post       the posterior matrix, float32[nblk][40], exactly the file format the
                      reference reads with -i (flappie's --post-output-file layout)
    <name>.list       what `viterbi_nanopore.out -m decode ...` wrote to -o for that input
and records parameters, exit code and the transmitted message in manifest.json.
encode_cases.json holds `-m encode` inputs/outputs.  Fixtures are data: inputs + the reference's
outputs.  The synthetic inputs come from nanopore_dna_storage_amd.synth (SURVEY.md 8d recipe).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from nanopore_dna_storage_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

# name, mem_conv, rate, msg_len, list_size, max_deviation (None = flag absent), rc, margin, seed, extra
CASES = [
    ("m6_r1_L1_cfg1", 6, 1, 180, 1, 20, False, 6.0, 101, {}),
    ("m6_r1_L4_rc", 6, 1, 60, 4, 20, True, 6.0, 102, {}),
    ("m6_r3_L8", 6, 3, 60, 8, 10, False, 3.0, 103, {}),
    ("m6_r5_L8_rc", 6, 5, 180, 8, 20, True, 3.0, 104, {}),
    ("m6_r1_L8_unbanded", 6, 1, 60, 8, None, False, 3.0, 105, {}),
    ("m6_r1_L16", 6, 1, 60, 16, 20, False, 3.0, 106, {}),
    ("m6_r1_L8_ties", 6, 1, 60, 8, 20, False, 3.0, 107, {"quantum": 0.25}),
    ("m6_r5_L8_ties_rc", 6, 5, 60, 8, 20, True, 3.0, 108, {"quantum": 0.25}),
    ("m6_r1_L4_sync", 6, 1, 60, 4, 20, False, 4.0, 109, {"sync_marker": "110", "sync_period": 9}),
    ("m6_r1_L4_sync_rc", 6, 1, 60, 4, 20, True, 4.0, 110, {"sync_marker": "110", "sync_period": 9}),
    ("m6_r1_L4_indel", 6, 1, 60, 4, 20, False, 4.0, 111, {"sub": 0.02, "dele": 0.02, "ins": 0.01}),
    ("m8_r1_L8", 8, 1, 100, 8, 20, False, 3.0, 201, {}),
    ("m8_r2_L2_rc", 8, 2, 100, 2, 20, True, 4.0, 202, {}),
    ("m8_r3_L8", 8, 3, 164, 8, 20, False, 3.0, 203, {}),
    ("m8_r4_L4", 8, 4, 100, 4, 20, False, 3.0, 204, {}),
    ("m8_r5_L8_rc", 8, 5, 180, 8, 20, True, 3.0, 205, {}),
    ("m8_r3_L8_edge0", 8, 3, 164, 8, 20, False, 3.0, 210, {}),
    ("m8_r3_L8_edge1", 8, 3, 164, 8, 20, True, 3.0, 211, {}),
    ("m8_r3_L8_edge2", 8, 3, 164, 8, 20, False, 2.5, 212, {}),
    ("m8_r3_L8_edge3", 8, 3, 164, 8, 20, True, 2.5, 213, {}),
    ("m8_r3_L8_edge4", 8, 3, 164, 8, 8, False, 3.0, 214, {}),
    ("m8_r3_L8_edge5", 8, 3, 164, 8, 8, True, 3.0, 215, {}),
    ("m11_r5_L1", 11, 5, 180, 1, 20, False, 6.0, 301, {}),
    ("m11_r5_L8_clean", 11, 5, 180, 8, 20, False, 6.0, 302, {}),
    ("m11_r5_L8_noisy_rc", 11, 5, 180, 8, 20, True, 3.0, 303, {}),
    ("m11_r5_L8_noisy", 11, 5, 180, 8, 20, False, 3.0, 304, {}),
    ("m11_r5_L8_clean_rc", 11, 5, 180, 8, 20, True, 6.0, 305, {}),
    ("m11_r1_L4_short", 11, 1, 40, 4, 20, False, 4.0, 306, {}),
    ("m11_r5_L64", 11, 5, 180, 64, 20, False, 6.0, 307, {}),
    ("m14_r7_L8", 14, 7, 180, 8, 20, False, 6.0, 401, {}),
    ("m14_r1_L2_short", 14, 1, 20, 2, 10, True, 4.0, 402, {}),
    # round 2: the expensive settings in the noisy / reverse-complement regime (stale band decides entries 2..L)
    ("m14_r7_L8_noisy", 14, 7, 180, 8, 20, False, 3.0, 403, {}),
    ("m14_r7_L8_rc", 14, 7, 180, 8, 20, True, 4.0, 404, {}),
    ("m14_r7_L4", 14, 7, 180, 4, 20, False, 3.0, 405, {}),          # supplement 5.2: list size 4 at m=14
    ("m11_r5_L64_noisy", 11, 5, 180, 64, 20, False, 3.0, 308, {}),
    ("m11_r5_L64_rc", 11, 5, 180, 64, 20, True, 4.0, 309, {}),
    ("m8_r3_L64", 8, 3, 164, 64, 20, False, 3.0, 206, {}),          # supplement 5.2: list size 64 at m=8
    ("m8_r3_L64_rc", 8, 3, 164, 64, 20, True, 3.0, 207, {}),
    # round 4: non-finite posteriors (NaN, +inf: whatever the reference does with them), messages wider than 192 bits
    # (four message planes), a list longer than 64 (the thread-per-target kernel)
    ("m6_r1_L4_nan", 6, 1, 60, 4, 20, False, 4.0, 601, {"nan": 0.01}),
    ("m6_r1_L4_posinf_rc", 6, 1, 60, 4, 20, True, 4.0, 602, {"posinf": 0.01}),
    ("m6_r1_L8_nan_posinf", 6, 1, 60, 8, 20, False, 3.0, 603, {"nan": 0.005, "posinf": 0.005}),
    ("m6_r1_L1_nan_posinf", 6, 1, 60, 1, 20, False, 4.0, 604, {"nan": 0.01, "posinf": 0.01}),
    ("m6_r1_L16_nan", 6, 1, 60, 16, 20, False, 3.0, 605, {"nan": 0.01}),
    ("m6_r1_L8_wide", 6, 1, 200, 8, 20, False, 3.0, 606, {}),
    ("m6_r5_L4_wide_ties_rc", 6, 5, 190, 4, 20, True, 3.0, 607, {"quantum": 0.5}),
    ("m6_r1_L100", 6, 1, 60, 100, 20, False, 3.0, 608, {}),
]

# decode invocations the reference refuses or aborts on: (name, args..., truncate post to n blocks)
ERROR_CASES = [
    ("err_short_post", 6, 1, 60, 2, 20, False, 6.0, 501, {"truncate": 30}),
]

ENCODE_CASES = [(6, 1, 180), (6, 3, 180), (6, 5, 180), (8, 1, 180), (8, 2, 180), (8, 4, 180), (8, 5, 180),
                (8, 3, 164), (11, 1, 180), (11, 2, 180), (11, 5, 180), (14, 1, 180), (14, 7, 180), (11, 5, 100)]
BAD_PARAMS = [(7, 1, 180), (6, 6, 180), (6, 2, 180), (11, 3, 180), (14, 5, 180)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--skip-existing", action="store_true")
    a = ap.parse_args()
    assert O.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    man_path = os.path.join(HERE, "manifest.json")
    manifest = json.load(open(man_path)) if os.path.exists(man_path) else {}

    for case in CASES + ERROR_CASES:
        name, m, r, msg_len, L, md, rc, margin, seed, extra = case
        if a.only and name not in a.only:
            continue
        if a.skip_existing and name in manifest and os.path.exists(os.path.join(HERE, name + ".list")):
            continue
        sync = {k: extra[k] for k in ("sync_marker", "sync_period") if k in extra}
        gen = {k: extra[k] for k in ("quantum", "sub", "dele", "ins") if k in extra}
        bad = {k: extra[k] for k in ("nan", "posinf") if k in extra}
        rd = synth.make_read(m, r, msg_len, seed, rc=rc, margin=margin, **gen)
        post = rd["post"]
        if "nan" in extra or "posinf" in extra:          # a fixed share of the posteriors replaced (seeded by the case)
            u = np.random.default_rng(seed).random(post.shape)
            post = post.copy()
            fn, fp = extra.get("nan", 0.0), extra.get("posinf", 0.0)
            post[u < fn] = np.nan
            post[(u >= fn) & (u < fn + fp)] = np.inf
        if "truncate" in extra:
            post = post[:extra["truncate"]]
        post.tofile(os.path.join(HERE, name + ".post"))
        t0 = time.time()
        code, lines = O.ref_decode(m, r, msg_len, post, L, md, rc=rc, num_threads=a.threads, **sync)
        dt = time.time() - t0
        with open(os.path.join(HERE, name + ".list"), "w") as f:
            for ln in lines:
                f.write(ln + "\n")
        truth = "".join(map(str, rd["msg"]))
        manifest[name] = dict(mem_conv=m, rate=r, msg_len=msg_len, list_size=L, max_deviation=md, rc=rc,
                              margin=margin, seed=seed, nblk=int(post.shape[0]), exit_code=code,
                              n_lines=len(lines), message=truth,
                              top_correct=bool(lines[:1] == [truth]), list_correct=bool(truth in lines),
                              ref_seconds=round(dt, 2), ref_threads=a.threads, **sync, **gen, **bad)
        print("%-22s nblk=%4d exit=%d lines=%d top=%s list=%s %.1fs" % (
            name, post.shape[0], code, len(lines), lines[:1] == [truth], truth in lines, dt), flush=True)
        json.dump(manifest, open(man_path, "w"), indent=1, sort_keys=True)

    if not a.only:
        enc = {"cases": [], "bad_params": []}
        rng = np.random.default_rng(7)
        for m, r, msg_len in ENCODE_CASES:
            msgs = rng.integers(0, 2, size=(3, msg_len), dtype=np.uint8)
            msgs[0] = 0
            msgs[1] = 1
            out = O.ref_encode(m, r, msg_len, msgs)
            enc["cases"].append(dict(mem_conv=m, rate=r, msg_len=msg_len,
                                     msgs=["".join(map(str, x)) for x in msgs], oligos=out))
        for m, r, msg_len in BAD_PARAMS:
            try:
                O.ref_encode(m, r, msg_len, [np.zeros(msg_len, np.uint8)])
                ok = True
            except RuntimeError:
                ok = False
            enc["bad_params"].append(dict(mem_conv=m, rate=r, msg_len=msg_len, accepted=ok))
        json.dump(enc, open(os.path.join(HERE, "encode_cases.json"), "w"), indent=1)
        print("encode cases written")


if __name__ == "__main__":
    main()
