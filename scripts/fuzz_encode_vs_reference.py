#!/usr/bin/env python3
"""SURVEY 8(f) row N1 on random input: the product's encoder (lva_encode, host C++ in liblva_hip.so -- no GPU needed) against
`viterbi_nanopore.out -m encode` of the UNMODIFIED reference binary (oracle/_ref, build container only): every (mem_conv, rate) the
code has, message lengths 8..240 (msg_len + mem_conv < 256: beyond, the reference writes past its position table `:75` and
the product refuses), four random messages each (+ all-zero and all-one), oligos compared base for base; parameter sets
the reference refuses must be refused.

    python scripts/fuzz_encode_vs_reference.py SEED N
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanopore_dna_storage_amd import decoder  # noqa: E402
from oracle import oracle as O  # noqa: E402

CODES = [(6, 1), (6, 3), (6, 5), (8, 1), (8, 2), (8, 3), (8, 4), (8, 5), (11, 1), (11, 2), (11, 5), (14, 1), (14, 7)]


def main():
    seed, n = int(sys.argv[1]), int(sys.argv[2])
    assert O.have_ref(), "build oracle/_ref first (make -C oracle ref): needs /root/reference"
    rng = np.random.default_rng(seed)
    bad = refused = 0
    for i in range(n):
        m, r = CODES[int(rng.integers(len(CODES)))]
        if rng.random() < 0.05:                                   # parameter sets outside the table: both must refuse
            m, r = int(rng.choice([6, 7, 8, 11, 14])), int(rng.integers(1, 8))
        msg_len = int(rng.integers(8, 241))
        msgs = rng.integers(0, 2, size=(6, msg_len), dtype=np.uint8)
        msgs[0] = 0
        msgs[1] = 1
        try:
            want = O.ref_encode(m, r, msg_len, msgs)
        except RuntimeError:
            want = None
        try:
            got = ["".join("ACGT"[b] for b in row) for row in decoder.encode(m, r, msg_len, msgs)]
        except Exception:
            got = None
        ok = got == want
        refused += want is None
        bad += not ok
        if not ok or i % 50 == 0:
            print("%s m=%d r=%d msg_len=%d %s" % ("ok      " if ok else "MISMATCH", m, r, msg_len, "refused by both" if want is None and ok else ""), flush=True)
    print("checked %d bad %d (refused by both: %d)" % (n, bad, refused))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
