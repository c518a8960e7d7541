#!/usr/bin/env python3
"""Golden vectors for SURVEY 8(f) row N4 (Reed-Solomon outer code + consensus), produced by running the REFERENCE:
RSCode_schifra/RSCode_16bit_fileio.py (MainEncoder / MainDecoder, imported from /root/reference) driving the
reference's own C++ codec, which it compiles itself for every codeword.  Build container only:

    python tests/golden/make_rs_golden.py        -> tests/golden/rs_cases.json

The module writes its temporary files and the compiled program next to itself (REPO_PATH); the reference tree
must not be written to, so REPO_PATH is pointed at a scratch directory that holds SYMLINKS to the reference's
schifra_*.hpp / schifra_RS_16bit_fileio.cpp (nothing is copied).  Inputs and outputs are committed as hex strings."""
import glob
import json
import os
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/RSCode_schifra"


def main():
    sys.path.insert(0, REF)
    import RSCode_16bit_fileio as RS
    scratch = tempfile.mkdtemp(prefix="rsgold_")
    for f in glob.glob(os.path.join(REF, "schifra_*")):
        os.symlink(f, os.path.join(scratch, os.path.basename(f)))
    RS.REPO_PATH = scratch + "/"
    os.chdir(scratch)
    rng = random.Random(77)
    cases = []
    # (data reads, symbols per read, redundancy, erased reads, reads with errors)
    # With no erased read the reference removes a file it never created (RSCode_16bit_fileio.py:130-131) and raises
    # after the first column, so every multi-column case has at least one erasure; the single-column case without
    # erasures runs with that file created beforehand, as a leftover of an earlier decode would be.
    plan = [(5, 3, 4, 2, 1), (8, 2, 6, 1, 2), (8, 2, 6, 6, 0), (12, 4, 10, 3, 3), (7, 1, 2, 1, 1),
            (20, 9, 6, 2, 2), (6, 2, 4, 5, 0), (9, 1, 4, 0, 2), (10, 2, 6, 1, 4), (30, 5, 20, 9, 5)]
    for nd, spr, red, n_er, n_err in plan:
        reads = [bytes(rng.randrange(256) for _ in range(2 * spr)) for _ in range(nd)]
        enc = RS.MainEncoder(reads, red)
        total = nd + red
        keep = list(range(total))
        erased = sorted(rng.sample(keep, n_er))
        rx = [[i, enc[i]] for i in keep if i not in erased]
        for j in rng.sample(range(len(rx)), n_err):
            b = bytearray(rx[j][1])
            if (nd, n_err) == (10, 4):            # beyond the code's capability in every column
                b = bytearray(rng.randrange(256) for _ in b)
            else:
                b[rng.randrange(len(b))] ^= rng.randrange(1, 256)
            rx[j][1] = bytes(b)
        rng.shuffle(rx)
        if n_er == 0:
            open(os.path.join(scratch, "trialerasurelocationfile.dat"), "wb").close()
        dec = RS.MainDecoder([list(x) for x in rx], red, total)
        cases.append(dict(data_reads=nd, symbols_per_read=spr, redundancy=red, total=total,
                          reads=[r.hex() for r in reads], encoded=[e.hex() for e in enc],
                          received=[[i, p.hex()] for i, p in rx], decoded=[d.hex() for d in dec],
                          recovered=bool(dec == reads)))
        print(nd, spr, red, n_er, n_err, "recovered" if dec == reads else "NOT recovered", flush=True)
    json.dump(dict(cases=cases), open(os.path.join(HERE, "rs_cases.json"), "w"), indent=1)
    print("wrote rs_cases.json")


if __name__ == "__main__":
    main()
