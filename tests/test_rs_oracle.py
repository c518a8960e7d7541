"""SURVEY 8(f) row N4 on the CPU: the numpy oracle of the Reed-Solomon outer code (oracle/rs_oracle.py) against
  * tests/golden/rs_cases.json -- outputs of the reference's own MainEncoder / MainDecoder (Python + C++), and
  * the compiled reference codec program (oracle/_ref/schifra_RS_16bit_fileio_<fec>.out), block by block,
    including words the decoder must give up on.
This is what pins the oracle; the GPU path is compared with the oracle in tests/test_gpu_rs.py."""
import json
import os

import numpy as np
import pytest

from oracle import rs_oracle as R

HERE = os.path.dirname(os.path.abspath(__file__))


def cases():
    return json.load(open(os.path.join(HERE, "golden", "rs_cases.json")))["cases"]


@pytest.mark.parametrize("i", range(10))
def test_oracle_matches_reference_main_encoder_decoder(i):
    c = cases()[i]
    reads = [bytes.fromhex(x) for x in c["reads"]]
    assert [e.hex() for e in R.MainEncoder(reads, c["redundancy"])] == c["encoded"]
    rx = [[j, bytes.fromhex(p)] for j, p in c["received"]]
    dec = R.MainDecoder(rx, c["redundancy"], c["total"])
    assert [d.hex() for d in dec] == c["decoded"]
    assert (dec == reads) == c["recovered"]


def test_golden_set_covers_failures():
    cs = cases()
    assert sum(not c["recovered"] for c in cs) >= 3 and sum(c["recovered"] for c in cs) >= 5
    # an undecodable column comes back as ASCII '0' bytes (RSCode_16bit_fileio.py:122-123)
    assert any(all(d[:4] == "3030" for d in c["decoded"]) for c in cs if not c["recovered"])


@pytest.mark.parametrize("fec", [2, 6, 20])
def test_oracle_block_decoder_matches_reference_program(fec):
    if not R.have_ref(fec):
        pytest.skip("oracle/_ref RS codec for fec %d not built" % fec)
    rng = np.random.default_rng(100 + fec)
    N = R.N
    seen = {True: 0, False: 0}
    for trial in range(10):
        n_data = int(rng.integers(1, 30))
        n_total = n_data + fec
        pad = N - n_total
        full = np.concatenate([np.full(pad, R.PAD), rng.integers(0, 65536, size=n_data)])
        enc = R.encode_block(full, fec)
        rc, ref_enc = R.ref_codec(fec, full, encode=True)
        assert rc == 0 and np.array_equal(ref_enc, enc)
        S = min(int(rng.integers(0, fec + 2)) if trial % 3 else 0, n_total)
        E = int(rng.integers(0, fec // 2 + 3))
        er = sorted(rng.choice(n_total, size=S, replace=False).tolist())
        rx = enc.copy()
        rx[[pad + p for p in er]] = R.PAD
        anywhere = trial % 5 == 4                     # errors may also hit the padding that is not transmitted
        for p in rng.choice(N if anywhere else n_total, size=min(E, n_total), replace=False):
            rx[int(p) if anywhere else pad + int(p)] ^= int(rng.integers(1, 65536))
        erl = [pad + p for p in er]
        rc, ref_dec = R.ref_codec(fec, rx, erasures=erl)
        ok, blk = R.decode_block(rx, fec, erl)
        assert (rc == 0) == ok, (fec, trial, S, E)
        if ok:
            assert np.array_equal(blk[:N - fec], ref_dec), (fec, trial, S, E)
        seen[ok] += 1
    assert seen[True] >= 2 and seen[False] >= 1


def test_consensus_rule():
    """decode_RS_from_decoded_lists.py:37-51: most frequent payload per index; among equal counts the one that got there first"""
    a, b, c = b"aa", b"bb", b"cc"
    got = R.consensus([(3, a), (1, c), (3, b), (3, b), (3, a), (1, c), (7, a)])
    assert got == [[3, b], [1, c], [7, a]]
    assert R.consensus([(0, a), (0, b)]) == [[0, a]]
    assert R.consensus([(0, a), (0, b), (0, b), (0, a), (0, a)]) == [[0, a]]
