"""SURVEY 8(f) row N4 on the GPU: the Reed-Solomon outer code (csrc/rs_kernels.hip through lva_rs_decode /
lva_rs_encode and the Python mirror rs_code.py) against the reference's own outputs (tests/golden/rs_cases.json),
against the numpy oracle on seeded random words -- decodable, undecodable and miscorrected alike -- and through the
encode -> erase / corrupt -> decode round trip at the size of the paper's experiments."""
import json
import os

import numpy as np
import pytest

from nanopore_dna_storage_amd import helper, rs_code
from oracle import rs_oracle as R

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_against_reference_vectors():
    for c in json.load(open(os.path.join(HERE, "golden", "rs_cases.json")))["cases"]:
        reads = [bytes.fromhex(x) for x in c["reads"]]
        assert [e.hex() for e in rs_code.MainEncoder(reads, c["redundancy"])] == c["encoded"]
        rx = [[j, bytes.fromhex(p)] for j, p in c["received"]]
        dec, ok = rs_code.MainDecoder(rx, c["redundancy"], c["total"], return_ok=True)
        assert [d.hex() for d in dec] == c["decoded"]
        assert (dec == reads) == c["recovered"]


@pytest.mark.parametrize("seed", range(6))
def test_random_words_against_oracle(seed):
    rng = np.random.default_rng(900 + seed)
    red = int(rng.choice([2, 5, 8, 16, 33]))
    nd = int(rng.integers(1, 60))
    spr = int(rng.integers(1, 6))
    total = nd + red
    reads = [bytes(rng.integers(0, 256, size=2 * spr, dtype=np.uint8)) for _ in range(nd)]
    enc = rs_code.MainEncoder(reads, red)
    assert enc == R.MainEncoder(reads, red) and enc[:nd] == reads
    for trial in range(4):
        n_er = int(rng.integers(0, red + 3)) if trial else 0
        n_er = min(n_er, total - 1)
        erased = set(rng.choice(total, size=n_er, replace=False).tolist())
        rx = [[i, enc[i]] for i in range(total) if i not in erased]
        for j in rng.choice(len(rx), size=min(int(rng.integers(0, red // 2 + 3)), len(rx)), replace=False):
            b = bytearray(rx[j][1])
            b[int(rng.integers(len(b)))] ^= int(rng.integers(1, 256))
            rx[j][1] = bytes(b)
        got = rs_code.MainDecoder(rx, red, total)
        want = R.MainDecoder(rx, red, total)
        assert got == want, (seed, trial, red, nd, spr, n_er)


def test_round_trip_at_experiment_size():
    """exp. 7 of the paper: 18 bytes = 9 symbols per oligo, 564 data + 169 parity oligos (encode_experiments.py, RS 0.3):
    100 oligos missing and 30 with a wrong payload (100 + 2*30 <= 169) decode; 40 wrong ones do not."""
    rng = np.random.default_rng(7)
    nd, red, spr = 564, 169, 9
    reads = [bytes(rng.integers(0, 256, size=2 * spr, dtype=np.uint8)) for _ in range(nd)]
    enc = rs_code.MainEncoder(reads, red)
    total = nd + red
    for n_bad, good in ((30, True), (40, False)):
        erased = set(rng.choice(total, size=100, replace=False).tolist())
        rx = [[i, enc[i]] for i in range(total) if i not in erased]
        for j in rng.choice(len(rx), size=n_bad, replace=False):
            rx[j][1] = bytes(rng.integers(0, 256, size=2 * spr, dtype=np.uint8))
        dec, ok = rs_code.MainDecoder(rx, red, total, return_ok=True)
        assert (dec == reads) == good and bool(ok.all()) == good
        if not good:
            assert dec == R.MainDecoder(rx, red, total)      # same give-ups, same fill bytes


def test_consensus_and_list_chain():
    """decode_RS_from_decoded_lists.py:30-55: lists -> CRC/index filter -> per-index majority -> RS decode"""
    rng = np.random.default_rng(11)
    bpo, nd, red = 4, 20, 8
    payloads = [bytes(rng.integers(0, 256, size=bpo, dtype=np.uint8)) for _ in range(nd)]
    enc = rs_code.MainEncoder(payloads, red)
    total = nd + red
    msgs = [helper.attach_index_crc(i, enc[i]) for i in range(total)]
    lists = []
    for i in range(total):
        if i % 7 == 3:
            continue                                          # oligo never read
        for rep in range(3):
            wrong = "".join(rng.choice(list("01"), size=len(msgs[i])))
            lists.append([wrong, msgs[i]] if rep else [msgs[i], wrong])
    order = rng.permutation(len(lists))
    data, n_ok = rs_code.decode_from_lists([lists[k] for k in order], bpo, red, total)
    assert n_ok == len(lists) and data == b"".join(payloads)
    assert rs_code.consensus([(3, b"a"), (3, b"b"), (3, b"b"), (3, b"a"), (1, b"c")]) == R.consensus([(3, b"a"), (3, b"b"), (3, b"b"), (3, b"a"), (1, b"c")])


def test_argument_errors():
    import ctypes
    from nanopore_dna_storage_amd._lib import load_library
    L = load_library()
    sym = np.zeros((1, 10), np.uint16)
    out = np.zeros((1, 6), np.uint16)
    er = np.array([1, 1], np.int32)
    assert L.lva_rs_decode(0, sym.ctypes.data, 1, 10, 4, er.ctypes.data, 2, 0x3030, 0x3030, out.ctypes.data, None) == -10   # duplicate erasure
    assert L.lva_rs_decode(0, sym.ctypes.data, 1, 10, 0, None, 0, 0x3030, 0x3030, out.ctypes.data, None) == -10          # no redundancy
    assert L.lva_rs_decode(0, sym.ctypes.data, 1, 10, 5000, None, 0, 0x3030, 0x3030, out.ctypes.data, None) == -10       # beyond the LDS budget
    assert L.lva_rs_decode(0, sym.ctypes.data, 0, 10, 4, None, 0, 0x3030, 0x3030, out.ctypes.data, None) == 0            # empty batch
