"""SURVEY 8(f) row N3 on the GPU: bc_basecall / bc_search / bc_finalize through the C ABI against the
CPU oracle (integer and fp32 add/compare work: results must be identical)."""
import numpy as np
import pytest

import nanopore_dna_storage_amd as pkg
from nanopore_dna_storage_amd import synth

pytestmark = pytest.mark.gpu

SB, EB = "CACCTGTGCTGCGTCAGGCTGTGTC", "GCTGTCCGTTCCGCATTGACACGGC"


@pytest.fixture(scope="module")
def dec():
    with pkg.Decoder(8, 3, 44, list_size=4, max_deviation=20, max_slots=4) as d:
        yield d


def _check_basecall(oracle, dec, posts):
    got = dec.basecall(posts)
    assert len(got) == len(posts)
    for p, (bc, trans) in zip(posts, got):
        want_bc, want_trans, _, _ = oracle.basecall(p)
        assert bc == want_bc
        assert np.array_equal(trans, want_trans)


def test_basecall_synthetic_reads(oracle, dec):
    posts = [synth.make_read(8, 3, 44, 300 + i, rc=bool(i & 1), margin=[2.0, 3.0, 6.0][i % 3])["post"] for i in range(9)]
    _check_basecall(oracle, dec, posts)


def test_basecall_random_matrices(oracle, dec):
    rng = np.random.default_rng(7)
    posts = [rng.normal(0, 2, (int(n), 40)).astype(np.float32) for n in (1, 2, 3, 17, 64, 500, 1301)]
    _check_basecall(oracle, dec, posts)


def test_basecall_ties_and_minus_inf(oracle, dec):
    rng = np.random.default_rng(8)
    flat = np.zeros((40, 40), np.float32)                                    # every comparison is a draw
    grid = (np.round(rng.normal(0, 2, (300, 40)) * 2) / 2).astype(np.float32)   # many equal scores
    holes = rng.normal(0, 2, (200, 40)).astype(np.float32)
    holes[rng.random(holes.shape) < 0.2] = -np.inf
    allneg = np.full((30, 40), -np.inf, np.float32)
    _check_basecall(oracle, dec, [flat, grid, holes, allneg])


def test_basecall_resident_buffer(oracle, dec):
    posts = [synth.make_read(8, 3, 44, 330 + i, margin=4.0)["post"] for i in range(3)]
    dev, off = dec.upload(posts)
    try:
        got = dec.basecall_resident(dev, off)
    finally:
        dec.free(dev)
    for p, (bc, trans) in zip(posts, got):
        want_bc, want_trans, _, _ = oracle.basecall(p)
        assert bc == want_bc and np.array_equal(trans, want_trans)


def _rand_case(rng, plant):
    n = int(rng.integers(40, 400))
    s = "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    sb, eb = SB[:int(rng.integers(4, 26))], EB[:int(rng.integers(4, 26))]
    if plant and n > 2 * (len(sb) + len(eb)) + 30:
        p = int(rng.integers(0, 12))
        s = s[:p] + sb + s[p + len(sb):]
        q = n - len(eb) - int(rng.integers(2, 12))
        s = s[:q] + eb[:-3] + "TT" + s[q + len(eb) - 1:]
    trans = np.cumsum(rng.integers(1, 9, len(s))) + 1
    return s, trans, sb, eb


def test_find_barcode_matches_oracle(oracle, dec):
    rng = np.random.default_rng(11)
    for it in range(40):
        s, trans, sb, eb = _rand_case(rng, plant=it % 2 == 0)
        got = dec.find_barcode([s], [trans], sb, eb)[0]
        want = oracle.find_barcode_pos(s, trans, sb, eb)
        assert (got["start_pos"], got["end_pos"], got["dist_start"], got["dist_end"]) == want, (it, s, sb, eb)


def test_find_barcode_batch_and_edge_cases(oracle, dec):
    rng = np.random.default_rng(12)
    cases = [_rand_case(rng, True) for _ in range(5)]
    sb, eb = SB, EB
    seqs = [c[0] for c in cases] + ["ACGT", SB + EB, "A" * 200, (SB + EB) * 3]
    transs = [c[1] for c in cases] + [np.arange(1, len(s) + 1) * 3 for s in seqs[5:]]
    got = dec.find_barcode(seqs, transs, sb, eb)
    for s, t, g in zip(seqs, transs, got):
        want = oracle.find_barcode_pos(s, t, sb, eb)
        assert (g["start_pos"], g["end_pos"], g["dist_start"], g["dist_end"]) == want
    # 64-character barcodes (the limit) and identical start / end barcodes
    long_bc = (SB + EB + SB)[:64]
    s = "".join("ACGT"[i] for i in rng.integers(0, 4, 500))
    t = np.arange(1, 501) * 2
    g = dec.find_barcode([s], [t], long_bc, long_bc)[0]
    assert (g["start_pos"], g["end_pos"], g["dist_start"], g["dist_end"]) == oracle.find_barcode_pos(s, t, long_bc, long_bc)
    with pytest.raises(pkg.LvaError):
        dec.find_barcode([s], [t], "", EB)
    with pytest.raises(pkg.LvaError):
        dec.find_barcode([s], [t], "A" * 65, EB)


@pytest.mark.parametrize("margin,sub", [(7.0, 0.0), (3.0, 0.0), (6.0, 0.03)])
def test_locate_payload_matches_oracle(oracle, dec, margin, sub):
    reads = [synth.make_barcoded_read(8, 3, 44, 700 + i, SB, EB, rc=bool(i & 1), margin=margin, flank=(5, 14), sub=sub)
             for i in range(8)]
    reads.append(dict(post=np.random.default_rng(1).normal(0, 1, (30, 40)).astype(np.float32)))   # too short: failure
    got = dec.locate_payload([x["post"] for x in reads], SB, EB)
    for x, g in zip(reads, got):
        want = oracle.locate_payload(x["post"], SB, EB, 8 + 44 + 1)
        assert g == {k: want[k] for k in ("ok", "start_pos", "end_pos", "rc", "dist_start", "dist_end")}
    assert not got[-1]["ok"]
    assert sum(g["ok"] for g in got) >= 6


def test_post_to_list_chain_on_device(oracle, dec):
    """untruncated posteriors -> payload window -> decoded list, against the oracle's chain
    (generate_decoded_lists.py:68-89: locate, helper.truncate_post_file, decode with --rc)"""
    reads = [synth.make_barcoded_read(8, 3, 44, 800 + i, SB, EB, rc=bool(i % 3 == 0), margin=5.0, flank=(5, 14))
             for i in range(6)]
    out = dec.decode_with_barcodes([x["post"] for x in reads], SB, EB)
    n_ok = 0
    for x, (loc, res) in zip(reads, out):
        want = oracle.locate_payload(x["post"], SB, EB, 8 + 44 + 1)
        assert loc == {k: want[k] for k in ("ok", "start_pos", "end_pos", "rc", "dist_start", "dist_end")}
        if not want["ok"]:
            assert res is None
            continue
        n_ok += 1
        window = x["post"][want["start_pos"]:want["end_pos"] + 1]
        wm, ws = oracle.OracleCode(8, 3, 44, rc=want["rc"]).decode(window, 4, 20)
        assert np.array_equal(res[0], wm) and np.array_equal(res[1].view(np.uint32), ws.view(np.uint32))
        assert np.array_equal(res[0][0], x["msg"])          # and the chain recovers the message
    assert n_ok >= 5


def test_argument_errors_and_empty_batches(dec):
    import ctypes
    from nanopore_dna_storage_amd import _lib
    L = _lib.load_library()
    assert dec.basecall([]) == []
    assert dec.locate_payload([], SB, EB) == []
    assert dec.find_barcode([], [], SB, EB) == []
    post = np.zeros((10, 40), np.float32)
    bad_off = np.array([0, 7, 5], np.int64)                          # decreasing offsets
    nb = np.zeros(2, np.int32)
    assert L.lva_basecall_batch(dec._h, post.ctypes.data, bad_off.ctypes.data, 2, None, None, nb.ctypes.data) == -10
    res = (_lib.PayloadPos * 2)()
    assert L.lva_locate_payload_batch(dec._h, post.ctypes.data, bad_off.ctypes.data, 2, SB.encode(), EB.encode(), 10, res) == -10
    assert L.lva_locate_payload_batch(dec._h, post.ctypes.data, None, 1, SB.encode(), EB.encode(), 10, res) == -10
    with pytest.raises(pkg.LvaError):
        dec.locate_payload([post], "ACGU", EB)                       # not a base the reverse complement knows
    # windows: a negative first block is refused, a window shorter than the trellis is reported per read
    dev, off = dec.upload([post])
    try:
        with pytest.raises(pkg.LvaError):
            dec.decode_windows_resident(dev, [-1], [10])
        assert dec.decode_windows_resident(dev, [0], [10]) == [-6]
    finally:
        dec.free(dev)
