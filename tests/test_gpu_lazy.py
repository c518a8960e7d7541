"""Kernel mode 4 ("lazy messages": list sizes 2 / 4 / 8; messages materialised every second time step, one-byte
back-pointers in between -- lva_kernels.hip) against the CPU oracle and the reference's golden lists, bit for bit:
clean and noisy reads, both orientations, every band width (the stale row below the band is where the two-hop
bookkeeping is delicate), tie stress (the exact path resolves messages the same way), sync markers, slot turnover,
odd and even numbers of time steps (the final gather follows one hop when the last step stored back-pointers)."""
import numpy as np
import pytest

import nanopore_dna_storage_amd as pkg
from nanopore_dna_storage_amd import synth
from golden_util import as_strings, load_case, manifest, sync_kw

pytestmark = pytest.mark.gpu


def _compare(oracle, m, r, msg_len, L, md, reads, max_slots=0, sync_marker="", sync_period=0):
    with pkg.Decoder(m, r, msg_len, list_size=L, max_deviation=md, max_slots=max_slots, kernel=4,
                     sync_marker=sync_marker, sync_period=sync_period) as dec:
        assert dec.profile()["kernel"] == 4
        got = dec.decode([x["post"] for x in reads], rc=[x["rc"] for x in reads])
    for i, (x, g) in enumerate(zip(reads, got)):
        code = oracle.OracleCode(m, r, msg_len, rc=x["rc"], sync_marker=sync_marker, sync_period=sync_period)
        want_msgs, want_scores = code.decode(x["post"], L, md, num_threads=16)
        assert not isinstance(g, int), "read %d: error %r" % (i, g)
        assert np.array_equal(g[1].view(np.uint32), want_scores.view(np.uint32)), "read %d: scores differ" % i
        assert np.array_equal(g[0], want_msgs), "read %d (nblk %d): list differs" % (i, x["post"].shape[0])


CASES = [
    (6, 1, 60, 4, 20, 6, 3.0), (6, 3, 60, 8, 10, 6, 3.0), (6, 5, 180, 8, 20, 4, 3.0), (8, 1, 100, 8, 20, 3, 3.0),
    (8, 2, 100, 2, 20, 3, 4.0), (8, 4, 100, 4, 20, 3, 3.0), (8, 5, 100, 8, 20, 3, 2.5), (6, 1, 60, 8, None, 3, 3.0),
    (8, 3, 164, 8, 20, 3, 3.0), (6, 1, 24, 2, 20, 6, 2.5),
]


@pytest.mark.parametrize("m,r,msg_len,L,md,n,margin", CASES)
def test_lazy_matches_oracle(oracle, m, r, msg_len, L, md, n, margin):
    reads = synth.make_reads(m, r, msg_len, n, seed0=300 * m + r, rc_mode="odd", margin=margin)
    _compare(oracle, m, r, msg_len, L, md, reads)


@pytest.mark.parametrize("md", [1, 2, 3, 5, 8])
def test_lazy_tiny_bands(oracle, md):
    reads = synth.make_reads(6, 1, 24, 6, seed0=40 + md, rc_mode="odd", margin=3.0)
    _compare(oracle, 6, 1, 24, 4, md, reads)
    reads = synth.make_reads(8, 3, 44, 4, seed0=90 + md, rc_mode="odd", margin=2.5)
    _compare(oracle, 8, 3, 44, 8, md, reads)


def test_lazy_tie_stress(oracle):
    reads = synth.make_reads(6, 1, 60, 4, seed0=7, rc_mode="odd", margin=3.0, quantum=0.25)
    _compare(oracle, 6, 1, 60, 8, 20, reads)
    reads = synth.make_reads(6, 5, 60, 4, seed0=8, rc_mode="odd", margin=3.0, quantum=0.5)
    _compare(oracle, 6, 5, 60, 4, 20, reads)


def test_lazy_sync_marker_and_indels(oracle):
    reads = synth.make_reads(6, 1, 60, 4, seed0=21, rc_mode="odd", margin=4.0)
    _compare(oracle, 6, 1, 60, 4, 20, reads, sync_marker="110", sync_period=9)
    reads = [synth.make_read(6, 1, 60, 500 + i, rc=bool(i & 1), margin=4.0, sub=0.02, dele=0.03, ins=0.01) for i in range(4)]
    _compare(oracle, 6, 1, 60, 8, 20, reads)


def test_lazy_slot_turnover(oracle):
    """150 reads of different lengths (odd and even block counts) through 5 slots"""
    reads = [synth.make_read(6, 1, 24, 6000 + i, rc=bool(i % 3 == 0), margin=3.0 + (i % 4)) for i in range(150)]
    assert len({x["post"].shape[0] & 1 for x in reads}) == 2
    _compare(oracle, 6, 1, 24, 4, 6, reads, max_slots=5)


@pytest.mark.parametrize("m,r,msg_len,L,rc", [(11, 1, 40, 4, False), (11, 5, 100, 8, True), (11, 2, 61, 2, False), (14, 1, 20, 2, True)])
def test_lazy_big_trellises(oracle, m, r, msg_len, L, rc):
    try:
        pkg.code_info(m, r, msg_len)
    except pkg.LvaError:
        pytest.skip("length does not terminate on a base boundary")
    reads = [synth.make_read(m, r, msg_len, 777 + i, rc=rc, margin=3.0) for i in range(2)]
    _compare(oracle, m, r, msg_len, L, 20, reads, max_slots=2)


GOLD = [n for n, v in sorted(manifest().items()) if v["list_size"] in (2, 4, 8) and v["exit_code"] == 0]


@pytest.mark.parametrize("name", GOLD)
def test_lazy_matches_reference_lists(name):
    m, post, lines = load_case(name)
    with pkg.Decoder(m["mem_conv"], m["rate"], m["msg_len"], list_size=m["list_size"], max_deviation=m["max_deviation"],
                     kernel=4, max_slots=2, **sync_kw(m)) as dec:
        res = dec.decode([post], rc=[m["rc"]])[0]
    assert as_strings(res[0]) == lines
