"""Lazy messages (kernel mode 4, the default for L = 2/4/8) with FOUR message planes -- messages wider than 192 bits,
the m = 14 production path (lva_step_lazy<.,4,.>, lva_step_fixup_lazy<4>, one entry in flight, load_msg_np<4,4>) --
against the CPU oracle, scores bit for bit: tie stress, tiny bands (the stale row and the two-hop bookkeeping),
slot turnover with odd and even block counts, a 4-entry work list (whole-step redo on the exact path).
Reference: viterbi/viterbi_convolutional_code.cpp:667-687 (band), :762-796 (merge).  m = 6 trellises keep the
oracle cheap; the kernels are the same instances m = 14 runs."""
import numpy as np
import pytest

import nanopore_dna_storage_amd as pkg
from nanopore_dna_storage_amd import synth

pytestmark = pytest.mark.gpu


def _compare(oracle, m, r, msg_len, L, md, reads, kernel=4, max_slots=0, threads=8):
    assert msg_len + m > 192, "four message planes"
    with pkg.Decoder(m, r, msg_len, list_size=L, max_deviation=md, max_slots=max_slots, kernel=kernel) as dec:
        assert dec.profile()["kernel"] == 4
        got = dec.decode([x["post"] for x in reads], rc=[x["rc"] for x in reads])
    for i, (x, g) in enumerate(zip(reads, got)):
        want_msgs, want_scores = oracle.OracleCode(m, r, msg_len, rc=x["rc"]).decode(x["post"], L, md, num_threads=threads)
        assert not isinstance(g, (int, np.integer)), "read %d: error %r" % (i, g)
        assert np.array_equal(g[1].view(np.uint32), want_scores.view(np.uint32)), "read %d: scores differ" % i
        assert np.array_equal(g[0], want_msgs), "read %d (nblk %d): list differs" % (i, x["post"].shape[0])


@pytest.mark.parametrize("m,r,msg_len,L,md,n,margin", [
    (6, 1, 200, 8, 20, 3, 3.0), (6, 5, 190, 4, 20, 3, 3.0), (6, 3, 247, 2, 20, 2, 2.5), (8, 1, 240, 8, 20, 2, 3.0),
    (8, 3, 188, 8, 10, 2, 2.5), (6, 1, 249, 8, None, 1, 3.0)])
def test_wide_lazy_matches_oracle(oracle, m, r, msg_len, L, md, n, margin):
    reads = synth.make_reads(m, r, msg_len, n, seed0=900 * m + r, rc_mode="odd", margin=margin)
    _compare(oracle, m, r, msg_len, L, md, reads)


@pytest.mark.parametrize("L,quantum", [(8, 0.25), (4, 0.5), (2, 0.25)])
def test_wide_lazy_tie_stress(oracle, L, quantum):
    reads = synth.make_reads(6, 1, 200, 3, seed0=17 + L, rc_mode="odd", margin=3.0, quantum=quantum)
    _compare(oracle, 6, 1, 200, L, 20, reads)
    reads = synth.make_reads(6, 5, 190, 2, seed0=27 + L, rc_mode="odd", margin=3.0, quantum=quantum)
    _compare(oracle, 6, 5, 190, L, 20, reads)


@pytest.mark.parametrize("md", [1, 2, 3, 5])
def test_wide_lazy_tiny_bands(oracle, md):
    reads = synth.make_reads(6, 1, 190, 4, seed0=140 + md, rc_mode="odd", margin=3.0)
    _compare(oracle, 6, 1, 190, 4, md, reads)
    reads = synth.make_reads(8, 3, 188, 2, seed0=190 + md, rc_mode="odd", margin=2.5)
    _compare(oracle, 8, 3, 188, 8, md, reads)


def test_wide_lazy_slot_turnover(oracle):
    """64 reads of different lengths (odd and even block counts) through 3 slots"""
    reads = [synth.make_read(6, 1, 190, 7000 + i, rc=bool(i % 3 == 0), margin=3.0 + (i % 4)) for i in range(64)]
    assert len({x["post"].shape[0] & 1 for x in reads}) == 2
    _compare(oracle, 6, 1, 190, 2, 6, reads, max_slots=3)
    _compare(oracle, 6, 1, 190, 8, 4, reads[:20], max_slots=3)


@pytest.mark.parametrize("kernel", [0, 4])
@pytest.mark.parametrize("m,r,msg_len,L,md", [(6, 1, 200, 8, 20), (6, 1, 190, 2, 3), (6, 5, 190, 4, 20)])
def test_wide_lazy_work_list_overflow(oracle, monkeypatch, kernel, m, r, msg_len, L, md):
    monkeypatch.setenv("LVA_WORK_CAP", "4")
    reads = synth.make_reads(m, r, msg_len, 5, seed0=41, rc_mode="odd", margin=3.0, quantum=0.25)
    reads[1]["post"] = reads[1]["post"][:-1].copy()            # the other parity of the last step
    _compare(oracle, m, r, msg_len, L, md, reads, kernel=kernel, max_slots=2)


@pytest.mark.parametrize("m,r,msg_len,L,md", [(11, 5, 187, 8, 20), (14, 7, 180, 4, 5)])
def test_wide_lazy_big_trellises(oracle, m, r, msg_len, L, md):
    """four planes on the production trellises: m=11 at msg_len 187 (198 bits), m=14 at the paper's 180 with a short band"""
    reads = [synth.make_read(m, r, msg_len, 4300 + m + i, rc=bool(i & 1), margin=3.0) for i in range(2)]
    _compare(oracle, m, r, msg_len, L, md, reads, kernel=0, max_slots=2, threads=32)
