"""The benchmark's operating point inside the -m gpu suite (BASELINE.json configs[1] shape): m=11 r=5/6 L=8 msg_len=180
reads through the DEFAULT 128 read slots, more reads than slots so that slots are refilled while others are mid-read --
every list and score against kernel mode 1 (lva_step_exact: one thread per target, the reference merge verbatim, no
fingerprints, no lazy messages), eight of them against the CPU oracle.  Plus the default path for list sizes above 64
(lva_step_exact) against the oracle.  Reference: viterbi/viterbi_convolutional_code.cpp:589-858."""
import numpy as np
import pytest

import nanopore_dna_storage_amd as pkg
from nanopore_dna_storage_amd import synth

pytestmark = pytest.mark.gpu


def test_benchmark_shape_through_default_slots(oracle):
    m, r, msg_len, L, md, n = 11, 5, 180, 8, 20, 134
    reads = [synth.make_read(m, r, msg_len, 81000 + i, rc=bool(i & 1), margin=3.0 if i % 5 == 0 else 6.0) for i in range(n)]
    posts, rcs = [x["post"] for x in reads], [x["rc"] for x in reads]
    with pkg.Decoder(m, r, msg_len, list_size=L, max_deviation=md) as dec:
        prof = dec.profile()
        assert prof["kernel"] == 4 and prof["slots"] == 128
        got = dec.decode(posts, rc=rcs)
    with pkg.Decoder(m, r, msg_len, list_size=L, max_deviation=md, kernel=1, max_slots=24) as dec:
        assert dec.profile()["kernel"] == 1
        want = dec.decode(posts, rc=rcs)
    for i, (g, w) in enumerate(zip(got, want)):
        assert not isinstance(g, (int, np.integer)) and not isinstance(w, (int, np.integer)), (i, g, w)
        assert np.array_equal(g[0], w[0]), "read %d: list differs from kernel mode 1" % i
        assert np.array_equal(g[1].view(np.uint32), w[1].view(np.uint32)), "read %d: scores differ from kernel mode 1" % i
    for i in (0, 5, 10, 41, 64, 77, 100, 133):              # noisy / clean, forward / rc, first and second wave of reads through the slots
        wm, ws = oracle.OracleCode(m, r, msg_len, rc=rcs[i]).decode(posts[i], L, md, num_threads=32)
        assert np.array_equal(got[i][0], wm), "read %d: list differs from the oracle" % i
        assert np.array_equal(got[i][1].view(np.uint32), ws.view(np.uint32)), "read %d: scores differ from the oracle" % i


@pytest.mark.parametrize("L,md", [(100, 20), (65, 6)])
def test_list_sizes_above_64_take_the_exact_kernel(oracle, L, md):
    reads = synth.make_reads(6, 1, 60, 3, seed0=3300 + L, rc_mode="odd", margin=3.0)
    with pkg.Decoder(6, 1, 60, list_size=L, max_deviation=md, max_slots=2) as dec:
        assert dec.profile()["kernel"] == 1
        got = dec.decode([x["post"] for x in reads], rc=[x["rc"] for x in reads])
    for x, g in zip(reads, got):
        wm, ws = oracle.OracleCode(6, 1, 60, rc=x["rc"]).decode(x["post"], L, md, num_threads=8)
        assert np.array_equal(g[0], wm)
        assert np.array_equal(g[1].view(np.uint32), ws.view(np.uint32))


# (one list size per kernel mode beyond the default: the non-finite paths of a mode do not depend on L -- suite time budget)
@pytest.mark.parametrize("kernel,L", [(0, 1), (0, 4), (0, 16), (1, 4), (2, 1), (2, 16), (3, 4)])
def test_nan_and_plus_inf_posteriors(oracle, kernel, L):
    """NaN and +inf log-posteriors: the reference decodes them (golden m6_r1_*_nan*, *_posinf*: its own lists); every kernel
    must do what it does -- non-finite sums leave the fast paths for the exact one (:725-727, :756-758, :790-796)"""
    if kernel == 3 and L == 1:
        pytest.skip("the wavefront-per-target kernel has no L = 1")
    reads = synth.make_reads(6, 1, 60, 4, seed0=660 + L, rc_mode="odd", margin=4.0)
    rng = np.random.default_rng(77 + L)
    for i, x in enumerate(reads):
        p = x["post"].copy()
        u = rng.random(p.shape)
        if i != 1:
            p[u < 0.006] = np.nan
        if i != 0:
            p[(u >= 0.006) & (u < 0.012)] = np.inf
        x["post"] = p
    with pkg.Decoder(6, 1, 60, list_size=L, max_deviation=20, kernel=kernel, max_slots=3) as dec:
        got = dec.decode([x["post"] for x in reads], rc=[x["rc"] for x in reads])
    for x, g in zip(reads, got):
        wm, ws = oracle.OracleCode(6, 1, 60, rc=x["rc"]).decode(x["post"], L, 20, num_threads=4)
        assert not isinstance(g, (int, np.integer)), g
        assert np.array_equal(g[0], wm)
        assert np.array_equal(g[1].view(np.uint32), ws.view(np.uint32))
