"""GPU parity: the HIP decoder (through the C ABI) against the CPU oracle, bit-exact."""
import numpy as np
import pytest

import nanopore_dna_storage_amd as pkg
from nanopore_dna_storage_amd import synth

pytestmark = pytest.mark.gpu


def _compare(oracle, m, r, msg_len, L, md, reads, kernel=0, max_slots=0, sync_marker="", sync_period=0):
    with pkg.Decoder(m, r, msg_len, list_size=L, max_deviation=md, max_slots=max_slots, kernel=kernel,
                     sync_marker=sync_marker, sync_period=sync_period) as dec:
        got = dec.decode([x["post"] for x in reads], rc=[x["rc"] for x in reads])
    for i, (x, g) in enumerate(zip(reads, got)):
        code = oracle.OracleCode(m, r, msg_len, rc=x["rc"], sync_marker=sync_marker, sync_period=sync_period)
        want_msgs, want_scores = code.decode(x["post"], L, md, num_threads=16)
        assert not isinstance(g, int), "read %d: error %r" % (i, g)
        assert g[0].shape == want_msgs.shape, "read %d: %d entries, oracle %d" % (i, len(g[0]), len(want_msgs))
        assert np.array_equal(g[0], want_msgs), "read %d: list differs" % i
        assert np.array_equal(g[1].view(np.uint32), want_scores.view(np.uint32)), "read %d: scores differ" % i


CASES = [
    # m, rate, msg_len, L, max_dev, n_reads, margin, rc_mode
    (6, 1, 60, 1, 20, 4, 6.0, "odd"),
    (6, 1, 60, 4, 20, 4, 3.0, "odd"),
    (6, 3, 60, 8, 10, 4, 3.0, "odd"),
    (6, 5, 180, 8, 20, 3, 3.0, "odd"),
    (8, 1, 100, 8, 20, 2, 3.0, "odd"),
    (8, 2, 100, 2, 20, 2, 4.0, "odd"),
    (8, 4, 100, 4, 20, 2, 3.0, "odd"),
    (8, 5, 100, 8, 20, 2, 3.0, "odd"),
    (6, 1, 60, 8, None, 2, 3.0, "odd"),      # unbanded (reference default max_deviation)
    (6, 1, 60, 16, 20, 2, 3.0, "odd"),       # list longer than 8
]


# (the exact and the wavefront kernel run ONE code path each whatever the code: every other case is enough for them -- the goldens,
#  the fuzz files and bench.py's cross-check exercise mode 1 on every other shape; the suite has a time budget)
@pytest.mark.parametrize("m,r,msg_len,L,md,n,margin,rc_mode", CASES[::2] + CASES[-1:])
def test_exact_kernel_matches_oracle(oracle, m, r, msg_len, L, md, n, margin, rc_mode):
    reads = synth.make_reads(m, r, msg_len, n, seed0=100 * m + r, rc_mode=rc_mode, margin=margin)
    _compare(oracle, m, r, msg_len, L, md, reads, kernel=1)


@pytest.mark.parametrize("m,r,msg_len,L,md,n,margin,rc_mode", CASES)
def test_fast_kernel_matches_oracle(oracle, m, r, msg_len, L, md, n, margin, rc_mode):
    reads = synth.make_reads(m, r, msg_len, n, seed0=100 * m + r, rc_mode=rc_mode, margin=margin)
    _compare(oracle, m, r, msg_len, L, md, reads, kernel=2)


@pytest.mark.parametrize("m,r,msg_len,L,md,n,margin,rc_mode", [c for c in CASES if c[3] >= 2][1::2])
def test_wave_kernel_matches_oracle(oracle, m, r, msg_len, L, md, n, margin, rc_mode):
    reads = synth.make_reads(m, r, msg_len, n, seed0=100 * m + r, rc_mode=rc_mode, margin=margin)
    _compare(oracle, m, r, msg_len, L, md, reads, kernel=3)


@pytest.mark.parametrize("kernel", [1, 2, 3])
def test_list_of_twelve_with_sync_marker_on_every_kernel(oracle, kernel):
    """a list size between 9 and 64 that is NOT on the record layout (12: plane layout, compact lists at one-bit positions --
    Geometry::cmp, which resolve_target / wave_target / exact_state must honour), forward and reverse complement, with a sync
    marker: the exact kernel (1), the big-list kernel + wavefront fix-up (2) and the wavefront kernel (3) against the oracle"""
    reads = synth.make_reads(6, 2, 62, 4, seed0=77, rc_mode="odd", margin=3.0)
    _compare(oracle, 6, 2, 62, 12, 20, reads, kernel=kernel, sync_marker="10", sync_period=7)


@pytest.mark.parametrize("kernel", [1, 2, 3])
def test_tie_stress(oracle, kernel):
    """posteriors on a 0.25 grid: exact fp32 score ties everywhere, libstdc++ heap order decides"""
    reads = synth.make_reads(6, 1, 60, 4, seed0=7, rc_mode="odd", margin=3.0, quantum=0.25)
    _compare(oracle, 6, 1, 60, 8, 20, reads, kernel=kernel)


def test_fast_kernel_queues_ties_for_the_exact_kernel():
    reads = synth.make_reads(6, 1, 60, 2, seed0=7, margin=3.0, quantum=0.25)
    with pkg.Decoder(6, 1, 60, list_size=8, max_deviation=20, kernel=2) as dec:
        dec.decode([x["post"] for x in reads])
        assert dec.profile()["fixup_states"] > 0
        assert dec.profile()["kernel"] == 2


def test_minus_inf_posteriors(oracle):
    """-inf log-posteriors (zero probability transitions) go through the exact path"""
    reads = synth.make_reads(6, 1, 60, 2, seed0=21, margin=4.0)
    rng = np.random.default_rng(5)
    for x in reads:
        p = x["post"].copy()
        p[rng.random(p.shape) < 0.02] = -np.inf
        x["post"] = p
    for kernel in (1, 2, 3):
        _compare(oracle, 6, 1, 60, 4, 20, reads, kernel=kernel)
    _compare(oracle, 6, 1, 60, 16, 20, reads, kernel=2)


@pytest.mark.parametrize("kernel", [1, 2])
def test_more_reads_than_slots(oracle, kernel):
    reads = synth.make_reads(6, 1, 60, 7, seed0=50, rc_mode="odd", margin=4.0)
    _compare(oracle, 6, 1, 60, 4, 20, reads, kernel=kernel, max_slots=2)


@pytest.mark.parametrize("kernel", [1, 2])
def test_sync_marker(oracle, kernel):
    reads = synth.make_reads(6, 1, 60, 2, seed0=9, margin=4.0)
    _compare(oracle, 6, 1, 60, 4, 20, reads, kernel=kernel, sync_marker="110", sync_period=9)


def test_short_post_is_reported_per_read(oracle):
    reads = synth.make_reads(6, 1, 60, 2, seed0=3, margin=6.0)
    posts = [reads[0]["post"], reads[1]["post"][:30]]
    with pkg.Decoder(6, 1, 60, list_size=2, max_deviation=20) as dec:
        got = dec.decode(posts)
    assert not isinstance(got[0], int)
    assert got[1] == -6


def test_work_list_overflow_falls_back_to_the_exact_step(oracle, monkeypatch):
    """a 4-entry work list overflows at once on tie-heavy input: the fix-up kernel then redoes whole steps"""
    monkeypatch.setenv("LVA_WORK_CAP", "4")
    reads = synth.make_reads(6, 1, 60, 3, seed0=31, rc_mode="odd", margin=3.0, quantum=0.25)
    _compare(oracle, 6, 1, 60, 8, 20, reads, kernel=2)


def test_work_cap_is_ignored_outside_tests(monkeypatch):
    """LVA_WORK_CAP without LVA_TESTING=1 changes nothing: a stray variable in a user's environment must not cost an order of
    magnitude silently"""
    monkeypatch.setenv("LVA_WORK_CAP", "4")
    reads = synth.make_reads(6, 1, 60, 2, seed0=31, margin=3.0, quantum=0.25)
    seen = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("LVA_TESTING", flag)
        with pkg.Decoder(6, 1, 60, list_size=8, max_deviation=20) as dec:
            dec.decode([x["post"] for x in reads])
            seen[flag] = dec.profile()["overflow_steps"]
    assert seen["0"] == 0 and seen["1"] > 0, seen


@pytest.mark.parametrize("kernel", [0, 4])
@pytest.mark.parametrize("m,r,msg_len,L,md", [(6, 1, 60, 8, 20), (6, 1, 60, 2, 3), (8, 3, 44, 4, 20)])
def test_work_list_overflow_in_lazy_mode_redoes_the_step_exactly(oracle, monkeypatch, kernel, m, r, msg_len, L, md):
    """kernel mode 4 (the default for L = 2/4/8): a 4-entry work list overflows on every step of tie-heavy input; the
    lazy fix-up then redoes the whole step on its exact path instead of refusing the batch (reference :762-796 decodes
    any finite matrix).  Odd and even block counts, both orientations, slot turnover."""
    monkeypatch.setenv("LVA_WORK_CAP", "4")
    reads = synth.make_reads(m, r, msg_len, 5, seed0=31, rc_mode="odd", margin=3.0, quantum=0.25)
    reads[1]["post"] = reads[1]["post"][:-1].copy()            # the other parity of the last step
    with pkg.Decoder(m, r, msg_len, list_size=L, max_deviation=md, kernel=kernel, max_slots=2) as dec:
        assert dec.profile()["kernel"] == 4
    _compare(oracle, m, r, msg_len, L, md, reads, kernel=kernel, max_slots=2)


def test_tie_dense_batch_through_default_slots(oracle):
    """300 copies of the two tie-stress goldens through the DEFAULT slot count (m=6 L=8: 1024 slots, kernel mode 4):
    tens of millions of fix-up targets per launch against a work list of 2^20 -- the overflow path at production
    sizes; every list must equal the reference's (.list files made by the reference binary)."""
    from golden_util import as_strings, load_case
    for name in ("m6_r1_L8_ties", "m6_r5_L8_ties_rc"):
        m, post, lines = load_case(name)
        with pkg.Decoder(m["mem_conv"], m["rate"], m["msg_len"], list_size=m["list_size"], max_deviation=m["max_deviation"]) as dec:
            assert dec.profile()["kernel"] == 4 and dec.profile()["slots"] == 1024
            got = dec.decode([post] * 300, rc=[m["rc"]] * 300)
            assert dec.profile()["overflow_steps"] > 0        # whole steps were redone on the exact path, and the caller can see it
        for g in got:
            assert not isinstance(g, int), "decode error %r" % (g,)
            assert as_strings(g[0]) == lines


@pytest.mark.parametrize("md", [0, 1, 2])
@pytest.mark.parametrize("kernel", [1, 2, 3])
def test_tiny_bands(oracle, md, kernel):
    """max_deviation 0 (empty band: the reference writes an empty list), 1 and 2 (ring of 3 / 5 positions)"""
    reads = synth.make_reads(6, 1, 24, 3, seed0=77, rc_mode="odd", margin=5.0)
    _compare(oracle, 6, 1, 24, 4, md, reads, kernel=kernel)


@pytest.mark.parametrize("m,r,msg_len,L,rc,sync", [(11, 1, 40, 8, False, ""), (11, 2, 61, 4, True, ""), (11, 5, 100, 8, True, "1011")])
def test_m11_variants(oracle, m, r, msg_len, L, rc, sync):
    reads = [synth.make_read(m, r, msg_len, 400 + i, rc=rc, margin=3.5) for i in range(2)]
    _compare(oracle, m, r, msg_len, L, 20, reads, kernel=0, sync_marker=sync, sync_period=12 if sync else 0)


@pytest.mark.parametrize("L", [3, 7, 12, 16, 33, 64])
def test_long_and_odd_lists_use_the_big_list_kernel(oracle, L):
    reads = synth.make_reads(8, 3, 44, 3, seed0=600 + L, rc_mode="odd", margin=3.0)
    with pkg.Decoder(8, 3, 44, list_size=L, max_deviation=20) as dec:
        assert dec.profile()["kernel"] == 2
    _compare(oracle, 8, 3, 44, L, 20, reads, kernel=0)
    _compare(oracle, 8, 3, 44, L, 20, reads, kernel=3)


@pytest.mark.parametrize("m,r,msg_len,L", [(8, 5, 180, 24), (8, 1, 240, 16), (6, 1, 150, 9)])
def test_big_list_kernel_wide_messages(oracle, m, r, msg_len, L):
    """3 and 4 message planes (lva_step_big<LL,3>, <LL,4>)"""
    reads = [synth.make_read(m, r, msg_len, 1200 + i, rc=bool(i & 1), margin=3.0) for i in range(2)]
    _compare(oracle, m, r, msg_len, L, 10, reads, kernel=2)


@pytest.mark.parametrize("L", [5, 16, 64])
def test_big_list_kernel_tie_stress(oracle, L):
    """quantised posteriors: equal scores everywhere, most targets go through the wavefront fix-up"""
    reads = [synth.make_read(6, 1, 40, 900 + i, rc=bool(i & 1), margin=2.0, quantum=0.5) for i in range(2)]
    _compare(oracle, 6, 1, 40, L, 20, reads, kernel=2)


def test_big_list_kernel_work_list_overflow(oracle, monkeypatch):
    monkeypatch.setenv("LVA_WORK_CAP", "4")
    reads = [synth.make_read(6, 1, 40, 950 + i, rc=bool(i & 1), margin=2.0, quantum=0.5) for i in range(2)]
    _compare(oracle, 6, 1, 40, 16, 20, reads, kernel=2)


@pytest.mark.parametrize("L", [5, 16, 36, 64])
def test_big_list_kernel_three_planes_tie_stress(oracle, L):
    """three message planes (the benchmark's message width): lva_step_big<LL,3> on the plane layout below 32 entries,
    lva_step_big_rec<LL> on the record layout from 32 on (L a multiple of 4; 36 = a list shorter than the instance's 64) --
    ties send most targets through the wavefront fix-up, which reads and writes the same layout"""
    reads = [synth.make_read(6, 1, 150, 960 + i, rc=bool(i & 1), margin=2.0, quantum=0.5) for i in range(2)]
    _compare(oracle, 6, 1, 150, L, 20, reads, kernel=2)


@pytest.mark.parametrize("L,md", [(32, 20), (12, 3), (40, 2)])
def test_big_list_kernel_three_planes_overflow_and_turnover(oracle, monkeypatch, L, md):
    """three message planes: whole-step redo on the wavefront path (4-entry work list), then 40 reads of odd and even block counts
    through 3 slots with tiny and full bands (stale ring contents, position 0 rewritten by later reads)"""
    reads = [synth.make_read(6, 1, 150, 970 + i, rc=bool(i & 1), margin=2.0, quantum=0.5) for i in range(3)]
    monkeypatch.setenv("LVA_WORK_CAP", "4")
    _compare(oracle, 6, 1, 150, L, md, reads, kernel=2, max_slots=2)
    monkeypatch.delenv("LVA_WORK_CAP")
    reads = [synth.make_read(6, 1, 150, 7700 + i, rc=bool(i % 3 == 0), margin=2.5 + (i % 3)) for i in range(10 if md > 10 else 40)]
    assert len({x["post"].shape[0] & 1 for x in reads}) == 2
    _compare(oracle, 6, 1, 150, L, md, reads, kernel=0, max_slots=3)


def test_many_short_reads_through_few_slots(oracle):
    """200 reads of different lengths through 5 slots: slot turnover, stale ring contents, mixed orientations"""
    reads = [synth.make_read(6, 1, 24, 5000 + i, rc=bool(i % 3 == 0), margin=3.0 + (i % 4)) for i in range(200)]
    with pkg.Decoder(6, 1, 24, list_size=4, max_deviation=6, max_slots=5) as dec:
        got = dec.decode([x["post"] for x in reads], rc=[x["rc"] for x in reads])
    codes = {rc: oracle.OracleCode(6, 1, 24, rc=rc) for rc in (False, True)}
    for x, g in zip(reads, got):
        wm, ws = codes[x["rc"]].decode(x["post"], 4, 6)
        assert np.array_equal(g[0], wm) and np.array_equal(g[1].view(np.uint32), ws.view(np.uint32))


@pytest.mark.parametrize("L", [4, 12])
def test_hundreds_of_slots(oracle, L):
    """small trellises take hundreds of read slots per launch (device slot table, 21 - m bits of slot index in the
    work-list items): 300 reads of different lengths through 200 slots, ties included, both the small- and big-list kernels"""
    reads = [synth.make_read(6, 1, 24, 9000 + i, rc=bool(i % 3 == 0), margin=2.5 + (i % 4), quantum=0.5 if i % 5 == 0 else None)
             for i in range(300)]
    with pkg.Decoder(6, 1, 24, list_size=L, max_deviation=6, max_slots=200) as dec:
        assert dec.profile()["slots"] == 200
        got = dec.decode([x["post"] for x in reads], rc=[x["rc"] for x in reads])
        assert dec.profile()["fixup_states"] > 0              # the exact path was exercised from high slot numbers too
    codes = {rc: oracle.OracleCode(6, 1, 24, rc=rc) for rc in (False, True)}
    for x, g in zip(reads, got):
        wm, ws = codes[x["rc"]].decode(x["post"], L, 6)
        assert np.array_equal(g[0], wm) and np.array_equal(g[1].view(np.uint32), ws.view(np.uint32))


def test_list_size_one_through_its_default_slots(oracle):
    """L = 1 takes four times the slots of the list kernels (4096 at m = 6): 4200 reads of different lengths through them in one
    call -- every slot index in use, the first 104 slots refilled -- against the CPU oracle."""
    reads = [synth.make_read(6, 1, 24, 12000 + i, rc=bool(i % 3 == 0), margin=2.5 + (i % 4), quantum=0.5 if i % 7 == 0 else None)
             for i in range(4200)]
    with pkg.Decoder(6, 1, 24, list_size=1, max_deviation=6) as dec:
        assert dec.profile()["slots"] == 4096
        got = dec.decode([x["post"] for x in reads], rc=[x["rc"] for x in reads])
    codes = {rc: oracle.OracleCode(6, 1, 24, rc=rc) for rc in (False, True)}
    for i, (x, g) in enumerate(zip(reads, got)):
        wm, ws = codes[x["rc"]].decode(x["post"], 1, 6)
        assert not isinstance(g, int), "read %d: error %r" % (i, g)
        assert np.array_equal(g[0], wm) and np.array_equal(g[1].view(np.uint32), ws.view(np.uint32)), "read %d" % i


def test_default_slot_count_grows_for_small_trellises():
    for m, r, ml, L, want in ((6, 1, 60, 2, 1024), (8, 1, 100, 2, 256), (11, 1, 40, 2, 128),
                              (6, 1, 60, 1, 4096), (8, 1, 100, 1, 1024), (11, 1, 40, 1, 128)):     # L = 1: launches a quarter as long
        with pkg.Decoder(m, r, ml, list_size=L, max_deviation=20) as dec:
            assert dec.profile()["slots"] == want


@pytest.mark.parametrize("m,r,msg_len", [(11, 5, 54), (11, 1, 40), (14, 7, 42)])
def test_list_size_one_xcd_aware_tile_order_and_plain_order(oracle, monkeypatch, m, r, msg_len):
    """The L = 1 kernel takes its tiles in the XCD-aware order (PosRec::xs, csrc/lva_kernels.hip xcd_tile) wherever there are at
    least 16 of them; LVA_NO_XCD_ORDER=1 (with LVA_TESTING=1, set by conftest) keeps blockIdx.x order.  Both against the CPU oracle,
    both orientations, reads through reused slots."""
    reads = [synth.make_read(m, r, msg_len, 7100 + i, rc=bool(i & 1), margin=2.0 + (i % 3), sub=0.02 * (i == 3)) for i in range(6)]
    codes = {rc: oracle.OracleCode(m, r, msg_len, rc=rc) for rc in (False, True)}
    want = [codes[x["rc"]].decode(x["post"], 1, 8) for x in reads]
    for plain in (False, True):
        if plain:
            monkeypatch.setenv("LVA_NO_XCD_ORDER", "1")
        with pkg.Decoder(m, r, msg_len, list_size=1, max_deviation=8, max_slots=4) as dec:
            got = dec.decode([x["post"] for x in reads], rc=[x["rc"] for x in reads])
        for i, (g, (wm, ws)) in enumerate(zip(got, want)):
            assert not isinstance(g, int), "read %d: error %r" % (i, g)
            assert np.array_equal(g[0], wm) and np.array_equal(g[1].view(np.uint32), ws.view(np.uint32)), "read %d plain=%s" % (i, plain)
