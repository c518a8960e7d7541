"""The product's host-side code tables (C ABI, no GPU) against the oracle's restatement of
set_conv_params / is_valid_state / find_prev_states."""
import numpy as np
import pytest

import nanopore_dna_storage_amd as pkg
from golden_util import encode_cases

CODES = [(6, 1, 180, ""), (6, 3, 180, ""), (6, 5, 60, ""), (8, 1, 100, ""), (8, 2, 100, ""), (8, 3, 164, ""),
         (8, 4, 100, ""), (8, 5, 180, ""), (11, 1, 40, ""), (11, 2, 180, ""), (11, 5, 180, ""), (14, 7, 180, ""),
         (6, 1, 60, "110"), (8, 5, 180, "1011")]


@pytest.mark.parametrize("m,r,msg_len,sync", CODES)
@pytest.mark.parametrize("rc", [False, True])
def test_tables_match_oracle(oracle, m, r, msg_len, sync, rc):
    period = 9 if sync else 0
    info = pkg.code_info(m, r, msg_len, rc, sync, period)
    tab = pkg.code_tables(m, r, msg_len, rc, sync, period)
    oc = oracle.OracleCode(m, r, msg_len, rc=rc, sync_marker=sync, sync_period=period)
    assert info.nstate_pos == oc.nstate_pos and info.nstate_conv == oc.nstate_conv
    assert info.initial_state == oc.initial_state and info.final_state == oc.final_state
    assert np.array_equal(tab["pos2msg"], oc.pos2msg())
    assert [int(x) for x in tab["ptype"]] == [oc.pattern_at(p) for p in range(oc.nstate_pos)]
    N = info.nstate_conv
    rng = np.random.default_rng(m * 100 + r)
    convs = np.arange(N) if N <= 2048 else np.unique(np.concatenate([rng.integers(0, N, 1500), [0, N - 1, oc.initial_state, oc.final_state]]))
    # is_valid_state == (c & vmask) == vval
    for pos in range(oc.nstate_pos):
        want = np.array([oc.is_valid_state(pos, int(c)) for c in (convs if N <= 256 else convs[::7])])
        cc = convs if N <= 256 else convs[::7]
        got = (cc & tab["vmask"][pos]) == tab["vval"][pos]
        assert np.array_equal(got, want), pos
    # find_prev_states: stay first, then for every source crf state exactly the one conv source the table names
    used = sorted(set(int(x) for x in tab["ptype"][1:]))
    for T in used:
        sh = 1 if T == 0 else 2
        for c in (convs if N <= 256 else convs[::5]):
            c = int(c)
            pk = int(tab["predtab"][T][c])
            for k in range(8):
                ps = oc.prev_states(c, k, T)
                assert tuple(ps[0][:5]) == (c, k, 4 if k >= 4 else k, k, 0)
                nib = (pk >> (4 * (k & 3))) & 0xF
                srcs = [s for s in range(8) if s != k and (k < 4 or s == k - 4)]
                if not (nib & 8):
                    assert len(ps) == 1
                    continue
                cp = ((c << sh) | (nib & 7)) & (N - 1)
                newest, second = c >> (m - 1), (c >> (m - 2)) & 1
                nb = newest if sh == 1 else 2 * second + newest
                assert len(ps) == 1 + len(srcs)
                for row, s in zip(ps[1:], srcs):
                    assert tuple(row) == (cp, s, 4 if k >= 4 else k, s, sh, nb)


def test_encoder_matches_reference_outputs():
    for c in encode_cases()["cases"]:
        msgs = np.array([[int(b) for b in s] for s in c["msgs"]], np.uint8)
        got = pkg.encode(c["mem_conv"], c["rate"], c["msg_len"], msgs)
        assert [pkg.bases_to_str(o) for o in got] == c["oligos"]


def test_bad_parameters():
    for c in encode_cases()["bad_params"]:
        if c["accepted"]:
            pkg.code_info(c["mem_conv"], c["rate"], c["msg_len"])
        else:
            with pytest.raises(pkg.LvaError):
                pkg.code_info(c["mem_conv"], c["rate"], c["msg_len"])
    with pytest.raises(pkg.LvaError):
        pkg.code_info(6, 1, 60, False, "12", 5)        # bad character
    with pytest.raises(pkg.LvaError):
        pkg.code_info(6, 1, 60, False, "110", 2)       # period shorter than marker


def test_algorithmic_bytes_match_survey():
    """111.9 GB/read at m=11 r=5 L=8 nblk=510; 0.446 GB at m=6 r=1 L=1 nblk=823"""
    b = pkg.algorithmic_bytes(11, 5, 180, 510, 8, 20)
    assert abs(b / 1e9 - 111.9) < 0.6
    b = pkg.algorithmic_bytes(6, 1, 180, 823, 1, 20)
    assert abs(b / 1e9 - 0.446) < 0.01


def test_one_bit_positions_have_complementary_base_pairs():
    """What the compact lists of the kernels rest on (lva_device.h Geometry::cmp): at a one-bit position (block type 0) the bases a
    valid conv state can end in are a complementary pair, {A,T} or {C,G} -- the two predecessors of a target differ in the register
    bit the step shifts out, which both generators tap (viterbi_convolutional_code.cpp:269-289, :890-904) -- so crf state k's list
    can be stored as list k >> 1.  Every supported memory, both orientations, a sync marker included."""
    import nanopore_dna_storage_amd as pkg
    seen = 0
    for m, r, ml, kw in [(6, 1, 60, {}), (6, 3, 60, {}), (6, 5, 180, {}), (8, 1, 100, {}), (8, 2, 100, {}), (8, 3, 164, {}), (8, 4, 100, {}),
                         (8, 5, 180, {}), (11, 1, 40, {}), (11, 2, 61, {}), (11, 5, 180, {}), (14, 1, 20, {}), (14, 7, 180, {}),
                         (6, 1, 60, dict(sync_marker="110", sync_period=9))]:
        for rc in (False, True):
            t = pkg.code_tables(m, r, ml, rc=rc, **kw)
            c = np.arange(1 << m)
            pk = t["predtab"][0][:1 << m].astype(np.uint32)
            has = ((pk >> 3) & 1) | (((pk >> 7) & 1) << 1) | (((pk >> 11) & 1) << 2) | (((pk >> 15) & 1) << 3)
            for pos in range(1, len(t["ptype"])):
                if t["ptype"][pos] != 0:
                    continue
                h = has[(c & t["vmask"][pos]) == t["vval"][pos]]
                assert np.all((h == 0b1001) | (h == 0b0110)), (m, r, ml, rc, pos)
                seen += len(h)
    assert seen > 100000
