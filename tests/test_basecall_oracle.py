"""SURVEY 8(f) row N3 on the CPU: the basecall oracle (PARITY UNPINNED against flappie, see
oracle/basecall_oracle.c) on hand-computable cases and properties, and the host-side barcode search
of the package against the oracle's restatement and brute force."""
import numpy as np
import pytest

from nanopore_dna_storage_amd import helper, synth

SB, EB = "CACCTGTGCTGCGTCAGGCTGTGTC", "GCTGTCCGTTCCGCATTGACACGGC"


def _post_for_path(states, stay_blocks=2, good=0.0, bad=-9.0):
    """log-weights that make `states` (one crf state per base) the unique best path: every block scores `bad`
    everywhere except one transition"""
    rows = []
    cur = None
    for st in states:
        for k in range(1 + stay_blocks):
            row = np.full(40, bad, np.float32)
            frm = st if (cur is None or k > 0) else cur
            row[synth.transition_index(frm, st)] = good
            rows.append(row)
        cur = st
    return np.stack(rows)


def test_hand_made_path(oracle):
    states = [0, 1, 5, 1, 2, 6, 2, 3]          # A C C(flop) C(flip) G G(flop) G T
    post = _post_for_path(states)
    bc, trans, path, score = oracle.basecall(post)
    assert score == 0.0
    # path[0] is already the first base (all-bad rows everywhere else)
    assert bc == "CCCGGGT"                      # change_positions starts at pos 1: the first base is not reported
    assert list(trans) == [4, 7, 10, 13, 16, 19, 22]    # path[k] = state after k blocks: base i enters at block 3i (path index 3i+1)
    assert len(path) == post.shape[0] + 1


def test_all_equal_scores_prefer_first_candidate(oracle):
    """strict '>' everywhere (decode.c:163,176, util.c:25): flop stays, flip takes state 0, argmax takes state 0"""
    bc, trans, path, score = oracle.basecall(np.zeros((7, 40), np.float32))
    assert bc == "" and len(trans) == 0 and set(path.tolist()) == {0} and score == 0.0


def test_last_block_transition_is_not_reported(oracle):
    """change_positions is called with nblock, not nblock+1 (flappie.c:274): a change into the last path
    entry is dropped"""
    states = [0, 1]
    post = _post_for_path(states, stay_blocks=0)          # 2 blocks: path = [0, 0, 1]
    bc, trans, path, _ = oracle.basecall(post)
    assert path.tolist() == [0, 0, 1] and bc == ""


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_clean_synthetic_posteriors_are_called_back(oracle, seed):
    rng = np.random.default_rng(seed)
    bases = rng.integers(0, 4, 300).astype(np.uint8)
    bases[10:16] = 2                                       # a homopolymer run: flip/flop alternation
    post = synth.posteriors_from_bases(bases, rng, margin=9.0)
    bc, trans, path, _ = oracle.basecall(post)
    assert bc == "".join("ACGT"[b] for b in bases)
    assert np.all(np.diff(trans) > 0) and trans[0] >= 1 and trans[-1] < post.shape[0]


def _brute(basecall, trans, sb, eb):
    n = len(basecall)
    if len(sb) + len(eb) > n:
        return (-1, -1, float("inf"), float("inf"))
    s = [helper.levenshtein(sb, basecall[i:i + len(sb)]) for i in range(n // 2 + 1 - len(sb))]
    e = [helper.levenshtein(eb, basecall[i:i + len(eb)]) for i in range(n // 2, n - len(eb))]
    if not s or not e:
        return (-1, -1, float("inf"), float("inf"))
    si, ei = int(np.argmin(s)), n // 2 + int(np.argmin(e))
    a, b = int(trans[si + len(sb)]) - 1, int(trans[ei - 1]) - 1
    return (-1, -1, float("inf"), float("inf")) if b < a else (a, b, min(s), min(e))


@pytest.mark.parametrize("seed", range(6))
def test_barcode_search_host_oracle_brute_force_agree(oracle, seed):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(60, 200))
    s = "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    sb, eb = SB[:int(rng.integers(5, 26))], EB[:int(rng.integers(5, 26))]
    if seed % 2 == 0:                                        # plant noisy copies
        p = int(rng.integers(0, 10))
        s = s[:p] + sb[:-2] + "A" + s[p + len(sb):]
        q = n - len(eb) - int(rng.integers(2, 10))
        s = s[:q] + eb + s[q + len(eb):]
    trans = np.cumsum(rng.integers(1, 9, len(s))) + 1
    want = _brute(s, trans, sb, eb)
    assert oracle.find_barcode_pos(s, trans, sb, eb) == want
    assert helper.find_barcode_pos(s, trans, sb, eb) == want


def test_barcode_search_failures(oracle):
    t = list(range(1, 100))
    assert helper.find_barcode_pos("ACGT", t, "ACG", "CGT")[0] == -1                    # too short (helper.py:177-179)
    assert oracle.find_barcode_pos("ACGT", t, "ACG", "CGT")[0] == -1
    s = "ACGTACGTAC"
    assert helper.find_barcode_pos(s, t, "ACGTACGT", "AC") == oracle.find_barcode_pos(s, t, "ACGTACGT", "AC")


def test_find_barcode_pos_in_post_reads_flappie_files(tmp_path, oracle):
    x = synth.make_barcoded_read(8, 3, 44, 5, SB, EB, margin=7.0, flank=(5, 12))
    bc, trans, _, _ = oracle.basecall(x["post"])
    (tmp_path / "r.fastq").write_text("@read\n%s\n+\n%s\n" % (bc, "I" * len(bc)))
    (tmp_path / "r.trans").write_text("".join("%d\n" % t for t in trans))
    got = helper.find_barcode_pos_in_post(str(tmp_path / "r.trans"), str(tmp_path / "r.fastq"), SB, EB)
    assert got == oracle.find_barcode_pos(bc, trans, SB, EB)
    assert got[2] == 0 and got[3] == 0
    # the window holds the oligo: decode-ready length
    assert got[1] - got[0] + 1 >= 8 + 44 + 1


def test_locate_payload_picks_the_orientation(oracle):
    for rc in (False, True):
        x = synth.make_barcoded_read(8, 3, 44, 9, SB, EB, rc=rc, margin=7.0, flank=(5, 12))
        r = oracle.locate_payload(x["post"], SB, EB, 8 + 44 + 1)
        assert r["ok"] and r["rc"] == rc and r["dist_start"] + r["dist_end"] <= 2
