"""Kernel instances and slot configurations no other test launches: every instance of the
dispatcher in launch_step_fast (lva_kernels.hip) runs at least once against the CPU oracle or the reference's own lists."""
import numpy as np
import pytest

import nanopore_dna_storage_amd as pkg
from nanopore_dna_storage_amd import synth
from golden_util import as_strings, load_case

pytestmark = pytest.mark.gpu


def _same(oracle, m, r, msg_len, L, md, reads, **kw):
    with pkg.Decoder(m, r, msg_len, list_size=L, max_deviation=md, **kw) as dec:
        assert dec.profile()["kernel"] == 2
        got = dec.decode([x["post"] for x in reads], rc=[x["rc"] for x in reads])
        prof = dec.profile()
    for i, (x, g) in enumerate(zip(reads, got)):
        want_msgs, want_scores = oracle.OracleCode(m, r, msg_len, rc=x["rc"]).decode(x["post"], L, md, num_threads=16)
        assert not isinstance(g, int), "read %d: error %r" % (i, g)
        assert np.array_equal(g[0], want_msgs), "read %d: list differs" % i
        assert np.array_equal(g[1].view(np.uint32), want_scores.view(np.uint32)), "read %d: scores differ" % i
    return prof


# (msg_len, L) -> instance: 156 message bits = three planes, L not a multiple of 4 -> lva_step_big<64,3> on the plane layout;
# 206 bits = four planes -> lva_step_big<32,4> (L = 24) and lva_step_big<64,4> (L = 40, 64)
@pytest.mark.parametrize("msg_len,L", [(150, 37), (150, 63), (200, 24), (200, 40), (200, 64)])
def test_plane_layout_big_list_instances(oracle, monkeypatch, msg_len, L):
    reads = [synth.make_read(6, 1, msg_len, 3300 + i, rc=bool(i & 1), margin=2.5) for i in range(2)]
    reads.append(synth.make_read(6, 1, msg_len, 3400, margin=2.0, quantum=0.5))          # ties: the wavefront fix-up on this layout
    assert len({x["post"].shape[0] & 1 for x in reads}) == 2
    prof = _same(oracle, 6, 1, msg_len, L, 20, reads, kernel=2, max_slots=2)
    assert prof["fixup_states"] > 0 and prof["overflow_steps"] == 0
    monkeypatch.setenv("LVA_WORK_CAP", "4")                                              # the whole-step redo behind the same instance
    prof = _same(oracle, 6, 1, msg_len, L, 20, reads[2:], kernel=2, max_slots=1)
    assert prof["overflow_steps"] > 0


def test_m14_through_default_slots():
    """configs[3] (16384 conv states, four message planes): ten reads of odd and even block counts and both orientations in ONE
    call through the default slot count, then through three slots (turnover); the reference's own lists for the three golden
    reads among them, kernel mode 1 (one thread per target, the reference merge verbatim) for all ten."""
    names = ["m14_r7_L8", "m14_r7_L8_noisy", "m14_r7_L8_rc"]
    gold = [load_case(n) for n in names]
    m0 = gold[0][0]
    m, r, msg_len, L, md = m0["mem_conv"], m0["rate"], m0["msg_len"], m0["list_size"], m0["max_deviation"]
    assert (m, r, L) == (14, 7, 8) and all(g[0]["msg_len"] == msg_len and g[0]["max_deviation"] == md for g in gold)
    reads = [dict(post=p, rc=bool(mm["rc"])) for mm, p, _ in gold]
    for i in range(7):
        x = synth.make_read(m, r, msg_len, 1400 + i, rc=bool(i % 3 == 1), margin=3.0 + (i % 2))
        if i == 4:
            x["post"] = x["post"][:-1].copy()
        reads.append(dict(post=x["post"], rc=x["rc"]))
    assert len({x["post"].shape[0] & 1 for x in reads}) == 2
    posts, rcs = [x["post"] for x in reads], [x["rc"] for x in reads]
    with pkg.Decoder(m, r, msg_len, list_size=L, max_deviation=md) as dec:
        assert dec.profile()["kernel"] == 4 and dec.profile()["slots"] >= 8
        got = dec.decode(posts, rc=rcs)
    for (mm, _, lines), g in zip(gold, got):
        assert as_strings(g[0]) == lines
    with pkg.Decoder(m, r, msg_len, list_size=L, max_deviation=md, max_slots=3) as dec:
        again = dec.decode(posts, rc=rcs)
    with pkg.Decoder(m, r, msg_len, list_size=L, max_deviation=md, kernel=1, max_slots=4) as dec:
        want = dec.decode(posts, rc=rcs)
    for i, (g, a, w) in enumerate(zip(got, again, want)):
        for h in (g, a):
            assert np.array_equal(h[0], w[0]) and np.array_equal(h[1].view(np.uint32), w[1].view(np.uint32)), "read %d" % i


@pytest.mark.parametrize("md,L", [(1, 4), (2, 12), (3, 8)])
def test_tiny_bands_and_barely_long_enough_reads_through_reused_slots(oracle, md, L):
    """max_deviation 1..3 with nblk = nstate_pos + 1 .. + 6 (the shortest reads the reference accepts, :600-601): the working band's
    lower clip (npos - nblk + t) is active at every step, the row below the band is stale from the first steps on, and 36 reads
    reuse two slots, so that anything the clipped band wrongly left unwritten would be a previous read's ring contents."""
    m, r, msg_len = 6, 1, 24
    npos = pkg.code_info(m, r, msg_len).nstate_pos
    reads = []
    for i in range(36):
        x = synth.make_read(m, r, msg_len, 8800 + i, rc=bool(i % 3 == 0), margin=2.5 + (i % 3))
        assert x["post"].shape[0] > npos + 6
        reads.append(dict(post=x["post"][:npos + 1 + (i % 6)].copy(), rc=x["rc"]))
    with pkg.Decoder(m, r, msg_len, list_size=L, max_deviation=md, max_slots=2) as dec:
        got = dec.decode([x["post"] for x in reads], rc=[x["rc"] for x in reads])
    codes = {rc: oracle.OracleCode(m, r, msg_len, rc=rc) for rc in (False, True)}
    for i, (x, g) in enumerate(zip(reads, got)):
        wm, ws = codes[x["rc"]].decode(x["post"], L, md)
        assert not isinstance(g, int), "read %d: error %r" % (i, g)
        assert np.array_equal(g[0], wm) and np.array_equal(g[1].view(np.uint32), ws.view(np.uint32)), "read %d" % i
