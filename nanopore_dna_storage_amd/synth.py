"""Deterministic synthetic flappie-style posteriors.

The reference's simulator.py produces decoder inputs with scrappy (signal simulation) and
flappie (basecaller network); neither can run here (dependencies and model weights absent),
so reads are synthesised directly in the format flappie writes with --post-output-file
(flappie/src/flappie.c:266-271): float32[nblk][40] log-posteriors over the 40 flip-flop
transitions of a block, normalised so that logsumexp over a block is 0
(flappie/src/decode.c:491).  Index b*8+s = transition into flip base b from state s;
index 32+s = into flop: from flip s (s<4) or staying in flop s (s>=4)
(viterbi_convolutional_code.cpp:553-575, :582-587).

Recipe per read i (RNG = numpy default_rng(seed0 + i)):
  1. msg = msg_len fair bits; oligo = conv_encode(msg); odd reads may be reverse-complemented;
     optional iid substitutions / deletions / insertions (simulator.py:24-26 defaults off here).
  2. flip-flop state path: a base enters flip unless it repeats the previous base, in which
     case the state toggles flip<->flop (supplementary_material.pdf section 2.6).
  3. dwell 1 + Poisson(3.4) blocks per base (first block = the transition, the rest = stays),
     plus one leading stay block.
  4. 40 logits per block ~ N(0, 1.5^2), +margin on the true transition; log-softmax in
     float64, stored as float32.  margin 6 = clean, margin 3 = noisy.
"""
import numpy as np

from .decoder import code_info, encode

FLOP = 4


def reverse_complement_bases(bases):
    """helper.reverse_complement (helper.py:227-229) on 0..3 coded bases: A<->T, C<->G, reversed."""
    return (3 - np.asarray(bases, dtype=np.uint8))[::-1].copy()


def mutate(bases, rng, sub=0.0, dele=0.0, ins=0.0):
    """iid substitution / deletion / insertion channel (helper.simulate_indelsubs, helper.py:34-57)."""
    out = []
    for b in bases:
        r = rng.random()
        if r < sub:
            out.append((int(b) + int(rng.integers(1, 4))) % 4)
        elif r < sub + dele:
            continue
        elif r < sub + dele + ins:
            out.append(int(rng.integers(0, 4)))
            out.append(int(b))
        else:
            out.append(int(b))
    return np.asarray(out, dtype=np.uint8)


def state_path(bases):
    """crf state (0-3 flip, 4-7 flop) per base."""
    states = np.zeros(len(bases), dtype=np.int64)
    prev_state = -1
    for i, b in enumerate(bases):
        b = int(b)
        if prev_state >= 0 and prev_state % 4 == b:
            st = b + FLOP if prev_state < FLOP else b      # repeat: toggle flip <-> flop
        else:
            st = b
        states[i] = st
        prev_state = st
    return states


def transition_index(frm, to):
    """index of transition frm -> to inside a 40-wide posterior block."""
    return to * 8 + frm if to < FLOP else 32 + frm


def posteriors_from_bases(bases, rng, margin=6.0, sigma=1.5, mean_extra_dwell=3.4, quantum=None):
    """bases (0..3) -> float32 [nblk, 40] log-posteriors."""
    states = state_path(bases)
    first = int(states[0])
    lead = (first % 4 + 1 + int(rng.integers(0, 3))) % 4       # a flip state of another base
    true_idx = [transition_index(lead, lead)]                   # one leading stay block
    cur = lead
    for st in states:
        st = int(st)
        true_idx.append(transition_index(cur, st))
        for _ in range(int(rng.poisson(mean_extra_dwell))):
            true_idx.append(transition_index(st, st))
        cur = st
    nblk = len(true_idx)
    logits = rng.normal(0.0, sigma, size=(nblk, 40))
    logits[np.arange(nblk), np.asarray(true_idx)] += margin
    if quantum:
        # tie-stress mode: posteriors on a coarse grid so that exact fp32 score ties are common
        return (np.round((logits - logits.max(axis=1, keepdims=True)) / quantum) * quantum).astype(np.float32)
    mx = logits.max(axis=1, keepdims=True)
    lse = mx + np.log(np.exp(logits - mx).sum(axis=1, keepdims=True))
    return (logits - lse).astype(np.float32)


def make_read(mem_conv, rate, msg_len, seed, rc=False, margin=6.0, sub=0.0, dele=0.0, ins=0.0, quantum=None):
    """-> dict(msg, oligo, read_bases, post, rc)"""
    rng = np.random.default_rng(seed)
    msg = rng.integers(0, 2, size=msg_len, dtype=np.uint8)
    oligo = encode(mem_conv, rate, msg_len, msg)
    seq = reverse_complement_bases(oligo) if rc else oligo
    if sub or dele or ins:
        seq = mutate(seq, rng, sub, dele, ins)
    post = posteriors_from_bases(seq, rng, margin=margin, quantum=quantum)
    return dict(msg=msg, oligo=oligo, read_bases=seq, post=post, rc=bool(rc), seed=seed)


BASES = "ACGT"


def bases_from_str(s):
    return np.array([BASES.index(c) for c in s], dtype=np.uint8)


def make_barcoded_read(mem_conv, rate, msg_len, seed, start_barcode, end_barcode, rc=False, margin=6.0,
                       flank=(10, 40), sub=0.0, dele=0.0, ins=0.0):
    """A read as the sequencer sees it in the reference's experiments (generate_decoded_lists.py:56-84):
    random flank + start barcode + oligo + end barcode + random flank, optionally the reverse-complement
    strand, with the untruncated posterior matrix.  -> dict(msg, oligo, strand, post, rc)"""
    rng = np.random.default_rng(seed)
    msg = rng.integers(0, 2, size=msg_len, dtype=np.uint8)
    oligo = encode(mem_conv, rate, msg_len, msg)
    f5 = rng.integers(0, 4, size=int(rng.integers(flank[0], flank[1] + 1)), dtype=np.uint8)
    f3 = rng.integers(0, 4, size=int(rng.integers(flank[0], flank[1] + 1)), dtype=np.uint8)
    strand = np.concatenate([f5, bases_from_str(start_barcode), oligo, bases_from_str(end_barcode), f3]).astype(np.uint8)
    if rc:
        strand = reverse_complement_bases(strand)
    if sub or dele or ins:
        strand = mutate(strand, rng, sub, dele, ins)
    post = posteriors_from_bases(strand, rng, margin=margin)
    return dict(msg=msg, oligo=oligo, strand=strand, post=post, rc=bool(rc), seed=seed)


def make_reads(mem_conv, rate, msg_len, n, seed0=0, rc_mode="none", margin=6.0, **kw):
    """rc_mode: 'none', 'all', or 'odd' (odd-numbered reads are reverse complements)."""
    code_info(mem_conv, rate, msg_len)    # validates the parameters
    reads = []
    for i in range(n):
        rc = rc_mode == "all" or (rc_mode == "odd" and (i & 1))
        reads.append(make_read(mem_conv, rate, msg_len, seed0 + i, rc=rc, margin=margin, **kw))
    return reads
