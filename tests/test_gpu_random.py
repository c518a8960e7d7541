"""Randomised differential test: GPU decoder vs the CPU oracle over random code parameters,
list sizes, band widths, orientations and noise levels (small trellises, seconds in total)."""
import numpy as np
import pytest

import nanopore_dna_storage_amd as pkg
from nanopore_dna_storage_amd import synth

pytestmark = pytest.mark.gpu

# (mem_conv, rate) pairs with message lengths that terminate on a base boundary
VALID = {(6, 1): [24, 37, 60], (6, 3): [24, 60, 90], (6, 5): [29, 64], (8, 1): [30, 52], (8, 2): [30, 60],
         (8, 3): [20, 44], (8, 4): [30, 60], (8, 5): [27, 62]}


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    keys = sorted(VALID)
    out = []
    while len(out) < n:
        m, r = keys[rng.integers(len(keys))]
        msg_len = int(rng.choice(VALID[(m, r)]))
        try:
            pkg.code_info(m, r, msg_len)
        except pkg.LvaError:
            continue
        L = int(rng.choice([1, 2, 4, 8, 3, 5, 16]))
        md = [None, 3, 6, 10, 20][rng.integers(5)]
        margin = float(rng.choice([2.0, 3.0, 4.0, 6.0]))
        out.append((m, r, msg_len, L, md, margin, int(rng.integers(1 << 30))))
    return out


@pytest.mark.parametrize("case", _cases(24, 2024), ids=lambda c: "m%d_r%d_n%d_L%d_md%s" % c[:5])
def test_random_case(oracle, case):
    m, r, msg_len, L, md, margin, seed = case
    reads = [synth.make_read(m, r, msg_len, seed + i, rc=bool(i & 1), margin=margin,
                             sub=0.01 * (i == 2), dele=0.02 * (i == 2), ins=0.01 * (i == 2)) for i in range(3)]
    with pkg.Decoder(m, r, msg_len, list_size=L, max_deviation=md, max_slots=2) as dec:
        got = dec.decode([x["post"] for x in reads], rc=[x["rc"] for x in reads])
    for x, g in zip(reads, got):
        code = oracle.OracleCode(m, r, msg_len, rc=x["rc"])
        try:
            wm, ws = code.decode(x["post"], L, md, num_threads=16)
        except oracle.OracleError as e:
            assert g == e.status
            continue
        assert not isinstance(g, int)
        assert np.array_equal(g[0], wm)
        assert np.array_equal(g[1].view(np.uint32), ws.view(np.uint32))
