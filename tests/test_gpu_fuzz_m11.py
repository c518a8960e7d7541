"""A slice of scripts/fuzz_m11.py in the driver-run suite: random m=11 / m=14 configurations (code rate,
message length, list size 1-8, band, margin, both orientations) on the GPU against the CPU oracle,
bit for bit -- the regime where the stale band and the list merge decide entries 2..L
(viterbi_convolutional_code.cpp:667-687, :743-800)."""
import numpy as np
import pytest

import nanopore_dna_storage_amd as pkg
from nanopore_dna_storage_amd import synth

pytestmark = pytest.mark.gpu

CFGS = [(11, 1, 40), (11, 2, 61), (11, 5, 100), (11, 5, 180), (11, 1, 90), (14, 1, 20), (14, 7, 58)]


@pytest.mark.parametrize("case", range(6))
def test_random_m11_m14_against_oracle(oracle, case):
    rng = np.random.default_rng(4200 + case)
    m, r, msg_len = CFGS[case % len(CFGS)] if case < 4 else CFGS[int(rng.integers(len(CFGS)))]
    if case == 3:
        m, r, msg_len = 11, 5, 180                      # the benchmark shape is always in the slice
    try:
        pkg.code_info(m, r, msg_len)
    except pkg.LvaError:
        pytest.skip("length does not terminate on a base boundary")
    L = int(rng.choice([2, 4, 8, 8]))
    md = int(rng.choice([20, 20, 10, 5]))
    margin = float(rng.choice([2.5, 3.0, 4.0]))
    seed = int(rng.integers(1 << 30))
    reads = [synth.make_read(m, r, msg_len, seed + i, rc=bool(i & 1), margin=margin) for i in range(2)]
    with pkg.Decoder(m, r, msg_len, list_size=L, max_deviation=md, max_slots=2) as dec:
        got = dec.decode([x["post"] for x in reads], rc=[x["rc"] for x in reads])
    for x, g in zip(reads, got):
        wm, ws = oracle.OracleCode(m, r, msg_len, rc=x["rc"]).decode(x["post"], L, md, num_threads=32)
        assert np.array_equal(g[0], wm), (m, r, msg_len, L, md, margin, seed)
        assert np.array_equal(g[1].view(np.uint32), ws.view(np.uint32)), (m, r, msg_len, L, md, margin, seed)
