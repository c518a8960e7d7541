"""The CPU oracle against the reference's own outputs (golden fixtures): this is what pins the
oracle.  No GPU needed."""
import os
import numpy as np
import pytest

from golden_util import as_strings, encode_cases, load_case, manifest, sync_kw

FAST = [n for n, m in manifest().items() if m["mem_conv"] <= 8 or n in ("m11_r1_L4_short", "m14_r1_L2_short")]
HEAVY = ["m11_r5_L8_noisy_rc"]                      # ~1 min on 8 cores: one full-size read by default
SLOW = [n for n in manifest() if n not in FAST and n not in HEAVY]


def _check(oracle, name, threads=8):
    m, post, lines = load_case(name)
    if m["exit_code"] != 0:
        with pytest.raises(oracle.OracleError) as e:
            oracle.OracleCode(m["mem_conv"], m["rate"], m["msg_len"], rc=m["rc"], **sync_kw(m)).decode(
                post, m["list_size"], m["max_deviation"])
        assert e.value.status == -6 and lines == []
        return
    code = oracle.OracleCode(m["mem_conv"], m["rate"], m["msg_len"], rc=m["rc"], **sync_kw(m))
    msgs, scores = code.decode(post, m["list_size"], m["max_deviation"], num_threads=threads)
    assert as_strings(msgs) == lines
    if not (m.get("nan") or m.get("posinf")):      # (std::sort over NaN scores leaves them wherever its comparisons put them)
        assert np.all(np.diff(scores) <= 0)


@pytest.mark.parametrize("name", FAST)
def test_oracle_matches_reference(oracle, name):
    _check(oracle, name)


@pytest.mark.parametrize("name", HEAVY)
def test_oracle_matches_reference_full_size(oracle, name):
    _check(oracle, name)


@pytest.mark.slow
@pytest.mark.parametrize("name", SLOW)
def test_oracle_matches_reference_slow(oracle, name, request):
    if not request.config.getoption("--runslow"):
        pytest.skip("minutes of CPU and up to 10 GB of memory: run with --runslow")
    _check(oracle, name)


def test_thread_count_does_not_matter(oracle):
    m, post, lines = load_case("m8_r3_L8_edge2")
    code = oracle.OracleCode(m["mem_conv"], m["rate"], m["msg_len"], rc=m["rc"])
    for t in (1, 3):
        assert as_strings(code.decode(post, m["list_size"], m["max_deviation"], num_threads=t)[0]) == lines


def test_unfused_band_start_is_a_different_decoder(oracle):
    """The reference as built by install.sh evaluates the band start with a fused multiply-subtract;
    the oracle follows it (band_fma=True).  The two variants disagree on ~1 step in 1400."""
    code = oracle.OracleCode(8, 3, 164)
    diff = sum(code.band(t, nblk, 20, True) != code.band(t, nblk, 20, False)
               for nblk in range(480, 560) for t in range(nblk))
    assert diff > 0


def test_encoder_matches_reference(oracle):
    for c in encode_cases()["cases"]:
        code = oracle.OracleCode(c["mem_conv"], c["rate"], c["msg_len"])
        for msg, oligo in zip(c["msgs"], c["oligos"]):
            got = code.encode(np.array([int(b) for b in msg], np.uint8))
            assert "".join("ACGT"[b] for b in got) == oligo


def test_bad_parameters_are_refused_like_the_reference(oracle):
    for c in encode_cases()["bad_params"]:
        if c["accepted"]:
            oracle.OracleCode(c["mem_conv"], c["rate"], c["msg_len"])
        else:
            with pytest.raises(oracle.OracleError):
                oracle.OracleCode(c["mem_conv"], c["rate"], c["msg_len"])


def test_oracle_matches_the_reference_binary_on_random_configurations():
    """Beyond the fixtures: 30 random configurations (list sizes 1-70, bands down to max_deviation 1, sync markers, ties, NaN / +inf
    posteriors, truncated reads, refusals) through the unmodified reference binary and through the oracle -- only where the binary
    exists (the build container; scripts/fuzz_oracle_vs_reference.py is the same draw at any size: 2500 cases in round 5)."""
    import subprocess
    import sys
    from oracle import oracle as O
    if not O.have_ref():
        pytest.skip("oracle/_ref/viterbi_nanopore.out not built (needs /root/reference)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_oracle_vs_reference.py"), "7", "30", "--threads", "4"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "checked 30 bad 0" in p.stdout
