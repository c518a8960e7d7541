"""What the gfx950 assembly of the trellis kernels must keep (no GPU needed: hipcc cross-compiles): register budgets that decide
how many wavefronts a CU holds, no scratch memory in the hot kernels, and counted waits in the big-list kernel's output phase
(a load behind a per-lane branch makes the compiler drain the whole memory queue)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("asm") / "lva_k.s")
    src = os.path.join(ROOT, "nanopore_dna_storage_amd", "csrc", "lva_kernels.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--cuda-device-only",
                    "-S", "-o", out, src], check=True, cwd=os.path.dirname(src))
    return open(out).read()


def _meta(asm):
    res = {}
    for b in asm.split("- .agpr_count:")[1:]:
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, b).group(1)
        res[g("name")] = dict(vgpr=int(g("vgpr_count")), lds=int(g("group_segment_fixed_size")), scratch=int(g("private_segment_fixed_size")))
    return res


def test_register_and_lds_budgets(asm):
    meta = _meta(asm)
    pick = lambda pat: {k: v for k, v in meta.items() if re.search(pat, k)}
    lazy = pick(r"lva_step_lazyILi8ELi3ELb")                       # the benchmark's two instances: four 512-thread workgroups per CU
    assert len(lazy) == 2
    for k, v in lazy.items():
        assert v["vgpr"] <= 64 and v["lds"] <= 40 * 1024 and v["scratch"] <= 16, (k, v)
    rec = pick(r"lva_step_big_recILi64E")                          # configs[4]: three 256-thread workgroups per CU
    assert len(rec) == 1
    for k, v in rec.items():
        assert v["vgpr"] <= 168 and v["lds"] <= 53 * 1024 and v["scratch"] == 0, (k, v)
    for k, v in pick(r"lva_step_acs").items():
        assert v["scratch"] == 0, (k, v)


def test_big_list_output_phase_requests_a_round_at_once(asm):
    """the records of a round are requested by all lanes in straight-line code, back to back, before anything is waited for --
    not one drain of the memory queue (s_waitcnt vmcnt(0)) behind every entry's loads, as the compiler emits for loads that sit
    behind per-lane branches"""
    m = re.search(r"^(_ZN3lva16lva_step_big_recILi64E\S*):", asm, re.M)
    body = asm[m.start():asm.index(".Lfunc_end", m.start())]
    lines = [ln.strip() for ln in body.split("\n") if re.match(r"\s+(global_load_dwordx[24]|s_waitcnt vmcnt)", ln)]
    run = best = 0
    for ln in lines:
        run = run + 1 if ln.startswith("global_load") else 0
        best = max(best, run)
    assert best >= 8, best          # four entries x (16 + 16 bytes) at three message planes


@pytest.mark.parametrize("inst", ["Lb0", "Lb1"])
def test_lazy_first_phase_is_one_round_trip(asm, inst):
    """round 6: every staging request of a thread -- four 16-byte chunks for 8 rows of 8 entries, at an anchor step the two words of
    back-pointer bytes, the posteriors -- is issued before the first LDS write waits; a loop over a run-time count would be compiled
    as load, s_waitcnt vmcnt(0), write, next (seven round trips in a row in front of the barrier, as it was until then).  And the
    merge loop finds a duplicate with unsigned minima over the tagged fingerprints, not with eight compare + select pairs."""
    m = re.search(r"^(_ZN3lva13lva_step_lazyILi8ELi3E%s\S*):" % inst, asm, re.M)
    body = asm[m.start():asm.index(".Lfunc_end", m.start())]
    head = body[:body.index("s_barrier")]
    # the four chunks are straight-line loads (two shared with the 4-row shape of a compact source position, two more for 8 rows)
    assert len(re.findall(r"^\s+global_load_dwordx4", head, re.M)) == 4
    # (the loop that was there held ONE such load)
    assert body.count("v_min3_u32") >= 6                                     # three per merge loop (8 lists, 2 lists)
