"""The whole chain on the GPU box: file -> RS outer code -> index/CRC -> convolutional code -> simulated reads ->
list decoder -> CRC/index filter -> consensus -> RS decode -> file.  The reference's own round-trip check is the
commented-out block at the end of its helper.py (:389-395): encode + simulate_and_decode + filecmp."""
import filecmp
import io
import os

import numpy as np
import pytest

import nanopore_dna_storage_amd as pkg
from nanopore_dna_storage_amd import (compute_error_rate_from_decoded_lists, decode_RS_from_decoded_lists, generate_decoded_lists,
                                      helper, synth)
from golden_util import as_strings, load_case

pytestmark = pytest.mark.gpu


def _data_file(tmp_path, n=1000, seed=5):
    p = tmp_path / "myfile_1K"
    p.write_bytes(bytes(np.random.default_rng(seed).integers(0, 256, size=n, dtype=np.uint8)))
    return str(p)


def test_encode_simulate_and_decode_round_trip(tmp_path):
    """experiment-7 shape in small (supplement 5.2: m=8, rate 3/4, 18 bytes per oligo, 30 % RS, list 8): 1 kB file,
    72 oligos, 400 reads at a signal-to-noise ratio where about half of the reads decode (margin 4.3 at rate 3/4; margin 3
    decodes none, margin 5 all), substitutions / deletions / insertions at simulate_and_decode's defaults, some oligos
    never recovered -> the outer code fills them in and the decoded file equals the input"""
    infile = _data_file(tmp_path)
    out = io.StringIO()
    oligos = helper.encode(data_file=infile, oligo_file=infile + ".oligos", bytes_per_oligo=18, RS_redundancy=0.3, conv_m=8,
                           conv_r=3, pad=False, out=out)
    assert len(oligos) == 56 + 16 and "oligo_len" in out.getvalue()
    conv_in = open(infile + ".oligos.conv_input").read().split()
    assert len(conv_in) == 72 and all(len(b) == 12 + 144 + 8 for b in conv_in)
    r = helper.simulate_and_decode(oligo_file=infile + ".oligos", decoded_data_file=infile + ".decoded", num_reads=400,
                                   data_file_size=1000, bytes_per_oligo=18, RS_redundancy=0.3, conv_m=8, conv_r=3, pad=False,
                                   list_size=8, seed=77, margin=4.3, out=io.StringIO())
    assert filecmp.cmp(infile, infile + ".decoded", shallow=False)
    assert 56 <= r["num_unique"] <= 72
    assert 0.25 * 400 < r["num_success"] < 0.95 * 400 and r["num_attempted"] == 400     # erasures AND successes: the regime the list + CRC + RS chain is for


def test_round_trip_with_pad_bit_and_m6(tmp_path):
    """the reference's commented-out example (helper.py:389-395): 12 bytes per oligo, 100 % redundancy, m=6 rate 1/2"""
    infile = _data_file(tmp_path, n=300, seed=9)
    helper.encode(infile, infile + ".oligos", 12, 1, 6, 1, pad=False, out=io.StringIO())
    helper.simulate_and_decode(infile + ".oligos", infile + ".decoded", 120, 300, 12, 1, 6, 1, pad=False, list_size=4, seed=3,
                               margin=4.5, out=io.StringIO())
    assert filecmp.cmp(infile, infile + ".decoded", shallow=False)


def test_list_files_to_error_rates_and_outer_decode(tmp_path):
    """generate_decoded_lists -> list_<i> files -> compute_error_rate_from_decoded_lists and decode_RS_from_decoded_lists
    (the reference's two consumers of the decoded-lists directory), on DECODER OUTPUT"""
    infile = _data_file(tmp_path, n=400, seed=21)
    oligos = helper.encode(infile, infile + ".oligos", 18, 0.3, 8, 3, out=io.StringIO())
    n_oligos = len(oligos)
    rng = np.random.default_rng(8)
    lists_dir = tmp_path / "lists"
    lists_dir.mkdir()
    rows = []
    n_reads = 3 * n_oligos
    for i in range(n_reads):
        o = oligos[int(rng.integers(n_oligos))]
        rc = bool(rng.integers(2))
        seq = synth.bases_from_str(helper.reverse_complement(o) if rc else o)
        post = synth.posteriors_from_bases(synth.mutate(seq, rng, 0.004, 0.0085, 0.0005), rng, margin=5.0)
        p = tmp_path / ("r%d.post" % i)
        post.tofile(p)
        rows.append("read%d\tref\t%s\t0\t%d\t%d" % (i, p, post.shape[0] - 1, int(rc)))
    man = tmp_path / "manifest.tsv"
    man.write_text("\n".join(rows) + "\n")
    args = generate_decoded_lists.build_parser().parse_args(
        ["--post_manifest", str(man), "--out_prefix", str(lists_dir / "list"), "--info_file", str(tmp_path / "info.txt"),
         "--mem_conv", "8", "--msg_len", "164", "--rate_conv", "3", "--list_size", "8", "--chunk", "50"])
    assert generate_decoded_lists.run(args, out=io.StringIO()) == n_reads
    out = io.StringIO()
    t = compute_error_rate_from_decoded_lists.main(
        ["--list_size", "8", "--decoded_lists_dir", str(lists_dir), "--conv_input_file", infile + ".oligos.conv_input",
         "--bytes_per_oligo", "18"], out=out)
    assert t["num_reads"] == n_reads and t["num_correct"] + t["num_erasure_CRC_index"] + t["num_error_CRC_index"] == n_reads
    assert t["num_correct"] > 0.7 * n_reads and "num_correct: %d" % t["num_correct"] in out.getvalue()
    out = io.StringIO()
    ok = decode_RS_from_decoded_lists.main(
        ["--num_trials", "3", "--list_size", "8", "--num_reads_total", str(n_reads), "--num_reads_to_use", str(int(0.8 * n_reads)),
         "--bytes_per_oligo", "18", "--decoded_lists_dir", str(lists_dir), "--rs_redundancy", "0.3", "--original_file", infile,
         "--seed", "4"], out=out)
    assert ok == 3 and out.getvalue().count("Success") == 3


@pytest.mark.parametrize("name", ["m11_r5_L64", "m11_r5_L64_noisy", "m11_r5_L64_rc"])
def test_config4_list64_with_crc_filter(name):
    """BASELINE configs[4] as written: list_size 64 at m=11 r=5/6 WITH the CRC-8 / index filter -- the GPU's 64-entry
    list through helper.decode_list_CRC_index gives what the reference binary's list (the committed .list) gives"""
    m, post, lines = load_case(name)
    with pkg.Decoder(m["mem_conv"], m["rate"], m["msg_len"], list_size=64, max_deviation=m["max_deviation"], max_slots=1) as dec:
        res = dec.decode([post], rc=[m["rc"]])[0]
    got = as_strings(res[0])
    assert got == lines
    bytes_per_oligo = (m["msg_len"] - helper.index_len - helper.crc_len) // 8
    for num_oligos in (1, 72, 4096):
        assert helper.decode_list_CRC_index(got, bytes_per_oligo, num_oligos, False) == \
            helper.decode_list_CRC_index(lines, bytes_per_oligo, num_oligos, False)
    # a random 180-bit message carries no valid CRC: the filter must reject (nearly) every entry, i.e. act as an erasure
    hits = sum(helper.decode_list_CRC_index([e], bytes_per_oligo, 4096, False)[0] is not None for e in got)
    assert hits <= 3
