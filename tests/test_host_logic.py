"""Host-side logic that needs no GPU: helper functions with known answers, the CLI shim's
encode mode and error behaviour, the C ABI's export list, and the no-fallback rule."""
import ctypes
import io
import os
import re

import numpy as np
import pytest

import nanopore_dna_storage_amd as pkg
from nanopore_dna_storage_amd import _lib, helper, synth, viterbi_nanopore
from golden_util import encode_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "lva_decoder.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(lva_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS)
    L = ctypes.CDLL(pkg.library_path())
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in pkg.load_library().lva_version()


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.LvaError) as e:
        pkg.Decoder(6, 1, 60)
    assert e.value.code == -11


def test_crc8_known_answers():
    assert helper.crc8(b"123456789") == b"\xf4"        # CRC-8 (poly 0x07) check value
    assert helper.crc8(b"") == b"\x00"
    assert helper.crc8(b"\x00\x00") == b"\x00"
    assert helper.crc8(b"\x01") == b"\x07"


def test_bit_byte_conversions():
    assert helper.bitstring2bytestring("1", 8) == b"\x01"
    assert helper.bitstring2bytestring("110000000001", 16) == b"\x0c\x01"     # left zero padding (helper.py:378-379)
    assert helper.bytestring2bitstring(b"\x0c\x01", 16) == "0000110000000001"
    s = "1011001110001111"
    assert helper.bytestring2bitstring(helper.bitstring2bytestring(s, 16), 16) == s


def test_reverse_complement():
    assert helper.reverse_complement("AACGTN") == "NACGTT"
    bases = np.array([0, 0, 1, 2, 3], np.uint8)
    assert pkg.bases_to_str(synth.reverse_complement_bases(bases)) == helper.reverse_complement(pkg.bases_to_str(bases))
    with pytest.raises(KeyError):
        helper.reverse_complement("AXG")


def test_truncate_post_file(tmp_path):
    post = np.arange(10 * 40, dtype=np.float32).reshape(10, 40)
    a, b = tmp_path / "a.post", tmp_path / "b.post"
    post.tofile(a)
    helper.truncate_post_file(str(a), str(b), 2, 5)
    got = helper.read_post_file(str(b))
    assert np.array_equal(got, post[2:6])
    assert os.path.getsize(b) == 4 * 160
    assert np.array_equal(helper.truncate_post(post, 2, 5), post[2:6])
    with pytest.raises(AssertionError):
        helper.truncate_post_file(str(a), str(b), 5, 10)


def test_read_post_file_partial_block(tmp_path):
    """read_crf_post (:553-575) completes a trailing partial block with the last value read"""
    vals = np.arange(45, dtype=np.float32)
    f = tmp_path / "p.post"
    vals.tofile(f)
    got = helper.read_post_file(str(f))
    assert got.shape == (2, 40)
    assert np.array_equal(got[1, :5], vals[40:]) and np.all(got[1, 5:] == vals[-1])


def test_crc_index_filter_roundtrip():
    payload = bytes(range(18))
    entry = helper.attach_index_crc(37, payload)
    assert len(entry) == 12 + 8 * 18 + 8
    bad = "".join("1" if c == "0" else "0" for c in entry[:3]) + entry[3:]
    idx, pay, msg = helper.decode_list_CRC_index([bad, entry], 18, 100, False)
    assert (idx, pay, msg) == (37, payload, entry)
    assert helper.decode_list_CRC_index([bad], 18, 100, False) == (None, None, None)
    assert helper.decode_list_CRC_index([entry], 18, 30, False) == (None, None, None)     # index out of range
    padded = helper.attach_index_crc(5, payload, pad=True)
    assert helper.decode_list_CRC_index([padded], 18, 100, True)[0] == 5
    t = helper.tally_decoded_lists([[entry], [bad]], {37: entry} and [entry if i == 37 else "" for i in range(100)], 18, False, 8)
    assert t == dict(num_reads=2, num_correct=1, num_erasure_CRC_index=1, num_error_CRC_index=0)


def test_distances():
    assert helper.hamming("10110", "10011") == 2
    assert helper.levenshtein("kitten", "sitting") == 3
    assert helper.levenshtein("", "abc") == 3
    with pytest.raises(ValueError):
        helper.hamming("1", "11")


def test_compute_parameters():
    assert helper.compute_parameters(18, 0.3, 180, False) == (164, 10, 3, 13)
    assert helper.compute_parameters(20, 0.0, 40, True)[0] == 12 + 8 + 160 + 1


def test_cli_encode_matches_reference(tmp_path):
    c = [x for x in encode_cases()["cases"] if (x["mem_conv"], x["rate"]) == (11, 5)][0]
    fin, fout = tmp_path / "in.txt", tmp_path / "out.txt"
    fin.write_text("".join(m + "\n" for m in c["msgs"]))
    rc = viterbi_nanopore.main(["-m", "encode", "-i", str(fin), "-o", str(fout), "--mem-conv", "11", "-r", "5",
                                "--msg-len", str(c["msg_len"]), ""], out=io.StringIO())
    assert rc == 0
    assert fout.read_text().split() == c["oligos"]


def test_cli_error_behaviour(tmp_path):
    fin = tmp_path / "in.txt"
    fin.write_text("0101\n")
    out = io.StringIO()
    base = ["-i", str(fin), "-o", str(tmp_path / "o")]
    assert viterbi_nanopore.main(["-m", "encode"] + base + ["--msg-len", "4"], out=out) == 255
    assert "Memory of convolutional code not specified." in out.getvalue()
    out = io.StringIO()
    assert viterbi_nanopore.main(["-m", "encode"] + base + ["--mem-conv", "7", "--msg-len", "4"], out=out) == 255
    assert "Invalid mem_conv" in out.getvalue()
    out = io.StringIO()
    assert viterbi_nanopore.main(["-m", "encode"] + base + ["--mem-conv", "6", "-r", "3", "--msg-len", "5"], out=out) == 255
    assert "Output length not even" in out.getvalue()
    out = io.StringIO()
    assert viterbi_nanopore.main(["-m", "transcode"] + base + ["--mem-conv", "6", "--msg-len", "4"], out=out) == 255
    assert "Invalid mode." in out.getvalue()
    out = io.StringIO()
    assert viterbi_nanopore.main(["-m", "encode"] + base + ["--mem-conv", "6", "--msg-len", "6"], out=out) == 255
    assert "Message length does not match" in out.getvalue()
    assert viterbi_nanopore.main(["-h"], out=io.StringIO()) == 0
    assert not os.path.exists(tmp_path / "o")


def test_synthetic_generator_is_deterministic_and_normalised():
    a = synth.make_read(6, 1, 60, seed=5, rc=True, margin=4.0)
    b = synth.make_read(6, 1, 60, seed=5, rc=True, margin=4.0)
    assert np.array_equal(a["post"], b["post"]) and np.array_equal(a["msg"], b["msg"])
    assert a["post"].dtype == np.float32 and a["post"].shape[1] == 40
    lse = np.log(np.exp(a["post"].astype(np.float64)).sum(axis=1))
    assert np.allclose(lse, 0, atol=1e-5)
    assert a["post"].shape[0] >= len(a["read_bases"]) + 1
    st = synth.state_path([0, 0, 0, 1, 1, 2])
    assert list(st) == [0, 4, 0, 1, 5, 2]
    assert synth.transition_index(3, 1) == 1 * 8 + 3 and synth.transition_index(2, 6) == 34 and synth.transition_index(6, 6) == 38


def test_reference_path_launcher_encodes(tmp_path):
    """viterbi/viterbi_nanopore.out (the path the reference's scripts spawn) with -m encode, as simulator.py:67 calls it"""
    import os
    import subprocess
    from golden_util import encode_cases
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    c = encode_cases()["cases"][0]
    fin, fout = tmp_path / "msg.txt", tmp_path / "oligo.txt"
    fin.write_text("".join(m + "\n" for m in c["msgs"]))
    p = subprocess.run([os.path.join(root, "viterbi", "viterbi_nanopore.out"), "-m", "encode", "-i", str(fin), "-o", str(fout),
                        "--mem-conv", str(c["mem_conv"]), "--msg-len", str(c["msg_len"]), "-r", str(c["rate"])],
                       stdout=subprocess.PIPE, text=True, cwd=str(tmp_path))
    assert p.returncode == 0
    assert fout.read_text().split() == c["oligos"]
    p = subprocess.run([os.path.join(root, "viterbi", "viterbi_nanopore.out"), "-m", "encode", "-i", str(fin), "-o", str(fout),
                        "--mem-conv", "7", "--msg-len", "180"], stdout=subprocess.PIPE, text=True, cwd=str(tmp_path))
    assert p.returncode == 255 and "Invalid mem_conv" in p.stdout


def test_crc_index_filter_against_reference_vectors():
    """tests/golden/crc_index_cases.json: outputs of the REFERENCE's helper.py functions (imported in the build
    container by make_crc_index_cases.py; the crc8 package itself is stubbed there -- its polynomial stays unpinned)"""
    import json
    import os
    from nanopore_dna_storage_amd import helper
    here = os.path.dirname(os.path.abspath(__file__))
    g = json.load(open(os.path.join(here, "golden", "crc_index_cases.json")))
    c = g["constants"]
    assert (helper.prp_a, helper.prp_b, helper.prp_a_inv, helper.index_len, helper.crc_len) == \
        (c["prp_a"], c["prp_b"], c["prp_a_inv"], c["index_len"], c["crc_len"])
    for x in g["bit_byte"]:
        b = helper.bitstring2bytestring(x["bits"], x["nbits"])
        assert b.hex() == x["bytes_hex"] and helper.bytestring2bitstring(b, x["nbits"]) == x["back"]
    hits = 0
    for x in g["filter"]:
        index, payload, entry = helper.decode_list_CRC_index(x["list"], x["bytes_per_oligo"], x["num_oligos"], x["pad"])
        assert index == x["index"] and entry == x["entry"]
        assert (None if payload is None else payload.hex()) == x["payload_hex"]
        hits += index is not None
    assert 10 < hits < len(g["filter"])
    for x in g["parameters"]:
        assert list(helper.compute_parameters(x["bytes_per_oligo"], x["RS_redundancy"], x["data_size_padded"], x["pad"])) == x["result"]


def test_error_rate_script_on_list_files(tmp_path):
    """compute_error_rate_from_decoded_lists.py:18-56 over list_<i> files: correct / erased / wrong-CRC-match tallies (host only)"""
    import io
    from nanopore_dna_storage_amd import compute_error_rate_from_decoded_lists as cer
    payloads = [bytes([i] * 4) for i in range(3)]
    conv_in = [helper.attach_index_crc(i, p) for i, p in enumerate(payloads)]
    (tmp_path / "conv_input.txt").write_text("\n".join(conv_in) + "\n")
    d = tmp_path / "lists"
    d.mkdir()
    junk = "0" * len(conv_in[0])
    (d / "list_0").write_text(junk + "\n" + conv_in[0] + "\n")                 # second entry passes: correct
    (d / "list_1").write_text(junk + "\n")                                     # nothing passes: erasure
    (d / "list_2").write_text(helper.attach_index_crc(1, bytes([9] * 4)) + "\n")   # valid CRC and index, wrong payload: error
    (d / "other.txt").write_text("ignored\n")
    out = io.StringIO()
    t = cer.main(["--list_size", "8", "--decoded_lists_dir", str(d), "--conv_input_file", str(tmp_path / "conv_input.txt"),
                  "--bytes_per_oligo", "4"], out=out)
    assert t == dict(num_reads=3, num_correct=1, num_erasure_CRC_index=1, num_error_CRC_index=1)
    assert "num_oligos 3" in out.getvalue() and "num_error_CRC_index: 1" in out.getvalue()
    # list_size 1: the correct entry of list_0 is out of reach
    t1 = cer.main(["--list_size", "1", "--decoded_lists_dir", str(d), "--conv_input_file", str(tmp_path / "conv_input.txt"),
                   "--bytes_per_oligo", "4"], out=io.StringIO())
    assert t1["num_correct"] == 0 and t1["num_erasure_CRC_index"] == 2


def test_list_files_are_written_atomically(tmp_path):
    """generate_decoded_lists writes OUT_PREFIX_i under a temporary name and renames it: no partial file can pass for a finished read"""
    from nanopore_dna_storage_amd import generate_decoded_lists as gdl
    p = tmp_path / "list_7"
    gdl.write_list_file(str(p), np.array([[1, 0, 1], [0, 0, 1]], np.uint8))
    assert p.read_text() == "101\n001\n"
    assert [f.name for f in tmp_path.iterdir()] == ["list_7"]
    # a temporary file a killed run left behind never looks like a list file (list consumers take list_<digits> only) and is
    # removed by the next run
    import subprocess
    dead = subprocess.Popen(["true"]); dead.wait()                     # a process id that no longer exists
    stale = tmp_path / (gdl.TMP_PREFIX + "list_9-%s-%d" % (gdl._HOST, dead.pid))
    stale.write_text("10")
    other = tmp_path / (gdl.TMP_PREFIX + "list_b_3-%s-%d" % (gdl._HOST, dead.pid))     # another prefix that shares the stem: not ours to delete
    other.write_text("10")
    alive = tmp_path / (gdl.TMP_PREFIX + "list_4-%s-%d" % (gdl._HOST, os.getppid()))   # a concurrent run on the same prefix (its writer is alive)
    alive.write_text("10")
    from nanopore_dna_storage_amd import compute_error_rate_from_decoded_lists as cer
    (tmp_path / "list_7.bak").write_text("1\n")
    assert [n for n, _ in cer.read_lists(str(tmp_path))] == ["list_7"]
    gdl.remove_stale_temp_files(str(tmp_path / "list"))
    assert not stale.exists() and p.exists() and other.exists() and alive.exists()
    args = gdl.build_parser().parse_args(["--post_manifest", "m", "--out_prefix", "o", "--info_file", "i", "--mem_conv", "6", "--msg_len", "60",
                                          "--rate_conv", "1", "--list_size", "4"])
    assert args.chunk == 4096 and args.max_deviation == 20 and args.gpus == 1


@pytest.mark.parametrize("m,r,msg_len,md,rc", [(6, 1, 60, 20, False), (6, 1, 24, 3, True), (8, 3, 164, 20, False), (11, 5, 180, 20, True),
                                               (8, 3, 44, 1, False), (6, 5, 180, None, False), (14, 7, 180, 5, False)])
def test_working_band_keeps_every_state_that_can_matter(oracle, m, r, msg_len, md, rc):
    """The kernels work on the reference's band (:677-679) minus the positions a path cannot have reached yet (> t + 1) and the
    positions that cannot reach the final one any more (lva_api.cpp working_band).  Property: the reference band is reproduced
    exactly (oracle), the working band lies inside it, and every (step, position) cell that lies on SOME monotone path
    (stay or +1 position per step, every cell inside the reference band) from (step 0, position <= 1) to (last step, last
    position) is still inside the working band -- for several read lengths, incl. the shortest the reference accepts."""
    info = pkg.code_info(m, r, msg_len, rc)
    npos = info.nstate_pos
    code = oracle.OracleCode(m, r, msg_len, rc=rc)
    for nblk in (npos + 1, npos + 2, 2 * npos + 3, int(4.4 * npos), int(4.4 * npos) + 1):
        ref, work = pkg.band_table(m, r, msg_len, nblk, md, rc=rc)
        want = np.array([code.band(t, nblk, md if md is not None else msg_len + m + 1) for t in range(nblk)], dtype=np.int64)
        assert np.array_equal(ref, want), "reference band differs from the oracle's"
        assert np.all(ref[:, 0] <= work[:, 0]) and np.all(work[:, 1] <= ref[:, 1]) and np.all(work[:, 0] <= work[:, 1])
        # forward: cells reachable from the start inside the reference band; backward: cells that reach the final cell
        inb = np.zeros((nblk, npos), bool)
        for t in range(nblk):
            inb[t, ref[t, 0]:ref[t, 1]] = True
        fwd = np.zeros((nblk, npos), bool)
        fwd[0, :2] = inb[0, :2]                                   # after step 0: position 0 (stay) and position 1 (one move)
        for t in range(1, nblk):
            fwd[t] = inb[t] & (fwd[t - 1] | np.concatenate([[False], fwd[t - 1][:-1]]))
        bwd = np.zeros((nblk, npos), bool)
        bwd[nblk - 1, npos - 1] = inb[nblk - 1, npos - 1]
        for t in range(nblk - 2, -1, -1):
            bwd[t] = inb[t] & (bwd[t + 1] | np.concatenate([bwd[t + 1][1:], [False]]))
        need = fwd & bwd
        for t in range(nblk):
            p = np.nonzero(need[t])[0]
            assert p.size == 0 or (work[t, 0] <= p.min() and p.max() < work[t, 1]), (nblk, t, p.min(), p.max(), work[t])
        # The stale row: the buffers are never cleared, so the lowest band position p = lo(t) reads row p - 1 of the
        # other parity buffer as it was LAST written -- at step t - 1 if the band held p - 1 then, else at the last step t' of
        # t - 1's parity whose band did.  The kernels walk the working band: wherever the consumer cell can matter, the row they
        # find must have been written by the same step as the reference's, or hold nothing in the reference either (a position
        # no path had reached when it was written: -inf lists, which the kernels neither write nor read).
        def last_writer(band, t, pos):
            for tt in range(t - 1, -1, -2):
                if band[tt, 0] <= pos < band[tt, 1]:
                    return tt
            return None
        for t in range(1, nblk):
            pw = int(work[t, 0])
            if pw < 1 or pw >= work[t, 1] or not need[t, pw]:
                continue
            t_ref, t_k = last_writer(ref, t, pw - 1), last_writer(work, t, pw - 1)
            if t_ref is None:
                assert t_k is None
            elif t_k != t_ref:
                assert t_k is None and pw - 1 > t_ref + 1, (nblk, t, pw, t_ref, t_k)
        # and it really is smaller where it can be
        assert work[0, 1] <= 2 and work[nblk - 1, 0] >= min(npos - 1, ref[nblk - 1, 1])


@pytest.mark.parametrize("m,r,rc", [(11, 5, False), (11, 5, True), (11, 1, False), (14, 7, False), (14, 7, True), (8, 3, False)])
def test_xcd_tile_order_is_a_bijection_and_pairs_the_two_readers_of_a_row(m, r, rc):
    """The L = 1 kernel's tile order (csrc/lva_kernels.hip xcd_tile, csrc/lva_api.cpp upload_codes `chain`), restated: the workgroup
    with blockIdx.x = x takes the tile that carries x & 7 -- the label of its XCD under round-robin dispatch -- in bits xs .. xs+2.
    Whatever xs, that is a bijection of the tiles (correctness rests on nothing else); and where xs[p] = xs[p+1] + (bits of the
    step into p) the workgroup that reads a 128-byte row of position p as its targets' stay entries and the workgroup at p + 1 that
    stages the same row as source entries carry the same label."""
    from nanopore_dna_storage_amd import decoder as dec
    tab = dec.code_tables(m, r, {5: 180, 1: 100, 7: 182, 3: 164}[r], rc)
    ptype = tab["ptype"]
    npos, N = len(ptype), 1 << m
    tile_bits = m - 6
    G = 1 << max(tile_bits, 0)

    def xcd_tile(x, xs):
        rest, lab = x >> 3, x & 7
        return ((rest >> xs) << (xs + 3)) | (lab << xs) | (rest & ((1 << xs) - 1))

    xs = [0] * npos
    if tile_bits >= 4:
        smax = s = min(tile_bits - 3, 2)
        for p in range(1, npos):
            xs[p] = s
            sh = 1 if ptype[p] == 0 else 2
            s = s - sh if s >= sh else smax
    else:
        assert G < 16                                   # fewer than 16 tiles: the plain order
    for s in set(xs):
        assert sorted(xcd_tile(x, s) for x in range(G)) == list(range(G))
    if tile_bits < 4:
        return
    label = {}                                          # (pos, tile) -> x & 7 of the workgroup that takes it
    for p in range(1, npos):
        for x in range(G):
            label[(p, xcd_tile(x, xs[p]))] = x & 7
    paired = 0
    for p in range(1, npos - 1):
        sh = 1 if ptype[p] == 0 else 2
        if xs[p] != xs[p + 1] + sh:
            continue
        paired += 1
        Tn = 64 >> sh
        for q in range(N // 16):                        # a row piece of 16 conv states = one 128-byte line of 8-byte entries
            c = 16 * q
            stay_tile = (c % (N >> sh)) // Tn           # tile_target: c = tile * Tn + i + x * (N >> sh)
            source_tile = c // 64                       # staged by the workgroup whose 64 source conv states contain c
            assert label[(p, stay_tile)] == label[(p + 1, source_tile)], (p, q)
    assert paired >= (npos - 2) // 3
