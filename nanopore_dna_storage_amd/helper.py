"""Host-side helpers around the decode path, mirroring the parts of the reference's helper.py
that sit next to the decoder call (helper.py:305): post truncation (helper.py:211-224), reverse
complement (:227-229), bit/byte conversions (:365-369) and the CRC-8 / index filter over a
decoded list (:371-388), and the two end-to-end drivers `encode` (:231-273) and `simulate_and_decode`
(:275-350) on top of the RS outer code (rs_code.py), the convolutional encoder (lva_encode) and the list
decoder.  The reference's signal simulation and fast5 writing need scrappy / fast5_research and are
replaced by synth.py's posterior generator.

CRC-8: the reference calls the PyPI module `crc8` (unpinned, install_python_packages.sh:3), which
is not installable here.  Its algorithm -- polynomial x^8+x^2+x+1 (0x07), init 0, no reflection,
no final xor -- is restated below and pinned by the standard check value crc8("123456789") = 0xF4.
"""
import math

import numpy as np

# PRP x -> a*x+b mod 2^12 that randomises the oligo index (helper.py:28-32)
prp_a = 1751
prp_b = 2532
prp_a_inv = 3303
index_len = 12
crc_len = 8

BYTES_PER_BLOCK = 160      # 40 float32 per flappie block (helper.py:211-216)

_COMPLEMENT = str.maketrans("ACGTN", "TGCAN")


def reverse_complement(dna):
    """helper.py:227-229 (raises KeyError-like ValueError on letters outside ACGTN, as the dict lookup does)."""
    if any(ch not in "ACGTN" for ch in dna):
        raise KeyError(next(ch for ch in dna if ch not in "ACGTN"))
    return dna.translate(_COMPLEMENT)[::-1]


def read_seq(path):
    with open(path) as f:
        return f.readline().rstrip("\n")


def find_barcode_pos(basecall, trans_arr, start_barcode, end_barcode):
    """The search of find_barcode_pos_in_post (helper.py:173-210) on an in-memory basecall and list of
    transition positions.  Host-side mirror of the device path (Decoder.find_barcode / locate_payload)."""
    inf = float("inf")
    n, ls, le = len(basecall), len(start_barcode), len(end_barcode)
    if ls + le > n:
        return (-1, -1, inf, inf)
    sd = [levenshtein(start_barcode, basecall[i:i + ls]) for i in range(n // 2 + 1 - ls)]
    ed = [levenshtein(end_barcode, basecall[i:i + le]) for i in range(n // 2, n - le)]
    if not sd or not ed:        # the reference raises here (min of an empty list)
        return (-1, -1, inf, inf)
    s_first = sd.index(min(sd))
    e_first = n // 2 + ed.index(min(ed))
    start_pos = int(trans_arr[s_first + ls]) - 1
    end_pos = int(trans_arr[e_first - 1]) - 1
    if end_pos < start_pos:
        return (-1, -1, inf, inf)
    return (start_pos, end_pos, min(sd), min(ed))


def find_barcode_pos_in_post(trans_filename, fastq_filename, start_barcode, end_barcode, decoder=None):
    """helper.py:157-210: position of the payload in the posterior matrix from flappie's fastq (second line =
    basecall) and --trans-output-file.  -> (start_pos, end_pos, start distance, end distance), both positions
    inclusive; (-1, -1, inf, inf) on failure.  With `decoder` (a Decoder) the search runs on its GPU."""
    with open(fastq_filename) as f:
        f.readline()
        basecall = f.readline().rstrip("\n")
    with open(trans_filename) as f:
        trans_arr = [int(x) for x in f.read().split()]
    if decoder is None:
        return find_barcode_pos(basecall, trans_arr, start_barcode, end_barcode)
    r = decoder.find_barcode([basecall], [trans_arr], start_barcode, end_barcode)[0]
    return (r["start_pos"], r["end_pos"], r["dist_start"], r["dist_end"])


def truncate_post(post, start_pos, end_pos):
    """rows [start_pos, end_pos] (inclusive) of a [nblk, 40] posterior matrix"""
    post = np.asarray(post, dtype=np.float32).reshape(-1, 40)
    assert end_pos >= start_pos
    assert post.shape[0] >= end_pos + 1
    return post[start_pos:end_pos + 1]


def truncate_post_file(old_post_filename, new_post_filename, start_pos, end_pos, bytes_per_blk=BYTES_PER_BLOCK):
    """helper.py:211-224: byte-slice blocks [start_pos, end_pos] of a .post file into a new file."""
    with open(old_post_filename, "rb") as f:
        data = f.read()
    assert len(data) % bytes_per_blk == 0
    assert end_pos >= start_pos
    assert len(data) >= (end_pos + 1) * bytes_per_blk
    with open(new_post_filename, "wb") as f:
        f.write(data[start_pos * bytes_per_blk:(end_pos + 1) * bytes_per_blk])


def read_post_file(path):
    """read_crf_post (viterbi_convolutional_code.cpp:553-575) -> float32 [nblk, 40].
    Like the reference, a trailing partial block is completed with the last value read."""
    raw = np.fromfile(path, dtype="<f4")
    full, rem = divmod(raw.size, 40)
    if rem == 0:
        return raw.reshape(full, 40)
    tail = np.full(40, raw[-1], dtype=np.float32)
    tail[:rem] = raw[full * 40:]
    return np.concatenate([raw[:full * 40].reshape(full, 40), tail[None, :]], axis=0)


def bitstring2bytestring(bitstring, bitstring_len):
    """helper.py:365-366: value of the bit string, left-padded with zeros to bitstring_len bits."""
    return int(bitstring, 2).to_bytes(bitstring_len // 8, "big")


def bytestring2bitstring(bytestring, bitstring_len):
    """helper.py:368-369"""
    return format(int.from_bytes(bytestring, "big"), "b").zfill(bitstring_len)


_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = ((_c << 1) ^ 0x07) & 0xFF if _c & 0x80 else (_c << 1) & 0xFF
    _CRC_TABLE.append(_c)


def crc8(data):
    """CRC-8 (poly 0x07, init 0x00, unreflected, xorout 0) -> bytes of length 1, like crc8.crc8(...).digest()"""
    c = 0
    for b in data:
        c = _CRC_TABLE[c ^ b]
    return bytes([c])


def compute_parameters(bytes_per_oligo, RS_redundancy, data_size_padded, pad):
    """helper.py:352-363 (without the prints)"""
    msg_len = index_len + crc_len + 8 * bytes_per_oligo + int(bool(pad))
    assert data_size_padded % bytes_per_oligo == 0
    num_oligos_data = data_size_padded // bytes_per_oligo
    num_oligos_RS = int(num_oligos_data * RS_redundancy)
    return msg_len, num_oligos_data, num_oligos_RS, num_oligos_data + num_oligos_RS


def attach_index_crc(index, payload, pad=False):
    """The bit string helper.encode writes per oligo (helper.py:253-262): PRP(index) | payload | CRC-8 [| 0]."""
    index_prp = (prp_a * index + prp_b) % (2 ** index_len)
    bits = format(index_prp, "b").zfill(index_len)
    index_bytes = bitstring2bytestring(bits, 8 * math.ceil(index_len / 8))
    crc = crc8(index_bytes + payload)
    out = bits + bytestring2bitstring(payload + crc, 8 * len(payload) + crc_len)
    return out + "0" if pad else out


def decode_list_CRC_index(decoded_msg_list, bytes_per_oligo, num_oligos, pad):
    """helper.py:371-388: first list entry whose CRC-8 checks and whose de-randomised index is in
    range -> (index, payload_bytes, entry); (None, None, None) when no entry qualifies."""
    for entry in decoded_msg_list:
        msg = entry[:-1] if pad else entry
        nbits = math.ceil(len(msg) / 8) * 8
        as_bytes = bitstring2bytestring(msg, nbits)
        if crc8(as_bytes[:-crc_len // 8]) != as_bytes[-crc_len // 8:]:
            continue
        nidx = math.ceil(index_len / 8)
        idx_bits = bytestring2bitstring(as_bytes[:nidx], 8 * nidx)[-index_len:]
        index = (prp_a_inv * (int(idx_bits, 2) - prp_b)) % (2 ** index_len)
        payload = bitstring2bytestring(msg[index_len:-crc_len], bytes_per_oligo * 8)
        if index < num_oligos:
            return index, payload, entry
    return None, None, None


def encode(data_file, oligo_file, bytes_per_oligo, RS_redundancy, conv_m, conv_r, pad=False, device=0, out=None):
    """helper.py:231-273, same arguments and files: data_file -> segments of bytes_per_oligo bytes (padded with b'0')
    -> RS outer code over the segments (rs_code.MainEncoder: csrc/rs_kernels.hip) -> per oligo PRP(index) | payload |
    CRC-8 [| pad bit] written to oligo_file + '.conv_input' -> convolutional code (lva_encode, the `-m encode` of the
    reference binary) -> oligo_file, one ACGT string per line.  -> the list of oligo strings."""
    import sys
    from . import rs_code
    from .decoder import encode as conv_encode
    out = sys.stdout if out is None else out
    assert bytes_per_oligo % 2 == 0
    assert conv_m in [6, 8, 11, 14]
    assert conv_r in [1, 2, 3, 4, 5, 7]
    with open(data_file, "rb") as f:
        data = f.read()
    data_size = len(data)
    data_size_padded = math.ceil(data_size / bytes_per_oligo) * bytes_per_oligo
    msg_len, num_oligos_data, num_oligos_RS, num_oligos = compute_parameters(bytes_per_oligo, RS_redundancy, data_size_padded, pad)
    data_padded = data.ljust(data_size_padded, b"0")
    segmented_data = [data_padded[i * bytes_per_oligo:(i + 1) * bytes_per_oligo] for i in range(num_oligos_data)]
    with_rs = rs_code.MainEncoder(segmented_data, num_oligos_RS, device=device) if num_oligos_RS else segmented_data
    bit_strings = [attach_index_crc(index, oligo, pad) for index, oligo in enumerate(with_rs)]
    with open(oligo_file + ".conv_input", "w") as f:
        for b in bit_strings:
            f.write(b + "\n")
    msgs = np.array([[int(ch) for ch in b] for b in bit_strings], dtype=np.uint8)
    bases = conv_encode(conv_m, conv_r, msg_len, msgs)
    oligos = ["".join("ACGT"[int(x)] for x in row) for row in np.atleast_2d(bases)]
    with open(oligo_file, "w") as f:
        for o in oligos:
            f.write(o + "\n")
    print("oligo_len", len(oligos[0]), file=out)
    print("writing rate (bits per base):", data_size * 8 / (len(oligos[0]) * num_oligos), file=out)
    return oligos


def simulate_and_decode(oligo_file, decoded_data_file, num_reads, data_file_size, bytes_per_oligo, RS_redundancy, conv_m, conv_r,
                        pad=False, syn_sub_prob=0.005, syn_del_prob=0.005, syn_ins_prob=0.0005, deepsimdwell=False, num_thr=16,
                        list_size=1, seed=None, margin=6.0, device=0, out=None, decoder=None):
    """helper.py:275-350, same arguments: num_reads times pick a random oligo and orientation, pass it through the
    substitution / deletion / insertion channel, turn it into a posterior matrix, decode a list of list_size candidates,
    take the first candidate whose CRC-8 checks and whose index is in range (the first payload seen for an index stands,
    :326-329), then RS-decode the indices that arrived and write the first data_file_size bytes.
    The signal simulator and the basecaller network of the reference (scrappy, flappie) are replaced by
    synth.posteriors_from_bases (`margin`: how far the true transition stands out); all reads are decoded in ONE batch
    on the GPU instead of one decoder process per read.  seed: numpy seed (the reference draws from the global state).
    -> dict(num_attempted, num_success, num_unique, decoded bytes)."""
    import sys
    from . import rs_code, synth
    from .decoder import Decoder
    out = sys.stdout if out is None else out
    data_size_padded = math.ceil(data_file_size / bytes_per_oligo) * bytes_per_oligo
    msg_len, num_oligos_data, num_oligos_RS, num_oligos = compute_parameters(bytes_per_oligo, RS_redundancy, data_size_padded, pad)
    with open(oligo_file) as f:
        oligo_list = [ln.rstrip("\n") for ln in f.readlines()]
    print("oligo_len", len(oligo_list[0]), file=out)
    rng = np.random.default_rng(seed)
    posts, rcs = [], []
    for _ in range(num_reads):
        oligo = oligo_list[int(rng.integers(len(oligo_list)))]
        rc = bool(rng.integers(2))
        if rc:
            oligo = reverse_complement(oligo)
        seq = synth.mutate(synth.bases_from_str(oligo), rng, syn_sub_prob, syn_del_prob, syn_ins_prob)
        posts.append(synth.posteriors_from_bases(seq, rng, margin=margin))
        rcs.append(rc)
    own = decoder is None
    dec = Decoder(conv_m, conv_r, msg_len, list_size=list_size, max_deviation=20, device=device) if own else decoder
    try:
        results = dec.decode(posts, rc=rcs)
    finally:
        if own:
            dec.close()
    decoded_dict = {}
    num_success = 0
    for res in results:
        if isinstance(res, (int, np.integer)):
            continue                                   # (the reference decoder aborts on such a read: no list)
        lst = ["".join("1" if b else "0" for b in row) for row in res[0]]
        index, payload_bytes, _ = decode_list_CRC_index(lst, bytes_per_oligo, num_oligos, pad)
        if index is not None:
            num_success += 1
            if index not in decoded_dict:
                decoded_dict[index] = payload_bytes
    print("num_attempted:", num_reads, file=out)
    print("num success:", num_success, file=out)
    print("num_unique", len(decoded_dict), file=out)
    decoded_list = [[k, decoded_dict[k]] for k in decoded_dict]
    if num_oligos_RS:
        RS_decoded_list = rs_code.MainDecoder(decoded_list, num_oligos_RS, num_oligos, device=device)
    else:
        RS_decoded_list = [decoded_dict.get(i, b"0" * bytes_per_oligo) for i in range(num_oligos_data)]
    assert len(RS_decoded_list) == num_oligos_data
    decoded_data = b"".join(RS_decoded_list)[:data_file_size]
    with open(decoded_data_file, "wb") as f:
        f.write(decoded_data)
    return dict(num_attempted=num_reads, num_success=num_success, num_unique=len(decoded_dict), data=decoded_data)


def hamming(a, b):
    """distance.hamming: number of differing positions of two equal-length sequences"""
    if len(a) != len(b):
        raise ValueError("expected two strings of the same length")
    return sum(x != y for x, y in zip(a, b))


def levenshtein(a, b):
    """distance.levenshtein: unit-cost edit distance"""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def tally_decoded_lists(lists, conv_input_list, bytes_per_oligo, pad, list_size):
    """compute_error_rate_from_decoded_lists.py:18-56 over in-memory lists:
    -> dict(num_reads, num_correct, num_erasure_CRC_index, num_error_CRC_index)"""
    num_oligos = len(conv_input_list)
    out = dict(num_reads=0, num_correct=0, num_erasure_CRC_index=0, num_error_CRC_index=0)
    for lst in lists:
        out["num_reads"] += 1
        index, _, msg = decode_list_CRC_index(lst[:list_size], bytes_per_oligo, num_oligos, pad)
        if index is None:
            out["num_erasure_CRC_index"] += 1
        elif msg == conv_input_list[index]:
            out["num_correct"] += 1
        else:
            out["num_error_CRC_index"] += 1
    return out
