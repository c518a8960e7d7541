"""CPU oracle of SURVEY.md section 8(f) row N4: the Reed-Solomon outer code.

TEST INFRASTRUCTURE ONLY (imported by tests/ and __graft_entry__.smoke(), never by the product package).

Two things live here, each citing the reference lines it follows (RS = /root/reference/RSCode_schifra):
  * `decode_block` / `encode_block`: a numpy restatement of what `schifra_RS_16bit_fileio 0|1 ...` computes for one
    codeword -- schifra::reed_solomon::decoder<65535, fec>::decode(block, erasure_list)
    (RS/schifra_reed_solomon_decoder.hpp:64-168 and the helpers :189-424) and encoder::encode
    (RS/schifra_reed_solomon_encoder.hpp:57-88) over GF(2^16) with primitive polynomial
    x^16 + x^12 + x^3 + x + 1 (RS/schifra_galois_field.hpp:511-512, field generated as :317-357), generator roots
    alpha^0 .. alpha^(fec-1) (RS/schifra_sequential_root_generator_polynomial_creator.hpp:33-55 with index 0).
    Pinned against the compiled reference (`ref_codec`, below) by tests/test_rs_oracle.py.
  * `MainDecoder` / `MainEncoder` and their helpers: the Python glue of RS/RSCode_16bit_fileio.py:235-299
    (column-wise codewords over the 16-bit symbols of the oligo payloads, ASCII '0' padding and dummies, erasure list).

`ref_codec(fec)` = the UNMODIFIED reference program RS/schifra_RS_16bit_fileio.cpp, compiled by oracle/Makefile
(`make rsref FEC="..."`) into oracle/_ref/schifra_RS_16bit_fileio_<fec>.out with the parameter header that
RS/RSCode_16bit_fileio.py:33-43 writes before every compile (build container only; the binaries travel).
"""
import os
import struct
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
N = 65535                       # code_length (RSCode_16bit_fileio.py:49, :96)
PAD = 0x3030                    # rjust(..., b'0'): two ASCII '0' bytes per padding / dummy symbol (:58, :104, :242)

# ---------------------------------------------------------------- GF(2^16), schifra_galois_field.hpp:317-357
_EXP = np.zeros(2 * N + 2, dtype=np.int64)
_LOG = np.zeros(N + 1, dtype=np.int64)
_x = 1
for _i in range(N):
    _EXP[_i] = _x
    _LOG[_x] = _i
    _x <<= 1
    if _x & 0x10000:
        _x ^= 0x1100B
_EXP[N:2 * N] = _EXP[:N]
_EXP[2 * N:] = _EXP[:2]
_LOG[0] = -1


def gmul(a, b):
    return 0 if a == 0 or b == 0 else int(_EXP[_LOG[a] + _LOG[b]])


def gdiv(a, b):
    return 0 if a == 0 or b == 0 else int(_EXP[_LOG[a] - _LOG[b] + N])          # field::div (:112-121)


def alpha(i):
    return int(_EXP[i % N])


def _vmul_scalar(v, s):
    """coefficient-wise product of an int64 vector with a field element"""
    out = np.zeros_like(v)
    if s:
        nz = v != 0
        out[nz] = _EXP[_LOG[v[nz]] + _LOG[s]]
    return out


def _strip(v):
    """field_polynomial::simplify (schifra_galois_field_polynomial.hpp:610-631): drop leading zero coefficients"""
    n = len(v)
    while n > 0 and v[n - 1] == 0:
        n -= 1
    return v[:n]


def _poly_eval_at_powers(poly, idx):
    """poly(alpha^i) for every i in idx (poly[k] = coefficient of x^k)"""
    idx = np.asarray(idx, dtype=np.int64)
    acc = np.zeros(len(idx), dtype=np.int64)
    for k, c in enumerate(poly):
        if c:
            acc ^= _EXP[(_LOG[c] + idx * k) % N]
    return acc


def syndromes(block, fec):
    """compute_syndrome (decoder.hpp:227-240): S_i = R(alpha^i), R(x) = sum_b block[b] x^(N-1-b)"""
    block = np.asarray(block, dtype=np.int64)
    nz = np.nonzero(block)[0]
    lg = _LOG[block[nz]]
    e = (N - 1 - nz).astype(np.int64)
    s = np.zeros(fec, dtype=np.int64)
    for i in range(fec):
        s[i] = np.bitwise_xor.reduce(_EXP[(lg + i * e) % N]) if len(nz) else 0
    return s


def decode_block(block, fec, erasures):
    """schifra decoder::decode(rsblock, erasure_list) (decoder.hpp:64-168) on a full-length block of N symbols.
    -> (ok, corrected block).  ok False = the reference program prints "Critical decoding failure" and exits 1."""
    block = np.array(block, dtype=np.int64)
    assert block.shape == (N,)
    S = len(erasures)
    if S > fec:                                                       # :66-75
        return False, block
    syn = syndromes(block, fec)
    if not syn.any():                                                 # :82-90
        return True, block
    lam = np.array([1], dtype=np.int64)                               # :92
    for pos in erasures:                                              # prepare_erasure_list + compute_gamma (:211-225, :242-248)
        a = alpha(N - 1 - int(pos))                                   # gamma_table_[loc] = 1 + X * alpha^loc (:205-208)
        nxt = np.zeros(len(lam) + 1, dtype=np.int64)
        nxt[:len(lam)] ^= lam
        nxt[1:] ^= _vmul_scalar(lam, a)
        lam = _strip(nxt)
    if S < fec:                                                       # modified_berlekamp_massey_algorithm (:296-337)
        i_ = -1
        l = S
        prev = np.concatenate([[0], lam])                             # lambda << 1
        for rnd in range(S, fec):
            ub = min(l, len(lam) - 1)                                 # compute_discrepancy (:275-294)
            d = 0
            for k in range(ub + 1):
                d ^= gmul(int(lam[k]), int(syn[rnd - k]))
            if d != 0:
                n = max(len(lam), len(prev))
                tau = np.zeros(n, dtype=np.int64)
                tau[:len(lam)] ^= lam
                tau[:len(prev)] ^= _vmul_scalar(prev, d)
                tau = _strip(tau)
                if l < rnd - i_:
                    tmp = rnd - i_
                    i_ = rnd - l
                    l = tmp
                    dinv = gdiv(1, d)
                    prev = _vmul_scalar(lam, dinv)                    # lambda / discrepancy, coefficient-wise (:327)
                lam = tau
            if len(prev) > 0:
                prev = np.concatenate([[0], prev])                    # previous_lambda <<= 1
    deg = len(lam) - 1
    # find_roots (:250-273): Chien search over alpha^1 .. alpha^N, stops once deg roots are found
    vals = _poly_eval_at_powers(lam, np.arange(1, N + 1))
    roots = (np.nonzero(vals == 0)[0] + 1)[:max(deg, 0)]
    L = len(roots)
    if L == 0:                                                        # :105-121
        return False, block
    if ((2 * L - S) % (1 << 64)) > fec:                               # size_t arithmetic (:122-146)
        return False, block
    # forney_algorithm (:339-385)
    conv = np.zeros(len(lam) + fec - 1, dtype=np.int64)
    for k, c in enumerate(lam):
        if c:
            conv[k:k + fec] ^= _vmul_scalar(syn, int(c))
    omega = _strip(_strip(conv)[:fec]) if len(_strip(conv)) >= fec else _strip(conv)   # operator*= simplifies, then % power (:418-426)
    der = np.zeros(len(lam), dtype=np.int64)                          # derivative (:580-598)
    if len(lam) > 1:
        der[0:len(lam) - 1:2] = lam[1:len(lam):2]
        der = _strip(der)
    else:
        der = np.zeros(1, dtype=np.int64)
    om = _poly_eval_at_powers(omega, roots)
    dn = _poly_eval_at_powers(der, roots)
    for r, o, dnm in zip(roots, om, dn):
        num = gmul(int(o), alpha(N - int(r)))                         # root_exponent_table_[i] = alpha^(N-i) (:193-196)
        if num != 0:
            if dnm != 0:
                block[int(r) - 1] ^= gdiv(num, int(dnm))
            else:
                return False, block                                   # e_decoder_error3
    return (deg == L), block                                          # :376-383


def encode_block(data, fec):
    """encoder::encode (encoder.hpp:57-88): systematic, parity = (data(x) x^fec) mod g(x).  data: N-fec symbols."""
    data = np.asarray(data, dtype=np.int64)
    assert data.shape == (N - fec,)
    g = np.array([1], dtype=np.int64)                                 # prod (x + alpha^i), i < fec
    for i in range(fec):
        nxt = np.zeros(len(g) + 1, dtype=np.int64)
        nxt[1:] ^= g
        nxt[:len(g)] ^= _vmul_scalar(g, alpha(i))
        g = nxt
    # remainder by LFSR over the data symbols, highest power first (block[0] is the coefficient of x^(N-1))
    rem = np.zeros(fec, dtype=np.int64)
    glow = g[:fec]
    for sym in data:
        fb = int(sym) ^ int(rem[fec - 1])
        rem[1:] = rem[:-1]
        rem[0] = 0
        if fb:
            rem ^= _vmul_scalar(glow, fb)
    return np.concatenate([data, rem[::-1]])                          # fec(i) = parities[fec-1-i] (:72-76)


# ---------------------------------------------------------------- the compiled reference program
def ref_path(fec):
    return os.path.join(_HERE, "_ref", "schifra_RS_16bit_fileio_%d.out" % fec)


def have_ref(fec):
    return os.access(ref_path(fec), os.X_OK)


def ref_codec(fec, block, erasures=None, encode=False):
    """run the reference program on one block of symbols (file formats of schifra_RS_16bit_fileio.cpp:89-103, :146-160,
    :194-204) -> (exit code, output symbols or None when it wrote no file)"""
    block = np.asarray(block, dtype="<u2")
    with tempfile.TemporaryDirectory() as d:
        fin, fout, fer = (os.path.join(d, x) for x in ("in.dat", "out.dat", "er.dat"))
        block.tofile(fin)
        use_er = bool(erasures is not None and len(erasures))
        if use_er:
            np.asarray(erasures, dtype="<u2").tofile(fer)
        r = subprocess.run([ref_path(fec), "1" if encode else "0", fin, fout, "1" if (use_er or encode) else "0", fer],
                           stdout=subprocess.DEVNULL)
        out = np.fromfile(fout, dtype="<u2").astype(np.int64) if os.path.exists(fout) else None
        return r.returncode, out


# ---------------------------------------------------------------- RSCode_16bit_fileio.py glue, restated
def _symbols(bytestring):
    return np.frombuffer(bytestring, dtype="<u2").astype(np.int64)


def _bytes(symbols):
    return np.asarray(symbols, dtype="<u2").tobytes()


def RS_encode_16bit(inputbytestring, codeword_data_len, codeword_redundancy, codec=None):
    """RSCode_16bit_fileio.py:48-77"""
    data_len = N - codeword_redundancy
    padded = inputbytestring.rjust(2 * data_len, b"0")
    enc = (codec or (lambda b: encode_block(b, codeword_redundancy)))(_symbols(padded))
    return _bytes(enc)[2 * (data_len - codeword_data_len):]


def RS_decode_16bit(inputbytestring, codeword_data_len, codeword_redundancy, proto_erasure_loc_list, codec=None):
    """RSCode_16bit_fileio.py:87-137.  codec(block, erasures) -> (ok, corrected block): decode_block by default."""
    data_len = N - codeword_redundancy
    padding_len = data_len - codeword_data_len
    padded = inputbytestring.rjust(2 * N, b"0")
    er = [x + padding_len for x in proto_erasure_loc_list]
    ok, blk = (codec or (lambda b, e: decode_block(b, codeword_redundancy, e)))(_symbols(padded), er)
    if ok:
        return _bytes(blk[:data_len])[2 * padding_len:]
    return b"".rjust(2 * codeword_data_len, b"0")                     # no output file (:122-123)


def listofreadstolistofRSinputdata(listofreads):
    """:168-183 -- column s of the payloads = one RS input string"""
    spr = len(listofreads[0]) // 2
    return [b"".join(r[2 * s:2 * s + 2] for r in listofreads) for s in range(spr)]


def listofRSoutputdatatolistofreads(listofRSoutputdata):
    """:196-211"""
    nreads = len(listofRSoutputdata[0]) // 2
    return [b"".join(col[2 * i:2 * i + 2] for col in listofRSoutputdata) for i in range(nreads)]


def MainEncoder(listofreads, redundancy, codec=None):
    """:266-277"""
    cols = listofreadstolistofRSinputdata(listofreads)
    return listofRSoutputdatatolistofreads([RS_encode_16bit(c, len(listofreads), redundancy, codec) for c in cols])


def MainDecoder(listofcorruptedreads, redundancy, totalnumreads, codec=None):
    """:289-299 (+ :242-255: missing reads become dummy reads of ASCII '0' and erasure locations)"""
    spr = len(listofcorruptedreads[0][1]) // 2
    reads = [b"".rjust(2 * spr, b"0") for _ in range(totalnumreads)]
    erasure = list(range(totalnumreads))
    for idx, payload in listofcorruptedreads:
        reads[idx] = payload
        erasure.remove(idx)
    cols = listofreadstolistofRSinputdata(reads)
    out = [RS_decode_16bit(c, totalnumreads - redundancy, redundancy, erasure, codec) for c in cols]
    return listofRSoutputdatatolistofreads(out)


def consensus(decoded, list_size=None):
    """decode_RS_from_decoded_lists.py:30-51: `decoded` = iterable of (index, payload_bytes) in read order (reads whose
    list has no CRC/index match are left out).  Per index the payload seen most often wins; among equal counts the one
    that reached the count first (stable sort by -count after every update).  -> [[index, payload]] in first-seen order"""
    d = {}
    for index, payload in decoded:
        if index in d:
            for tup in d[index]:
                if tup[0] == payload:
                    tup[1] += 1
                    break
            else:
                d[index].append([payload, 1])
            d[index] = sorted(d[index], key=lambda x: -x[1])
        else:
            d[index] = [[payload, 1]]
    return [[k, d[k][0][0]] for k in d]
