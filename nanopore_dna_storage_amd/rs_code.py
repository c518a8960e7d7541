"""The Reed-Solomon outer code of the pipeline on the GPU.

Mirrors the reference's RSCode_schifra/RSCode_16bit_fileio.py -- same function names, argument meaning and
results -- and the consensus step in front of it (decode_RS_from_decoded_lists.py:30-55):

  MainEncoder(listofreads, redundancy)                     RSCode_16bit_fileio.py:266-277
  MainDecoder(listofcorruptedreads, redundancy, total)     RSCode_16bit_fileio.py:289-299
  consensus(decoded)                                       decode_RS_from_decoded_lists.py:37-51
  decode_from_lists(lists, ...)                            decode_RS_from_decoded_lists.py:30-55 for one trial

The payload of oligo i is a byte string of 2*s bytes = s 16-bit symbols; symbol column c of all oligos is one
RS(65535, 65535 - redundancy) codeword over GF(2^16), shortened by padding with ASCII '0' bytes.  The reference
compiles and runs its C++ codec once per column; here all columns go to the device in one call (lva_rs_decode /
lva_rs_encode, csrc/rs_kernels.hip).  No CPU fallback: without the HIP library or a GPU the calls raise.

Differences from the reference, by necessity: redundancy is limited to 4096 symbols (LDS budget of one workgroup;
the reference's experiments use 30 % of a few hundred oligos); a decode without any erased read works (the
reference raises FileNotFoundError there, RSCode_16bit_fileio.py:130-131).
"""
import numpy as np

from . import helper
from ._lib import LvaError, load_library

PAD = 0x3030          # b'0' * 2: rjust(..., b'0') padding (:58, :104), dummy reads (:242) and the fill of a failed column (:123)


def _check(st):
    if st != 0:
        raise LvaError(st, load_library().lva_rs_last_error().decode())


def _columns(reads):
    """list of n byte strings of 2*s bytes -> uint16 [s, n]: column c = listofreadstolistofRSinputdata(...)[c] (:168-183)"""
    a = np.frombuffer(b"".join(reads), dtype="<u2").reshape(len(reads), -1)
    return np.ascontiguousarray(a.T)


def _reads(cols):
    """uint16 [s, n] -> list of n byte strings (listofRSoutputdatatolistofreads, :196-211)"""
    a = np.ascontiguousarray(cols.T).astype("<u2")
    return [a[i].tobytes() for i in range(a.shape[0])]


def MainEncoder(listofreads, redundancy, device=0):
    """list of numreads payloads (equal length, 2 bytes per symbol) -> numreads + redundancy payloads:
    the data reads followed by the parity reads."""
    cols = _columns(listofreads)
    s, n = cols.shape
    out = np.zeros((s, n + redundancy), np.uint16)
    _check(load_library().lva_rs_encode(device, cols.ctypes.data, s, n, redundancy, PAD, out.ctypes.data))
    return _reads(out)


def MainDecoder(listofcorruptedreads, redundancy, totalnumreads, device=0, return_ok=False):
    """listofcorruptedreads: [[index, payload]] for the reads that arrived (index < totalnumreads).
    -> totalnumreads - redundancy decoded payloads; a column the decoder gives up on is ASCII '0' in every read."""
    if not listofcorruptedreads:
        raise IndexError("list index out of range")                   # listofcorruptedreads[0][1] (:237)
    spr = len(listofcorruptedreads[0][1]) // 2
    reads = [b"".rjust(2 * spr, b"0")] * totalnumreads                 # dummy reads (:242-243)
    present = np.zeros(totalnumreads, bool)
    for idx, payload in listofcorruptedreads:
        if not 0 <= idx < totalnumreads or present[idx]:
            raise ValueError("list.remove(x): x not in list")          # erasure_loc_list.remove (:247)
        reads[idx] = payload
        present[idx] = True
    cols = _columns(reads)
    erasures = np.ascontiguousarray(np.nonzero(~present)[0], dtype=np.int32)
    n_data = totalnumreads - redundancy
    out = np.zeros((spr, n_data), np.uint16)
    ok = np.zeros(spr, np.int32)
    _check(load_library().lva_rs_decode(device, cols.ctypes.data, spr, totalnumreads, redundancy,
                                        erasures.ctypes.data if len(erasures) else None, len(erasures), PAD, PAD,
                                        out.ctypes.data, ok.ctypes.data))
    dec = _reads(out)
    return (dec, ok.astype(bool)) if return_ok else dec


def consensus(decoded):
    """decode_RS_from_decoded_lists.py:37-51.  decoded: iterable of (index, payload_bytes) in read order.  Per index the
    payload seen most often wins; among equal counts the one that reached the count first (the reference re-sorts the
    candidates with a stable sort by -count after every read).  -> [[index, payload]] in first-seen order of the indices."""
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


def decode_from_lists(lists, bytes_per_oligo, num_oligos_RS, num_oligos, pad=False, list_size=None, device=0):
    """One trial of decode_RS_from_decoded_lists.py:30-55 on in-memory decoded lists (one list of '0'/'1' strings per
    read, best first): CRC-8/index filter per read (helper.decode_list_CRC_index), per-index consensus, RS decode.
    -> (data bytes = the decoded payloads joined, number of reads that passed the filter)"""
    decoded = []
    for lst in lists:
        index, payload, _ = helper.decode_list_CRC_index(lst if list_size is None else lst[:list_size], bytes_per_oligo, num_oligos, pad)
        if index is not None:
            decoded.append((index, payload))
    if not decoded:                 # nothing to decode from (the reference's MainDecoder raises IndexError on an empty list)
        return b"", 0
    rs_out = MainDecoder(consensus(decoded), num_oligos_RS, num_oligos, device=device)
    return b"".join(rs_out), len(decoded)
