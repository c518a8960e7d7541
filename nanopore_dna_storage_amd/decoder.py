"""Python face of the C ABI: code description, encoder, batched GPU list decoding.

Mirrors the reference's operator surface for this path:
  * `encode`  <->  `viterbi_nanopore.out -m encode`  (viterbi_convolutional_code.cpp:215-225)
  * `Decoder.decode`  <->  `viterbi_nanopore.out -m decode` per read (:226-254), batched.
"""
import ctypes
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import LvaError, load_library

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def _sm(sync_marker):
    return sync_marker.encode() if sync_marker else None


@dataclass
class CodeInfo:
    mem_conv: int
    rate: int
    msg_len: int
    nstate_pos: int
    nstate_conv: int
    oligo_len: int
    msg_words: int
    initial_state: int
    final_state: int
    g: tuple
    pattern: tuple


def code_info(mem_conv, rate, msg_len, rc=False, sync_marker="", sync_period=0):
    """set_conv_params (:264-415): raises LvaError for parameters the reference rejects."""
    L = load_library()
    s = _lib.CodeInfoStruct()
    st = L.lva_code_describe(mem_conv, rate, msg_len, int(bool(rc)), _sm(sync_marker), sync_period, ctypes.byref(s))
    if st != 0:
        raise LvaError(st)
    return CodeInfo(mem_conv, rate, msg_len, s.nstate_pos, s.nstate_conv, s.oligo_len, s.msg_words,
                    s.initial_state, s.final_state, (s.g0, s.g1), tuple(s.pattern[:s.pattern_len]))


def code_tables(mem_conv, rate, msg_len, rc=False, sync_marker="", sync_period=0):
    """The per-position / per-conv-state tables the kernels index (for inspection and tests)."""
    info = code_info(mem_conv, rate, msg_len, rc, sync_marker, sync_period)
    L = load_library()
    pos2msg = np.zeros(info.nstate_pos, np.uint32)
    ptype = np.zeros(info.nstate_pos, np.uint8)
    vmask = np.zeros(info.nstate_pos, np.uint32)
    vval = np.zeros(info.nstate_pos, np.uint32)
    predtab = np.zeros((4, info.nstate_conv), np.uint16)
    st = L.lva_code_tables(mem_conv, rate, msg_len, int(bool(rc)), _sm(sync_marker), sync_period,
                           pos2msg.ctypes.data, ptype.ctypes.data, vmask.ctypes.data, vval.ctypes.data,
                           predtab.ctypes.data)
    if st != 0:
        raise LvaError(st)
    return dict(pos2msg=pos2msg, ptype=ptype, vmask=vmask, vval=vval, predtab=predtab)


def band_table(mem_conv, rate, msg_len, nblk, max_deviation=None, rc=False, sync_marker="", sync_period=0):
    """-> (reference, working): int arrays [nblk, 2] of [lo, hi) per time step -- the reference's band (:677-679) and the band the
    kernels work on (without positions whose lists cannot reach the output; include/lva_decoder.h lva_band_table)."""
    L = load_library()
    ref = np.zeros((nblk, 2), np.uint32)
    work = np.zeros((nblk, 2), np.uint32)
    md = 0xFFFFFFFF if max_deviation is None else int(max_deviation)
    st = L.lva_band_table(mem_conv, rate, msg_len, int(bool(rc)), _sm(sync_marker), sync_period, int(nblk), md,
                          ref.ctypes.data, work.ctypes.data)
    if st != 0:
        raise LvaError(st)
    return ref.astype(np.int64), work.astype(np.int64)


def encode(mem_conv, rate, msg_len, msgs):
    """msgs: array [n, msg_len] (or [msg_len]) of 0/1 -> uint8 array [n, oligo_len] of 0..3 (A,C,G,T)."""
    msgs = np.ascontiguousarray(msgs, dtype=np.uint8)
    single = msgs.ndim == 1
    if single:
        msgs = msgs[None, :]
    if msgs.shape[1] != msg_len:
        raise LvaError(-10, "Message length does not match msg_len parameter.")   # :219-222
    info = code_info(mem_conv, rate, msg_len)
    out = np.zeros((msgs.shape[0], info.oligo_len), np.uint8)
    st = load_library().lva_encode(mem_conv, rate, msg_len, msgs.ctypes.data, msgs.shape[0], out.ctypes.data)
    if st != 0:
        raise LvaError(st)
    return out[0] if single else out


def bases_to_str(bases):
    return _BASES[np.asarray(bases, dtype=np.uint8)].tobytes().decode()


def str_to_bits(s):
    return np.frombuffer(s.encode(), dtype=np.uint8) - ord("0")


def algorithmic_bytes(mem_conv, rate, msg_len, nblk, list_size, max_deviation=None, rc=False,
                      sync_marker="", sync_period=0):
    """sum_t [2 R(t) L (4+4W) + 160] for one read of nblk blocks."""
    out = ctypes.c_double(0)
    md = _lib.MAX_DEVIATION_DEFAULT if max_deviation is None else max_deviation
    st = load_library().lva_algorithmic_bytes(mem_conv, rate, msg_len, int(bool(rc)), _sm(sync_marker), sync_period,
                                              nblk, list_size, md, ctypes.byref(out))
    if st != 0:
        raise LvaError(st)
    return out.value


class Decoder:
    """A list-Viterbi decoder bound to one GPU.  Fails loudly without a GPU (no CPU path).

    kernel: 0 = default (the fastest mode for the configuration: 4 for list sizes 2 / 4 / 8 with up to 192 message bits,
    else 2 for list sizes up to 64, the thread-per-target exact kernel beyond), 1 = thread-per-target exact kernel,
    2 = fast kernel + exact fix-up, 3 = wavefront-per-target exact kernel, 4 = fast kernel with lazy messages
    (materialised every second time step).  All modes give the reference's lists bit for bit; they differ in speed only."""

    def __init__(self, mem_conv, rate, msg_len, list_size=1, max_deviation=None, sync_marker="", sync_period=0,
                 device=0, max_slots=0, kernel=0, mem_budget_bytes=0):
        self._L = load_library()
        self._sync = _sm(sync_marker)
        cfg = _lib.Config(mem_conv, rate, msg_len, list_size,
                          _lib.MAX_DEVIATION_DEFAULT if max_deviation is None else max_deviation,
                          self._sync, sync_period, device, max_slots, kernel, mem_budget_bytes)
        h = ctypes.c_void_p()
        st = self._L.lva_decoder_create(ctypes.byref(cfg), ctypes.byref(h))
        if st != 0:
            raise LvaError(st, self._L.lva_last_hip_error().decode())
        self._h = h
        self.mem_conv, self.rate, self.msg_len, self.list_size = mem_conv, rate, msg_len, list_size

    def close(self):
        if getattr(self, "_h", None):
            self._L.lva_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @staticmethod
    def _pack(posts):
        posts = [np.ascontiguousarray(p, dtype=np.float32).reshape(-1, 40) for p in posts]
        off = np.zeros(len(posts) + 1, np.int64)
        off[1:] = np.cumsum([p.shape[0] for p in posts])
        flat = np.concatenate(posts, axis=0) if posts else np.zeros((0, 40), np.float32)
        return np.ascontiguousarray(flat), off

    def _outputs(self, n):
        return (np.zeros((n, self.list_size, self.msg_len), np.uint8), np.zeros((n, self.list_size), np.float32),
                np.zeros(n, np.int32))

    def _unpack(self, n, msgs, scores, counts):
        out = []
        for i in range(n):
            c = int(counts[i])
            out.append((msgs[i, :c].copy(), scores[i, :c].copy()) if c >= 0 else c)
        return out

    def decode(self, posts, rc=None):
        """posts: list of float32 [nblk_i, 40] matrices (.post layout).  rc: optional bool per read.
        -> list of (msgs uint8[count, msg_len], scores float32[count]) or a negative error code per read."""
        flat, off = self._pack(posts)
        return self.decode_packed(flat, off, rc)

    pack = _pack

    def decode_packed(self, flat, off, rc=None):
        """decode() on a host buffer that is already in the C ABI's form: `flat` float32 [sum nblk, 40] (all reads
        back to back), `off` int64 [n+1] block offsets.  The host->device copy happens inside the call."""
        n = len(off) - 1
        assert flat.dtype == np.float32 and flat.flags.c_contiguous and off.dtype == np.int64
        rcf = None if rc is None else np.ascontiguousarray(rc, dtype=np.uint8)
        msgs, scores, counts = self._outputs(n)
        st = self._L.lva_decode_batch(self._h, flat.ctypes.data, off.ctypes.data, n,
                                      None if rcf is None else rcf.ctypes.data,
                                      msgs.ctypes.data, scores.ctypes.data, counts.ctypes.data)
        if st != 0:
            raise LvaError(st, self._L.lva_last_hip_error().decode())
        return self._unpack(n, msgs, scores, counts)

    # --- inputs resident in HBM (bench.py) ---------------------------------------------------
    def upload(self, posts):
        flat, off = self._pack(posts)
        p = ctypes.c_void_p()
        st = self._L.lva_device_alloc(self._h, flat.nbytes, ctypes.byref(p))
        if st != 0:
            raise LvaError(st, self._L.lva_last_hip_error().decode())
        st = self._L.lva_device_upload(self._h, p, flat.ctypes.data, flat.nbytes)
        if st != 0:
            raise LvaError(st, self._L.lva_last_hip_error().decode())
        return p, off

    def free(self, dev_ptr):
        self._L.lva_device_free(self._h, dev_ptr)

    def decode_resident(self, dev_ptr, off, rc=None):
        n = len(off) - 1
        rcf = None if rc is None else np.ascontiguousarray(rc, dtype=np.uint8)
        msgs, scores, counts = self._outputs(n)
        st = self._L.lva_decode_batch_device(self._h, dev_ptr, off.ctypes.data, n,
                                             None if rcf is None else rcf.ctypes.data,
                                             msgs.ctypes.data, scores.ctypes.data, counts.ctypes.data)
        if st != 0:
            raise LvaError(st, self._L.lva_last_hip_error().decode())
        return self._unpack(n, msgs, scores, counts)

    def _check(self, st):
        if st != 0:
            raise LvaError(st, self._L.lva_last_hip_error().decode())

    def decode_windows_resident(self, dev_ptr, first_block, n_blocks, rc=None):
        """decode windows [first_block[i], first_block[i] + n_blocks[i]) of a resident posterior buffer in place
        (helper.truncate_post_file + the decode call of generate_decoded_lists.py:80-89, without the copy)"""
        fb = np.ascontiguousarray(first_block, dtype=np.int64)
        nb = np.ascontiguousarray(n_blocks, dtype=np.int64)
        n = len(fb)
        rcf = None if rc is None else np.ascontiguousarray(rc, dtype=np.uint8)
        msgs, scores, counts = self._outputs(n)
        self._check(self._L.lva_decode_windows_device(self._h, dev_ptr, fb.ctypes.data, nb.ctypes.data, n,
                                                      None if rcf is None else rcf.ctypes.data,
                                                      msgs.ctypes.data, scores.ctypes.data, counts.ctypes.data))
        return self._unpack(n, msgs, scores, counts)

    # --- basecall of the posterior matrix + barcode localisation ------------
    def _basecall_out(self, off, bases, trans, nb):
        return [(bases[off[i]:off[i] + nb[i]].tobytes().decode("ascii"), trans[off[i]:off[i] + nb[i]].astype(np.int64))
                for i in range(len(nb))]

    def basecall(self, posts):
        """flappie's basecall of each posterior matrix (flappie.c:273-285): [(base string, trans positions)]"""
        n = len(posts)
        flat, off = self._pack(posts)
        T = max(int(off[-1]), 1)
        bases, trans, nb = np.zeros(T, np.uint8), np.zeros(T, np.uint32), np.zeros(max(n, 1), np.int32)
        self._check(self._L.lva_basecall_batch(self._h, flat.ctypes.data, off.ctypes.data, n, bases.ctypes.data,
                                               trans.ctypes.data, nb.ctypes.data))
        return self._basecall_out(off, bases, trans, nb[:n])

    def basecall_resident(self, dev_ptr, off):
        n = len(off) - 1
        T = max(int(off[-1]), 1)
        bases, trans, nb = np.zeros(T, np.uint8), np.zeros(T, np.uint32), np.zeros(max(n, 1), np.int32)
        self._check(self._L.lva_basecall_batch_device(self._h, dev_ptr, off.ctypes.data, n, bases.ctypes.data,
                                                      trans.ctypes.data, nb.ctypes.data))
        return self._basecall_out(off, bases, trans, nb[:n])

    @staticmethod
    def _payload_out(res, n):
        inf = float("inf")
        big = 0x7FFFFFFF
        return [dict(ok=bool(r.ok), start_pos=r.start_pos, end_pos=r.end_pos, rc=bool(r.rc),
                     dist_start=inf if r.dist_start == big else r.dist_start,
                     dist_end=inf if r.dist_end == big else r.dist_end) for r in res[:n]]

    def find_barcode(self, basecalls, trans_lists, start_barcode, end_barcode):
        """helper.find_barcode_pos_in_post (helper.py:157-210) for a batch of (basecall, trans list) pairs"""
        n = len(basecalls)
        off = np.zeros(n + 1, np.int64)
        off[1:] = np.cumsum([len(b) for b in basecalls])
        for b, t in zip(basecalls, trans_lists):
            if len(t) < len(b):
                raise ValueError("trans list shorter than the basecall")
        bases = np.frombuffer("".join(basecalls).encode("ascii") or b"\0", dtype=np.uint8).copy()
        trans = np.concatenate([np.asarray(t, dtype=np.uint32)[:len(b)] for b, t in zip(basecalls, trans_lists)]
                               + [np.zeros(1, np.uint32)])
        res = (_lib.PayloadPos * max(n, 1))()
        self._check(self._L.lva_find_barcode_batch(self._h, bases.ctypes.data, trans.ctypes.data, off.ctypes.data, n,
                                                   start_barcode.encode(), end_barcode.encode(), res))
        return self._payload_out(res, n)

    def locate_payload(self, posts, start_barcode, end_barcode):
        """generate_decoded_lists.py:68-79 per read: basecall, barcode search in both orientations, choice.
        -> [dict(ok, start_pos, end_pos, rc, dist_start, dist_end)]"""
        n = len(posts)
        flat, off = self._pack(posts)
        res = (_lib.PayloadPos * max(n, 1))()
        self._check(self._L.lva_locate_payload_batch(self._h, flat.ctypes.data, off.ctypes.data, n, start_barcode.encode(),
                                                     end_barcode.encode(), self.mem_conv + self.msg_len + 1, res))
        return self._payload_out(res, n)

    def locate_payload_resident(self, dev_ptr, off, start_barcode, end_barcode):
        n = len(off) - 1
        res = (_lib.PayloadPos * max(n, 1))()
        self._check(self._L.lva_locate_payload_batch_device(self._h, dev_ptr, off.ctypes.data, n, start_barcode.encode(),
                                                            end_barcode.encode(), self.mem_conv + self.msg_len + 1, res))
        return self._payload_out(res, n)

    def decode_with_barcodes(self, posts, start_barcode, end_barcode):
        """The real-data chain of generate_decoded_lists.py:68-89 on the device: posteriors are uploaded once,
        payload windows located, then decoded in place.  -> [(locate dict, decode result or None)]"""
        n = len(posts)
        dev, off = self.upload(posts)
        try:
            loc = self.locate_payload_resident(dev, off, start_barcode, end_barcode)
            good = [i for i in range(n) if loc[i]["ok"]]
            dec = self.decode_windows_resident(dev, [off[i] + loc[i]["start_pos"] for i in good],
                                               [loc[i]["end_pos"] - loc[i]["start_pos"] + 1 for i in good],
                                               rc=[loc[i]["rc"] for i in good]) if good else []
        finally:
            self.free(dev)
        out = [(loc[i], None) for i in range(n)]
        for i, r in zip(good, dec):
            out[i] = (loc[i], r)
        return out

    def set_launch_events(self, on=True):
        """HIP events around every trellis-step launch of later decode calls (profile(): dominant_kernel_ms, step_pair_ms)"""
        self._check(self._L.lva_decoder_set_launch_events(self._h, int(bool(on))))

    def profile(self):
        p = _lib.Profile()
        self._L.lva_decoder_profile(self._h, ctypes.byref(p))
        d = {k: getattr(p, k) for k, _ in _lib.Profile._fields_}
        d["fixup_reason"] = list(p.fixup_reason)
        return d
