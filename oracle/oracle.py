"""ctypes binding of the CPU oracle (oracle/liblva_oracle.so) and a runner for the
unmodified reference binary (oracle/_ref/viterbi_nanopore.out).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.
"""
import ctypes
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblva_oracle.so")
REF_BIN = os.path.join(_HERE, "_ref", "viterbi_nanopore.out")

_lib = None


def build(quiet=True):
    """(re)build the oracle .so and, when /root/reference exists, oracle/_ref."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.lva_oracle_code_new.restype = ctypes.c_void_p
        L.lva_oracle_code_new.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int,
                                          ctypes.c_char_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_int)]
        L.lva_oracle_code_free.argtypes = [ctypes.c_void_p]
        for name in ("nstate_pos", "nstate_conv", "initial_state", "final_state"):
            f = getattr(L, "lva_oracle_" + name)
            f.restype = ctypes.c_uint32
            f.argtypes = [ctypes.c_void_p]
        L.lva_oracle_pos2msg.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.lva_oracle_pattern_at.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.lva_oracle_is_valid_state.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
        L.lva_oracle_prev_states.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int,
                                             ctypes.c_void_p, ctypes.c_int]
        L.lva_oracle_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.lva_oracle_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                        ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        L.lva_oracle_band.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                      ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
        L.bc_oracle_basecall.restype = ctypes.c_int
        L.bc_oracle_basecall.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, status):
        super().__init__("oracle status %d" % status)
        self.status = status


class OracleCode:
    """set_conv_params of the reference, as the oracle restates it."""

    def __init__(self, mem_conv, rate, msg_len, rc=False, sync_marker="", sync_period=0):
        st = ctypes.c_int(0)
        self._h = lib().lva_oracle_code_new(mem_conv, rate, msg_len, int(bool(rc)),
                                            sync_marker.encode() if sync_marker else None,
                                            sync_period, ctypes.byref(st))
        if not self._h:
            raise OracleError(st.value)
        self.mem_conv, self.rate, self.msg_len, self.rc = mem_conv, rate, msg_len, bool(rc)
        self.nstate_pos = lib().lva_oracle_nstate_pos(self._h)
        self.nstate_conv = lib().lva_oracle_nstate_conv(self._h)
        self.initial_state = lib().lva_oracle_initial_state(self._h)
        self.final_state = lib().lva_oracle_final_state(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().lva_oracle_code_free(self._h)
            self._h = None

    def pos2msg(self):
        out = np.zeros(self.nstate_pos, dtype=np.uint32)
        lib().lva_oracle_pos2msg(self._h, out.ctypes.data)
        return out

    def pattern_at(self, pos):
        return lib().lva_oracle_pattern_at(self._h, pos)

    def is_valid_state(self, pos, conv):
        return bool(lib().lva_oracle_is_valid_state(self._h, pos, conv))

    def prev_states(self, conv, crf, pattern):
        out = np.zeros((32, 6), dtype=np.int32)
        n = lib().lva_oracle_prev_states(self._h, conv, crf, pattern, out.ctypes.data, 32)
        return out[:n].copy()

    def band(self, t, nblk, max_deviation, band_fma=True):
        a, b = ctypes.c_uint32(0), ctypes.c_uint32(0)
        lib().lva_oracle_band(self._h, t, nblk, max_deviation, int(band_fma), ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def encode(self, msg_bits):
        msg = np.ascontiguousarray(msg_bits, dtype=np.uint8)
        assert msg.shape == (self.msg_len,)
        out = np.zeros(self.nstate_pos - 1, dtype=np.uint8)
        st = lib().lva_oracle_encode(self._h, msg.ctypes.data, out.ctypes.data)
        if st != 0:
            raise OracleError(st)
        return out

    def decode(self, post, list_size, max_deviation=None, num_threads=1, max_steps=0, band_fma=True):
        """-> (msgs uint8[count, msg_len], scores float32[count])"""
        post = np.ascontiguousarray(post, dtype=np.float32).reshape(-1, 40)
        if max_deviation is None:
            max_deviation = self.msg_len + self.mem_conv + 1      # reference default (:238-240)
        msgs = np.zeros((max(list_size, 1), self.msg_len), dtype=np.uint8)
        scores = np.zeros(max(list_size, 1), dtype=np.float32)
        cnt = ctypes.c_uint32(0)
        st = lib().lva_oracle_decode(self._h, post.ctypes.data, post.shape[0], list_size, max_deviation,
                                     num_threads, max_steps, int(band_fma), msgs.ctypes.data,
                                     scores.ctypes.data, ctypes.byref(cnt))
        if st != 0:
            raise OracleError(st)
        return msgs[:cnt.value].copy(), scores[:cnt.value].copy()


# ---------------------------------------------------------------- reference binary

def have_ref():
    return os.path.exists(REF_BIN) and os.access(REF_BIN, os.X_OK)


def _ref_args(mem_conv, rate, msg_len, sync_marker, sync_period):
    a = ["--msg-len", str(msg_len), "--mem-conv", str(mem_conv), "-r", str(rate)]
    if sync_marker:
        a += ["--sync-marker", sync_marker, "--sync-period", str(sync_period)]
    return a


def ref_encode(mem_conv, rate, msg_len, msgs, sync_marker="", sync_period=0):
    """run `viterbi_nanopore.out -m encode`; msgs: iterable of 0/1 arrays -> list of base strings"""
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.txt"), os.path.join(d, "out.txt")
        with open(fin, "w") as f:
            for m in msgs:
                f.write("".join(str(int(b)) for b in m) + "\n")
        r = subprocess.run([REF_BIN, "-m", "encode", "-i", fin, "-o", fout]
                           + _ref_args(mem_conv, rate, msg_len, sync_marker, sync_period),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode != 0:
            raise RuntimeError("reference encode failed rc=%d: %s" % (r.returncode, r.stdout.decode()[:200]))
        with open(fout) as f:
            return [ln.strip() for ln in f if ln.strip()]


def ref_decode(mem_conv, rate, msg_len, post, list_size, max_deviation=None, rc=False, num_threads=1,
               sync_marker="", sync_period=0, timeout=None):
    """run `viterbi_nanopore.out -m decode` on a posterior matrix -> (returncode, list of '0'/'1' strings)"""
    post = np.ascontiguousarray(post, dtype=np.float32)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.post"), os.path.join(d, "out.txt")
        post.tofile(fin)
        cmd = [REF_BIN, "-m", "decode", "-i", fin, "-o", fout, "-l", str(list_size), "-t", str(num_threads)]
        cmd += _ref_args(mem_conv, rate, msg_len, sync_marker, sync_period)
        if max_deviation is not None:
            cmd += ["--max-deviation", str(max_deviation)]
        if rc:
            cmd += ["--rc"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
        if r.returncode != 0 or not os.path.exists(fout):
            return r.returncode, []
        with open(fout) as f:
            return 0, [ln.rstrip("\n") for ln in f]


# ---------------------------------------------------------------------------------------------
# flappie basecall of the .post matrix + barcode localisation.
# PARITY UNPINNED for the basecall (see oracle/basecall_oracle.c); the barcode search restates
# helper.py / generate_decoded_lists.py, which cannot be imported here (h5py, distance, scrappy
# are absent) -- checked against brute force and hand-made cases only.
# ---------------------------------------------------------------------------------------------
def basecall(post):
    """flappie.c:273-285 on a float32[nblk][40] posterior matrix -> (basecall str, trans positions
    (the --trans-output-file lines), state path of nblk+1 entries, Viterbi score)."""
    post = np.ascontiguousarray(post, dtype=np.float32).reshape(-1, 40)
    nblk = post.shape[0]
    path = np.zeros(nblk + 1, dtype=np.int32)
    bases = np.zeros(max(nblk, 1), dtype=np.uint8)
    trans = np.zeros(max(nblk, 1), dtype=np.uint32)
    score = ctypes.c_float(0.0)
    n = lib().bc_oracle_basecall(post.ctypes.data, nblk, path.ctypes.data, bases.ctypes.data, trans.ctypes.data,
                                 ctypes.byref(score))
    if n < 0:
        raise OracleError(n)
    return bases[:n].tobytes().decode("ascii"), trans[:n].astype(np.int64), path, float(score.value)


def levenshtein(a, b):
    """distance.levenshtein (the `distance` package used by helper.py:5,183,187): unit-cost edit distance."""
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, cb in enumerate(b, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb))
        prev = cur
    return prev[len(b)]


def find_barcode_pos(basecall_str, trans_arr, start_barcode, end_barcode):
    """helper.find_barcode_pos_in_post (helper.py:157-210) on an in-memory basecall and trans list.
    Returns (start_pos, end_pos, min start distance, min end distance); (-1, -1, inf, inf) on failure.
    Where the reference would raise (an empty search range, helper.py:190-191 min() of an empty
    list) this returns the failure tuple too."""
    inf = float("inf")
    n, ls, le = len(basecall_str), len(start_barcode), len(end_barcode)
    if ls + le > n:                                                       # :177-179
        return (-1, -1, inf, inf)
    sd = [levenshtein(start_barcode, basecall_str[i:i + ls]) for i in range(n // 2 + 1 - ls)]       # :181-183
    ed = [levenshtein(end_barcode, basecall_str[i:i + le]) for i in range(n // 2, n - le)]          # :185-187
    if not sd or not ed:
        return (-1, -1, inf, inf)
    s_first = sd.index(min(sd))                                           # :190
    e_first = n // 2 + ed.index(min(ed))                                  # :191
    s_last = s_first + ls - 1
    start_pos = int(trans_arr[s_last + 1]) - 1                            # :193
    end_pos = int(trans_arr[e_first - 1]) - 1                             # :194
    if end_pos < start_pos:                                               # :206-208
        return (-1, -1, inf, inf)
    return (start_pos, end_pos, min(sd), min(ed))


def reverse_complement(dna):
    return "".join({"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}[c] for c in dna[::-1])   # helper.py:227-229


def locate_payload(post, start_barcode, end_barcode, min_len):
    """generate_decoded_lists.py:68-84 for one read: both orientations, the smaller barcode
    distance sum wins (forward on a draw); returns dict(ok, start_pos, end_pos, rc, dist_start, dist_end).
    min_len = MEM_CONV + MSG_LEN + 1 (:76)."""
    bc, trans, _, _ = basecall(post)
    f = find_barcode_pos(bc, trans, start_barcode, end_barcode)
    r = find_barcode_pos(bc, trans, reverse_complement(end_barcode), reverse_complement(start_barcode))   # :33-34, :69
    rc = f[2] + f[3] > r[2] + r[3]                                        # :71
    sp, ep, ds, de = r if rc else f
    ok = not (sp == -1 or ep - sp + 1 < min_len)                          # :76
    return dict(ok=ok, start_pos=sp, end_pos=ep, rc=bool(rc), dist_start=ds, dist_end=de, basecall=bc, trans=trans)
