"""ctypes loader of the product library liblva_hip.so (C ABI: include/lva_decoder.h)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "liblva_hip.so"

MAX_DEVIATION_DEFAULT = 0xFFFFFFFF
ABI_VERSION = 5                # LVA_ABI_VERSION of include/lva_decoder.h

ERRORS = {
    -1: "LVA_ERR_MEM_CONV", -2: "LVA_ERR_RATE", -3: "LVA_ERR_MSG_LEN", -4: "LVA_ERR_SYNC",
    -5: "LVA_ERR_TOO_MANY_STATES", -6: "LVA_ERR_POST_TOO_SHORT", -7: "LVA_ERR_MSG_TOO_LONG",
    -8: "LVA_ERR_NOMEM", -9: "LVA_ERR_HIP", -10: "LVA_ERR_ARG", -11: "LVA_ERR_NO_DEVICE",
    -12: "LVA_ERR_UNSUPPORTED",
}

# every symbol include/lva_decoder.h declares
EXPORTS = [
    "lva_version", "lva_abi_version", "lva_strerror", "lva_last_hip_error", "lva_code_describe", "lva_code_tables", "lva_band_table",
    "lva_encode", "lva_algorithmic_bytes", "lva_decoder_create", "lva_decoder_destroy",
    "lva_decode_batch", "lva_decode_batch_device", "lva_decoder_profile", "lva_decoder_set_launch_events", "lva_device_alloc",
    "lva_device_free", "lva_device_upload", "lva_device_synchronize",
    "lva_decode_windows_device", "lva_basecall_batch", "lva_basecall_batch_device", "lva_find_barcode_batch",
    "lva_locate_payload_batch", "lva_locate_payload_batch_device",
    "lva_rs_decode", "lva_rs_encode", "lva_rs_last_error",
]


class LvaError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        name = ERRORS.get(code, "LVA_ERR_%d" % code)
        msg = name
        try:
            msg += ": " + load_library().lva_strerror(code).decode()
        except Exception:  # pragma: no cover
            pass
        if detail:
            msg += " (" + detail + ")"
        super().__init__(msg)


class Config(ctypes.Structure):
    _fields_ = [("mem_conv", ctypes.c_int32), ("rate", ctypes.c_int32), ("msg_len", ctypes.c_uint32),
                ("list_size", ctypes.c_uint32), ("max_deviation", ctypes.c_uint32),
                ("sync_marker", ctypes.c_char_p), ("sync_period", ctypes.c_uint32),
                ("device", ctypes.c_int32), ("max_slots", ctypes.c_int32), ("kernel", ctypes.c_int32),
                ("mem_budget_bytes", ctypes.c_uint64)]


class CodeInfoStruct(ctypes.Structure):
    _fields_ = [("nstate_pos", ctypes.c_uint32), ("nstate_conv", ctypes.c_uint32), ("oligo_len", ctypes.c_uint32),
                ("msg_words", ctypes.c_uint32), ("initial_state", ctypes.c_uint32), ("final_state", ctypes.c_uint32),
                ("g0", ctypes.c_uint32), ("g1", ctypes.c_uint32), ("pattern_len", ctypes.c_int32),
                ("pattern", ctypes.c_uint8 * 16)]


class Profile(ctypes.Structure):
    _fields_ = [("step_kernel_ms", ctypes.c_double), ("total_ms", ctypes.c_double),
                ("step_launches", ctypes.c_uint64), ("read_steps", ctypes.c_uint64),
                ("algorithmic_bytes", ctypes.c_double), ("fixup_states", ctypes.c_uint64),
                ("fixup_reason", ctypes.c_uint64 * 4),
                ("slots", ctypes.c_int32), ("kernel", ctypes.c_int32),
                ("dominant_kernel_ms", ctypes.c_double), ("step_pair_ms", ctypes.c_double),
                ("timed_launches", ctypes.c_uint64), ("h2d_ms", ctypes.c_double), ("h2d_bytes", ctypes.c_uint64),
                ("overflow_steps", ctypes.c_uint64), ("working_bytes", ctypes.c_double)]


class PayloadPos(ctypes.Structure):
    _fields_ = [("start_pos", ctypes.c_int32), ("end_pos", ctypes.c_int32), ("dist_start", ctypes.c_int32),
                ("dist_end", ctypes.c_int32), ("rc", ctypes.c_int32), ("ok", ctypes.c_int32)]


_lib = None


def build_id():
    """The source hash the loaded library was built from (lva_version(): '... build <id>')."""
    return load_library().lva_version().decode().rsplit("build ", 1)[-1]


def library_path():
    # LVA_LIB_PATH: point at another build of the same library (kernel experiments)
    return os.environ.get("LVA_LIB_PATH") or os.path.join(_HERE, _LIB_NAME)


def load_library():
    """Load liblva_hip.so.  Raises if it has not been built: the product has no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C nanopore_dna_storage_amd/csrc`" % path)
    L = ctypes.CDLL(path)
    vp, i32, u32, u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint64
    cp = ctypes.c_char_p
    L.lva_version.restype = cp
    if not hasattr(L, "lva_abi_version"):            # a library built before the ABI carried a version (round 4 and earlier)
        raise ImportError("%s has no lva_abi_version (ABI version < 5), this package was written for %d: rebuild it" % (path, ABI_VERSION))
    L.lva_abi_version.restype = ctypes.c_int
    if L.lva_abi_version() != ABI_VERSION:
        raise ImportError("%s has ABI version %d, this package was written for %d (struct layouts of include/lva_decoder.h): rebuild it"
                          % (path, L.lva_abi_version(), ABI_VERSION))
    L.lva_strerror.restype = cp
    L.lva_strerror.argtypes = [ctypes.c_int]
    L.lva_last_hip_error.restype = cp
    L.lva_code_describe.argtypes = [i32, i32, u32, i32, cp, u32, ctypes.POINTER(CodeInfoStruct)]
    L.lva_code_tables.argtypes = [i32, i32, u32, i32, cp, u32, vp, vp, vp, vp, vp]
    L.lva_band_table.argtypes = [i32, i32, u32, i32, cp, u32, u32, u32, vp, vp]
    L.lva_encode.argtypes = [i32, i32, u32, vp, i32, vp]
    L.lva_algorithmic_bytes.argtypes = [i32, i32, u32, i32, cp, u32, u32, u32, u32, ctypes.POINTER(ctypes.c_double)]
    L.lva_decoder_create.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(vp)]
    L.lva_decoder_destroy.argtypes = [vp]
    L.lva_decoder_destroy.restype = None
    L.lva_decode_batch.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp]
    L.lva_decode_batch_device.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp]
    L.lva_decoder_profile.argtypes = [vp, ctypes.POINTER(Profile)]
    L.lva_decoder_set_launch_events.argtypes = [vp, i32]
    L.lva_device_alloc.argtypes = [vp, u64, ctypes.POINTER(vp)]
    L.lva_device_free.argtypes = [vp, vp]
    L.lva_device_upload.argtypes = [vp, vp, vp, u64]
    L.lva_device_synchronize.argtypes = [vp]
    L.lva_decode_windows_device.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, vp]
    L.lva_basecall_batch.argtypes = [vp, vp, vp, i32, vp, vp, vp]
    L.lva_basecall_batch_device.argtypes = [vp, vp, vp, i32, vp, vp, vp]
    L.lva_find_barcode_batch.argtypes = [vp, vp, vp, vp, i32, cp, cp, vp]
    L.lva_locate_payload_batch.argtypes = [vp, vp, vp, i32, cp, cp, u32, vp]
    L.lva_locate_payload_batch_device.argtypes = [vp, vp, vp, i32, cp, cp, u32, vp]
    u16 = ctypes.c_uint16
    L.lva_rs_decode.argtypes = [i32, vp, i32, i32, i32, vp, i32, u16, u16, vp, vp]
    L.lva_rs_encode.argtypes = [i32, vp, i32, i32, i32, u16, vp]
    L.lva_rs_last_error.restype = cp
    _lib = L
    return L
