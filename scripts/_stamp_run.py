import sys, ctypes, numpy as np
sys.path.insert(0, '.')
import nanopore_dna_storage_amd as pkg
from nanopore_dna_storage_amd import synth, _lib
reads = [synth.make_read(11, 5, 180, seed=1000 + i, rc=bool(i & 1), margin=6.0 if i % 4 else 3.0) for i in range(32)]
with pkg.Decoder(11, 5, 180, list_size=8, max_deviation=20, max_slots=32) as dec:
    dec.decode([x["post"] for x in reads], rc=[x["rc"] for x in reads])
    out = (ctypes.c_ulonglong * 8)()
    _lib.load_library().lva_debug_stamps(out)
    p = dec.profile()
    w = max(out[3], 1)
    print("per merge-wave: setup(stay loads+heads) %.0f  loop %.0f  tail+gather %.0f  | waves %d | flip total %.0f flop total %.0f | staging per wave %.0f (%d waves) | kernel ms/launch %.3f" % (
        out[0] / w, out[1] / w, out[2] / w, out[3], out[4] / w, out[5] / w, out[6] / max(out[7], 1), out[7], p["step_kernel_ms"] / p["step_launches"]))
