"""MI355X-native list-Viterbi decoder for nanopore DNA storage.

Drop-in for the convolutional-code decode path of shubhamchandak94/nanopore_dna_storage
(viterbi/viterbi_convolutional_code.cpp driven by helper.py / simulator.py /
generate_decoded_lists.py).  Python host code over a C ABI (include/lva_decoder.h) into
hand-written HIP kernels for gfx950.  There is no CPU fallback: decoding raises when the
HIP library or a GPU is missing.
"""
from ._lib import LvaError, library_path, load_library  # noqa: F401
from .decoder import (CodeInfo, Decoder, algorithmic_bytes, band_table, code_info, code_tables, encode,  # noqa: F401
                      bases_to_str, str_to_bits)
