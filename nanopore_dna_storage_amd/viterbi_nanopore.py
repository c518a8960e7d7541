"""Drop-in for the reference executable viterbi/viterbi_nanopore.out
(viterbi_convolutional_code.cpp:137-262): same flags, same file formats, same stdout messages,
decode on the GPU.

    python -m nanopore_dna_storage_amd.viterbi_nanopore -m decode -i X.post -o OUT \
        --msg-len 180 --mem-conv 11 -r 5 -l 8 -t 8 [--rc] --max-deviation 20

Exit codes follow the reference: 0 ok; 255 (`return -1`) for parameter errors, with the message
and usage on stdout; 134 (abort on an uncaught std::runtime_error) for a post matrix that is too
short / too many states, with no output file written; 1 (message on stderr, no output file) for
failures the reference cannot have: no usable GPU, device memory, a HIP error, a list size the C ABI refuses.
One process per read pays interpreter start-up, HIP initialisation and the decoder's tables and trellis
allocation every time (measured in DESIGN.md): batch callers should use Decoder.decode or
generate_decoded_lists, which keep one decoder for all reads.  Callers of the reference pass '' as an
argv element when the read is not reverse-complemented (simulator.py:82-85); it is ignored here
as cxxopts ignores it.
"""
import argparse
import sys

import numpy as np

from . import helper
from ._lib import LvaError
from .decoder import Decoder, bases_to_str, code_info, encode, str_to_bits

USAGE = """Viterbi decoder for nanopore dna storage codes
Usage:
  viterbi_nanopore [OPTION...]

  -m, --mode arg           Mode: encode, decode
  -i, --infile arg         Infile with message (encoding) or posterior matrix
                           (decoding)
  -o, --outfile arg        Outfile with encoded/decoded message (list)
      --msg-len arg        Message length
      --mem-conv arg       Code memory for convolutional code
      --sync-marker arg    Sync marker for convolutional code decoding as
                           string (e.g. 110) (default '') (default: "")
      --sync-period arg    Sync marker period for convolutional code decoding
  -l, --list-size arg      List size for convolutional code decoding (default
                           1) (default: 1)
  -r, --rate arg           Rate of convolutional code: options 1 (1/2), 2
                           (2/3), 3 (3/4), 4 (4/5), 5 (5/6), 7 (7/8) (default 1).
                           Use standard puncturing patterns, expects
                           appropriate padding (at most 1 bit needed) to make
                           output length even. (default: 1)
      --max-deviation arg  Max allowable deviation of st_pos around its
                           expected value during decoding (tradeoff b/w speed
                           and accuracy) (default: infinite)
      --rc                 Reverse complement read (for decoding)
  -t, --num-thr arg        Number of threads for convolutional code decoding
                           (default 1) (default: 1)
  -h, --help               Display this message
"""

_PARAM_MESSAGES = {
    -1: "Invalid mem_conv (allowed: 6, 8, 11, 14)",
    -2: "Invalid rate parameter (allowed: 1, 3, 5, 7)",
    -3: "Output length not even. Try padding with a single 0 at end.",
    -4: "Invalid sync marker",
}


class _Parser(argparse.ArgumentParser):
    def error(self, message):          # cxxopts throws on unknown options -> abort
        raise SystemExit(134)


def _parse(argv):
    p = _Parser(add_help=False)
    p.add_argument("-m", "--mode")
    p.add_argument("-i", "--infile")
    p.add_argument("-o", "--outfile")
    p.add_argument("--msg-len", type=int)
    p.add_argument("--mem-conv", type=int)
    p.add_argument("--sync-marker", default="")
    p.add_argument("--sync-period", type=int, default=0)
    p.add_argument("-l", "--list-size", type=int, default=1)
    p.add_argument("-r", "--rate", type=int, default=1)
    p.add_argument("--max-deviation", type=int, default=None)
    p.add_argument("--rc", action="store_true")
    p.add_argument("-t", "--num-thr", type=int, default=1)
    p.add_argument("-h", "--help", action="store_true")
    p.add_argument("--device", type=int, default=0)         # extension: GPU ordinal
    return p.parse_args([a for a in argv if a != ""])


def main(argv=None, out=sys.stdout):
    a = _parse(sys.argv[1:] if argv is None else argv)
    if a.help:
        print(USAGE, file=out)
        return 0
    if not a.mode or not a.infile or not a.outfile:
        print("Invalid options.", file=out); print(USAGE, file=out)
        return 255
    if a.mode not in ("encode", "decode"):
        print("Invalid mode.", file=out); print(USAGE, file=out)
        return 255
    if a.mem_conv is None:
        print("Memory of convolutional code not specified.", file=out); print(USAGE, file=out)
        return 255
    if a.msg_len is None:
        print("msg-len not specified.", file=out); print(USAGE, file=out)
        return 255
    rc = a.rc and a.mode == "decode"
    if rc:
        print("Reverse complement flag detected.", file=out)
    try:
        code_info(a.mem_conv, a.rate, a.msg_len, rc, a.sync_marker, a.sync_period)
    except LvaError as e:
        print(_PARAM_MESSAGES.get(e.code, str(e)), file=out); print(USAGE, file=out)
        return 255

    if a.mode == "encode":
        msgs = []
        with open(a.infile) as f:                 # read_bit_array (:501-524)
            for line in f.read().split("\n")[:-1]:
                if any(ch not in "01" for ch in line):
                    return 134                    # "invalid character in input file"
                msgs.append(line)
        for m in msgs:
            if len(m) != a.msg_len:
                print("Message length does not match msg_len parameter.", file=out)
                return 255
        with open(a.outfile, "w") as f:
            if msgs:
                oligos = encode(a.mem_conv, a.rate, a.msg_len, np.stack([str_to_bits(m) for m in msgs]))
                for o in oligos:
                    f.write(bases_to_str(o) + "\n")
        return 0

    try:
        post = helper.read_post_file(a.infile)
    except OSError:
        return 134
    try:
        with Decoder(a.mem_conv, a.rate, a.msg_len, list_size=a.list_size, max_deviation=a.max_deviation,
                     sync_marker=a.sync_marker, sync_period=a.sync_period, device=a.device, max_slots=1) as dec:
            res = dec.decode([post], rc=[rc])[0]
    except LvaError as e:
        if e.code in (-5, -7):
            return 134                            # runtime_error -> abort, no output file
        # errors the reference cannot have (no GPU, out of device memory, HIP failure, list size 0 or > 65535):
        # message on stderr, exit code 1, no output file -- never a Python traceback
        print("viterbi_nanopore: %s" % e, file=sys.stderr)
        return 1
    if isinstance(res, int):
        return 134                                # "Too small post matrix"
    with open(a.outfile, "w") as f:               # :248-253
        for row in res[0]:
            f.write("".join("1" if b else "0" for b in row) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
