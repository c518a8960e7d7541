"""Counterpart of the reference's compute_error_rate_from_decoded_lists.py (:1-61): walks DECODED_LISTS_DIR for
`list_<i>` files (what generate_decoded_lists writes), runs the CRC-8 / index filter over the first LIST_SIZE entries
of each (helper.decode_list_CRC_index) and prints the reference's four tallies.  The reference's constants at the top of
the script are flags here (same names, lower case)."""
import argparse
import os
import re
import sys

from . import helper


def build_parser():
    p = argparse.ArgumentParser(description="error rates of decoded lists against the encoder's input")
    p.add_argument("--list_size", type=int, default=8)
    p.add_argument("--decoded_lists_dir", type=str, required=True)
    p.add_argument("--conv_input_file", type=str, required=True)
    p.add_argument("--pad", action="store_true")
    p.add_argument("--bytes_per_oligo", type=int, default=18)
    return p


def read_lists(directory):
    """every list_<i> file of the directory, in os.listdir order like the reference (:24-30) -> [(name, [entries])]"""
    out = []
    for filename in os.listdir(directory):
        if not re.fullmatch(r"list_\d+", filename):      # (the reference takes every name starting with list_: :26)
            continue
        with open(os.path.join(directory, filename)) as f:
            out.append((filename, [ln.rstrip("\n") for ln in f.readlines()]))
    return out


def main(argv=None, out=sys.stdout):
    a = build_parser().parse_args(argv)
    print("list size:", a.list_size, file=out)
    with open(a.conv_input_file) as f:
        conv_input_list = [s.rstrip("\n") for s in f.readlines()]
    print("num_oligos", len(conv_input_list), file=out)
    t = helper.tally_decoded_lists([lst for _, lst in read_lists(a.decoded_lists_dir)], conv_input_list, a.bytes_per_oligo,
                                   a.pad, a.list_size)
    for k in ("num_reads", "num_correct", "num_erasure_CRC_index", "num_error_CRC_index"):
        print(k + ":", t[k], file=out)
    return t


if __name__ == "__main__":
    main()
