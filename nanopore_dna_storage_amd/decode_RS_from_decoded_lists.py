"""Counterpart of the reference's decode_RS_from_decoded_lists.py (:1-66): NUM_TRIALS times draw NUM_READS_TO_USE of the
NUM_READS_TOTAL read numbers, run the CRC-8 / index filter over `list_<i>` of every drawn read that has a list, take the
per-index consensus payload (:37-51), RS-decode (rs_code.MainDecoder on the GPU) and compare with the original file.
The reference's constants are flags here (same names, lower case); --seed fixes the draws."""
import argparse
import math
import os
import random
import sys

from . import helper, rs_code


def build_parser():
    p = argparse.ArgumentParser(description="outer-code decoding from decoded lists")
    p.add_argument("--num_trials", type=int, default=10)
    p.add_argument("--list_size", type=int, default=8)
    p.add_argument("--num_reads_total", type=int, required=True)
    p.add_argument("--num_reads_to_use", type=int, required=True)
    p.add_argument("--bytes_per_oligo", type=int, default=18)
    p.add_argument("--decoded_lists_dir", type=str, required=True)
    p.add_argument("--rs_redundancy", type=float, default=0.3)
    p.add_argument("--pad", action="store_true")
    p.add_argument("--original_file", type=str, required=True)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--device", type=int, default=0)
    return p


def main(argv=None, out=sys.stdout):
    a = build_parser().parse_args(argv)
    rnd = random.Random(a.seed)
    with open(a.original_file, "rb") as f:
        original = f.read()
    data_file_size = len(original)
    data_size_padded = math.ceil(data_file_size / a.bytes_per_oligo) * a.bytes_per_oligo
    msg_len, num_oligos_data, num_oligos_RS, num_oligos = helper.compute_parameters(a.bytes_per_oligo, a.rs_redundancy, data_size_padded, a.pad)
    print("NUM_READS_TO_USE:", a.num_reads_to_use, file=out)
    print("list size:", a.list_size, file=out)
    num_successes = 0
    for _ in range(a.num_trials):
        lists = []
        for i in rnd.sample(list(range(a.num_reads_total)), a.num_reads_to_use):
            list_file = os.path.join(a.decoded_lists_dir, "list_" + str(i))
            if os.path.isfile(list_file):
                with open(list_file) as f:
                    lists.append([ln.rstrip("\n") for ln in f.readlines()][:a.list_size])
        data, passed = rs_code.decode_from_lists(lists, a.bytes_per_oligo, num_oligos_RS, num_oligos, pad=a.pad, device=a.device)
        ok = passed > 0 and data[:data_file_size] == original     # no read passed the filter: the reference dies in MainDecoder
        num_successes += int(ok)
        print("Success" if ok else "Failure", file=out)
    print("NUM_TRIALS", a.num_trials, file=out)
    print("num_successes", num_successes, file=out)
    return num_successes


if __name__ == "__main__":
    main()
