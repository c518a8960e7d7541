"""Counterpart of the reference's simulator.py for this path: same CLI flags, same per-trial
and summary statistics (simulator.py:15-27, :86-126), with the scrappy + flappie signal chain
replaced by the deterministic synthetic posterior generator (synth.py; neither dependency can
run here) and all trials decoded in one GPU batch.

    python -m nanopore_dna_storage_amd.simulator --num_trials 100 --list_size 1 --mem_conv 6 \
        --rate 1 --msg_len 180
"""
import argparse
import math
import sys

import numpy as np

from . import helper, synth
from .decoder import Decoder, bases_to_str


def build_parser():
    p = argparse.ArgumentParser(description="Simulation for convolutional code.")
    p.add_argument("--num_trials", type=int, default=100)
    p.add_argument("--list_size", type=int, default=1)
    p.add_argument("--num_thr", type=int, default=8)          # accepted, unused on the GPU
    p.add_argument("--mem_conv", type=int, default=6)
    p.add_argument("--rate", type=int, default=1)
    p.add_argument("--msg_len", type=int, default=100)
    p.add_argument("--deepsimdwell", type=str, default="False")   # accepted, unused (no signal simulator)
    p.add_argument("--reversecomp", type=str, default="False")
    p.add_argument("--syn_sub_prob", type=float, default=0.002)
    p.add_argument("--syn_del_prob", type=float, default=0.0085)
    p.add_argument("--syn_ins_prob", type=float, default=0.0005)
    # extensions
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--margin", type=float, default=6.0)
    p.add_argument("--max_deviation", type=int, default=20)
    p.add_argument("--device", type=int, default=0)
    return p


def run(args, out=sys.stdout, decoder=None):
    revcomp = args.reversecomp != "False"
    reads = [synth.make_read(args.mem_conv, args.rate, args.msg_len, args.seed + i, rc=revcomp,
                             margin=args.margin, sub=args.syn_sub_prob, dele=args.syn_del_prob,
                             ins=args.syn_ins_prob) for i in range(args.num_trials)]
    own = decoder is None
    if own:
        decoder = Decoder(args.mem_conv, args.rate, args.msg_len, list_size=args.list_size,
                          max_deviation=args.max_deviation, device=args.device)
    try:
        results = decoder.decode([r["post"] for r in reads], rc=[revcomp] * len(reads))
    finally:
        if own:
            decoder.close()
    top, lst, ham, ham8, ham16, edit = [], [], [], [], [], []
    n = args.msg_len
    for rd, res in zip(reads, results):
        msg = "".join(map(str, rd["msg"]))
        print(msg, file=out)
        print("len(seq):", len(bases_to_str(rd["oligo"])), file=out)
        decoded = ["".join(map(str, row)) for row in res[0]] if not isinstance(res, int) else []
        if not decoded:
            raise RuntimeError("decoder produced no list (the reference would crash reading its output file)")
        print("Top message:", file=out)
        print(decoded[0], file=out)
        print("List size:", len(decoded), file=out)
        top.append(decoded[0] == msg)
        lst.append(msg in decoded)
        print("Top correct:", top[-1], file=out)
        print("List correct:", lst[-1], file=out)
        ham.append(helper.hamming(msg, decoded[0]))
        ham8.append(sum(decoded[0][i * 8:(i + 1) * 8] != msg[i * 8:(i + 1) * 8] for i in range(math.ceil(n / 8))))
        ham16.append(sum(decoded[0][i * 16:(i + 1) * 16] != msg[i * 16:(i + 1) * 16] for i in range(math.ceil(n / 16))))
        print("Hamming distance of top:", ham[-1], file=out)
        print("Hamming distance of top (8 blocks):", ham8[-1], file=out)
        print("Hamming distance of top (16 blocks):", ham16[-1], file=out)
        edit.append(helper.levenshtein(msg, decoded[0]))
        print("Edit distance", edit[-1], file=out)
        if not top[-1]:
            print("Error pattern (original, errors):", file=out)
            print(msg, file=out)
            print("".join(m if m == d else "*" for m, d in zip(msg, decoded[0])), file=out)
    T = args.num_trials
    stats = {
        "Number total": T,
        "Number top correct": sum(top),
        "Number list correct": sum(lst),
        "Average bit error rate of top": sum(ham) / (n * T) if T else 0.0,
        "Average 8 block error rate of top": sum(ham8) / (math.ceil(n / 8) * T) if T else 0.0,
        "Average 16 block error rate of top": sum(ham16) / (math.ceil(n / 16) * T) if T else 0.0,
        "Average edit distance rate of top": sum(edit) / (n * T) if T else 0.0,
    }
    print("Summary statistics:", file=out)
    for k, v in stats.items():
        print(k + ":", v, file=out)
    return stats


def main(argv=None):
    args = build_parser().parse_args(argv)
    print(args)
    run(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
