"""Counterpart of the reference's generate_decoded_lists.py for this path.

The reference walks raw signals in an HDF5 file: fast5 -> flappie -> barcode search -> post
truncation -> decoder subprocess per read (generate_decoded_lists.py:50-98).  HDF5, flappie and
its weights are unavailable here, so this driver starts from what flappie + the barcode search
produce: one .post file per read plus the [start, end] block range and orientation, listed in a
tab-separated manifest (--post_manifest):

    readid <TAB> ref <TAB> post_path <TAB> start_pos <TAB> end_pos <TAB> rc(0/1)

start_pos = -1 marks a read whose barcodes were not found.  Everything downstream is the
reference's: the skip rule (:76), helper.truncate_post_file semantics (:84), one decoded list
file OUT_PREFIX_i per read (:85-90, max-deviation 20), and the info file "readid<TAB>ref" (:56).
The original flags are kept (those describing the unavailable input side are accepted and ignored).
"""
import argparse
import sys

from . import helper
from .decoder import Decoder


def build_parser():
    p = argparse.ArgumentParser(description="generate decoded lists from posterior matrices")
    p.add_argument("--post_manifest", type=str, required=True)
    p.add_argument("--out_prefix", type=str, required=True)
    p.add_argument("--info_file", type=str, required=True)
    p.add_argument("--mem_conv", type=int, required=True)
    p.add_argument("--msg_len", type=int, required=True)
    p.add_argument("--rate_conv", type=int, required=True)
    p.add_argument("--list_size", type=int, required=True)
    p.add_argument("--num_threads", type=int, default=1)
    p.add_argument("--hdf_file", type=str, default=None)        # input side of the reference: unused
    p.add_argument("--read_id_file", type=str, default=None)
    p.add_argument("--start_barcode", type=str, default=None)
    p.add_argument("--end_barcode", type=str, default=None)
    p.add_argument("--max_deviation", type=int, default=20)
    p.add_argument("--device", type=int, default=0)
    return p


def run(args, out=sys.stdout):
    rows = []
    with open(args.post_manifest) as f:
        for line in f:
            line = line.rstrip("\n")
            if line:
                rid, ref, path, s, e, rc = line.split("\t")
                rows.append((rid, ref, path, int(s), int(e), rc not in ("0", "", "False")))
    posts, rcs, keep = [], [], []
    with open(args.info_file, "w") as f_info:
        for i, (rid, ref, path, s, e, rc) in enumerate(rows):
            print("i:", i, file=out); print(rid, file=out); print(ref, file=out)
            f_info.write(rid + "\t" + ref + "\n")
            if s == -1 or e - s + 1 < args.mem_conv + args.msg_len + 1:
                print("Failure in barcode removing.", file=out)
                continue
            posts.append(helper.truncate_post(helper.read_post_file(path), s, e))
            rcs.append(rc)
            keep.append(i)
    written = 0
    if posts:
        with Decoder(args.mem_conv, args.rate_conv, args.msg_len, list_size=args.list_size,
                     max_deviation=args.max_deviation, device=args.device) as dec:
            results = dec.decode(posts, rc=rcs)
        for i, res in zip(keep, results):
            if isinstance(res, int):
                continue       # the reference decoder aborts (no output file) on such a read
            with open(args.out_prefix + "_" + str(i), "w") as f:
                for row in res[0]:
                    f.write("".join("1" if b else "0" for b in row) + "\n")
            written += 1
    return written


def main(argv=None):
    args = build_parser().parse_args(argv)
    print(args)
    run(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
