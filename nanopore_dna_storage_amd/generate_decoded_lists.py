"""Counterpart of the reference's generate_decoded_lists.py for this path.

The reference walks raw signals in an HDF5 file: fast5 -> flappie -> barcode search -> post
truncation -> decoder subprocess per read (generate_decoded_lists.py:50-98).  HDF5, flappie and
its weights are unavailable here, so this driver starts from flappie's --post-output-file: one
.post file per read, listed in a tab-separated manifest (--post_manifest), either

    readid <TAB> ref <TAB> post_path
        the untruncated matrix: with --start_barcode / --end_barcode the basecall, the barcode
        search in both orientations and the choice between them (:68-79) run on the GPU
        (Decoder.decode_with_barcodes) and the payload window is decoded in place, or
    readid <TAB> ref <TAB> post_path <TAB> start_pos <TAB> end_pos <TAB> rc(0/1)
        the [start, end] block range and orientation found by an earlier barcode search.

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
            if not line:
                continue
            cols = line.split("\t")
            if len(cols) == 3:
                if not (args.start_barcode and args.end_barcode):
                    raise SystemExit("3-column manifest rows need --start_barcode and --end_barcode")
                rows.append((cols[0], cols[1], cols[2], None, None, None))
            else:
                rid, ref, path, s, e, rc = cols
                rows.append((rid, ref, path, int(s), int(e), rc not in ("0", "", "False")))
    min_len = args.mem_conv + args.msg_len + 1
    results = {}
    with Decoder(args.mem_conv, args.rate_conv, args.msg_len, list_size=args.list_size,
                 max_deviation=args.max_deviation, device=args.device) as dec:
        # rows without a window: basecall + barcode search + decode on the device (:68-89)
        todo = [i for i, r in enumerate(rows) if r[3] is None]
        located = {}
        if todo:
            chain = dec.decode_with_barcodes([helper.read_post_file(rows[i][2]) for i in todo],
                                             args.start_barcode, args.end_barcode)
            for i, (loc, res) in zip(todo, chain):
                located[i] = loc
                if res is not None:
                    results[i] = res
        # rows with a window from an earlier search (:76, :84)
        keep = [i for i, r in enumerate(rows) if r[3] is not None and not (r[3] == -1 or r[4] - r[3] + 1 < min_len)]
        if keep:
            posts = [helper.truncate_post(helper.read_post_file(rows[i][2]), rows[i][3], rows[i][4]) for i in keep]
            for i, res in zip(keep, dec.decode(posts, rc=[rows[i][5] for i in keep])):
                results[i] = res
    written = 0
    with open(args.info_file, "w") as f_info:
        for i, (rid, ref, path, s, e, rc) in enumerate(rows):
            print("i:", i, file=out); print(rid, file=out); print(ref, file=out)
            f_info.write(rid + "\t" + ref + "\n")
            if i in located:
                loc = located[i]
                print("start_pos_in_post", loc["start_pos"], file=out); print("end_pos_in_post", loc["end_pos"], file=out)
                print("--rc" if loc["rc"] else "", file=out)
            res = results.get(i)
            if res is None:
                print("Failure in barcode removing.", file=out)
                continue
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
