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

Beyond the reference:
  --chunk C  the manifest is decoded C reads at a time (default 4096): bounded host memory, and every finished
             chunk's list files are on disk (each written to a temporary name and renamed), so a killed run
             resumes from its last finished chunk
  --resume   skip reads whose OUT_PREFIX_i already exists (the reference re-runs by computing the read-ids
             that are not done yet, util/extra/pick_new_reads.py:11-18; one output file per read is its
             checkpoint)
  --gpus N   the reference scales by running N copies of the driver on disjoint read-id files and merging the
             lists (util/extra/generate_read_id_files.py:23-36, merge_lists.py:11-21); here N rank processes
             (one per GPU, torch.distributed over RCCL) each decode a strided share of the manifest and rank 0
             gathers the lists (sharding.decode_sharded) and writes every output file.
"""
import argparse
import os
import sys

import numpy as np

from . import helper, sharding
from .decoder import Decoder

BARCODE_FAILURE = -100          # travels in the gather in place of a list: "Failure in barcode removing."


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
    p.add_argument("--resume", action="store_true")
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--chunk", type=int, default=4096, help="reads decoded (and written) per pass over the manifest")
    return p


def read_manifest(args):
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
    return rows


def decode_rows(args, rows, dec):
    """the per-read work of generate_decoded_lists.py:50-98 for `rows` on one GPU (`dec`: this rank's Decoder)
    -> (results: one of list | negative error code | BARCODE_FAILURE per row, located: {row index: window dict})"""
    min_len = args.mem_conv + args.msg_len + 1
    results = [BARCODE_FAILURE] * len(rows)
    located = {}
    if not rows:
        return results, located
    # rows without a window: basecall + barcode search + decode on the device (:68-89)
    todo = [i for i, r in enumerate(rows) if r[3] is None]
    if todo:
        chain = dec.decode_with_barcodes([helper.read_post_file(rows[i][2]) for i in todo],
                                         args.start_barcode, args.end_barcode)
        for i, (loc, res) in zip(todo, chain):
            located[i] = dict(start_pos=int(loc["start_pos"]), end_pos=int(loc["end_pos"]), rc=bool(loc["rc"]))
            if res is not None:
                results[i] = res
    # rows with a window from an earlier search (:76, :84)
    keep = [i for i, r in enumerate(rows) if r[3] is not None and not (r[3] == -1 or r[4] - r[3] + 1 < min_len)]
    if keep:
        posts = [helper.truncate_post(helper.read_post_file(rows[i][2]), rows[i][3], rows[i][4]) for i in keep]
        for i, res in zip(keep, dec.decode(posts, rc=[rows[i][5] for i in keep])):
            results[i] = res
    return results, located


TMP_PREFIX = ".tmp-"
_HOST = "".join(ch if ch.isalnum() else "." for ch in (os.uname().nodename or "host"))     # (temporary names say whose they are)
PUBLISH_BATCH = 256                 # list files flushed and renamed together


def write_list_temp(path, msgs):
    """one decoded list under a temporary name beside its final one (never matches OUT_PREFIX_<i>: list consumers take every
    list_*); publish_list_files moves a whole chunk's files into place.  -> (temporary name, final name)"""
    d, base = os.path.split(path)
    tmp = os.path.join(d, TMP_PREFIX + base + "-%s-%d" % (_HOST, os.getpid()))
    with open(tmp, "w") as f:
        for row in msgs:
            f.write("".join("1" if b else "0" for b in row) + "\n")
    return tmp, path


def publish_list_files(pairs):
    """A chunk's list files become visible under their final names only after their data is durable: a run killed in the middle
    never leaves a truncated OUT_PREFIX_i that --resume would take for a finished read.  Durability is paid once per chunk --
    every file is written first, then flushed (after the first flush has committed the journal the others find nothing left
    to write), then renamed, then the directory is flushed -- not one write barrier per read: on small trellises the decoder
    produces thousands of lists a second (the reference never flushes at all)."""
    published = 0
    try:
        for k in range(0, len(pairs), PUBLISH_BATCH):        # (sub-batches: a failure loses a few hundred lists, not a whole chunk)
            part = pairs[k:k + PUBLISH_BATCH]
            for tmp, _ in part:
                fd = os.open(tmp, os.O_RDONLY)
                try:
                    os.fsync(fd)                         # the rename must not become durable before the data
                finally:
                    os.close(fd)
            for tmp, path in part:
                os.replace(tmp, path)
                published += 1
            for d in {os.path.dirname(path) or "." for _, path in part}:
                fd = os.open(d, os.O_RDONLY)
                try:
                    os.fsync(fd)
                finally:
                    os.close(fd)
    finally:
        for tmp, _ in pairs[published:]:                     # whatever was not published does not stay behind
            try:
                os.remove(tmp)
            except OSError:
                pass
    return published


def write_list_file(path, msgs):
    """one decoded list, durable before it is visible (a chunk of them: write_list_temp + publish_list_files)"""
    publish_list_files([write_list_temp(path, msgs)])


def remove_stale_temp_files(out_prefix):
    """temporary list files a killed run left behind (start-up of every run): exactly the names write_list_temp makes for THIS
    prefix ON THIS HOST -- .tmp-<base>_<i>-<host>-<pid>, so that a run on prefix `list` leaves `list_b`'s files alone and a run
    on another host that shares the directory keeps its files -- and only when the process that made them is gone (a
    concurrent run on the same prefix may be alive)."""
    import re
    d, base = os.path.split(out_prefix)
    pat = re.compile(re.escape(TMP_PREFIX + base) + r"_\d+-" + re.escape(_HOST) + r"-(\d+)$")
    for name in os.listdir(d or "."):
        mt = pat.match(name)
        if not mt:
            continue
        pid = int(mt.group(1))
        if pid != os.getpid():
            try:
                os.kill(pid, 0)                  # (signal 0: existence check only)
                continue                         # its writer is alive
            except ProcessLookupError:
                pass
            except PermissionError:
                continue                         # alive, another user's
        try:
            os.remove(os.path.join(d or ".", name))
        except OSError:
            pass


def run(args, out=sys.stdout, dist=None, device=None, coll_dev=None):
    """The manifest is walked in chunks of --chunk reads (bounded host memory: a chunk's posterior matrices are the only
    ones loaded; per-chunk progress: every finished chunk's list files are on disk, which is what --resume picks up
    after a crash).  Per chunk: strided shards over the ranks, decode, gather on rank 0, rank 0 writes the files."""
    rows = read_manifest(args)
    n = len(rows)
    if dist is None or dist.get_rank() == 0:
        remove_stale_temp_files(args.out_prefix)
    done = [args.resume and os.path.exists(args.out_prefix + "_" + str(i)) for i in range(n)]
    work = [i for i in range(n) if not done[i]]
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    chunk = max(1, int(args.chunk))
    written = 0
    cursor = 0                                  # rank 0: rows [0, cursor) have been reported
    f_info = open(args.info_file, "w") if rank == 0 else None

    def report(upto, results, loc):
        nonlocal cursor, written
        pending = []                                       # this chunk's list files, published together
        for i in range(cursor, upto):
            rid, ref = rows[i][0], rows[i][1]
            print("i:", i, file=out); print(rid, file=out); print(ref, file=out)
            f_info.write(rid + "\t" + ref + "\n")
            if done[i]:
                continue                                   # --resume: the list file of an earlier run stands
            if i in loc:
                print("start_pos_in_post", loc[i]["start_pos"], file=out); print("end_pos_in_post", loc[i]["end_pos"], file=out)
                print("--rc" if loc[i]["rc"] else "", file=out)
            r = results.get(i, BARCODE_FAILURE)
            if isinstance(r, (int, np.integer)):
                if r == BARCODE_FAILURE:
                    print("Failure in barcode removing.", file=out)
                continue       # (other codes: the reference decoder aborts, no output file, on such a read)
            pending.append(write_list_temp(args.out_prefix + "_" + str(i), r[0]))
        written += publish_list_files(pending)
        cursor = max(cursor, upto)
        f_info.flush()

    if dist is not None:       # same flags, same library build, same code tables on every rank (the reference assumes it: merge_lists.py:11-21)
        sharding.assert_same_configuration(dist, sharding.configuration_record(args.mem_conv, args.rate_conv, args.msg_len, args.list_size,
                                                                               args.max_deviation), device=coll_dev)
    with Decoder(args.mem_conv, args.rate_conv, args.msg_len, list_size=args.list_size, max_deviation=args.max_deviation,
                 device=args.device if device is None else device) as dec:
        for k in range(0, len(work), chunk):
            part = work[k:k + chunk]
            shards = sharding.shard_strided(len(part), world)
            mine = [part[int(j)] for j in shards[rank]]
            res, loc = decode_rows(args, [rows[i] for i in mine], dec)
            loc = {mine[j]: v for j, v in loc.items()}
            if dist is not None:
                res = sharding.gather_results(res, shards, args.list_size, args.msg_len, dist=dist, device=coll_dev)
                locs = [None] * world if rank == 0 else None
                dist.gather_object(loc, locs, dst=0)
                if rank == 0:
                    loc = {kk: v for d in locs for kk, v in d.items()}
            if rank == 0:
                report(part[-1] + 1, {part[j]: r for j, r in enumerate(res)}, loc)
    if rank == 0:
        report(n, {}, {})
        f_info.close()
    return written


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = build_parser().parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # start the ranks as child processes; this parent never touches the GPU
        return sharding.launch_ranks("nanopore_dna_storage_amd.generate_decoded_lists", argv, args.gpus, module=True)
    dist, rank, world, device, coll_dev = sharding.init_rank()
    if rank == 0:
        print(args)
        if dist is not None:
            print("ranks %d backend %s" % (world, dist.get_backend()))
    if dist is None:
        run(args)
    else:
        run(args, dist=dist, device=device, coll_dev=coll_dev)
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
