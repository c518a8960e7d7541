"""Timing of SURVEY 8(f) row N3 on the GPU: basecall + barcode localisation of N barcoded reads of the
benchmark shape (m=11 r=5/6 msg_len=180), posteriors resident in HBM.  Prints one JSON line."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
import nanopore_dna_storage_amd as pkg
from nanopore_dna_storage_amd import synth
SB, EB = "CACCTGTGCTGCGTCAGGCTGTGTC", "GCTGTCCGTTCCGCATTGACACGGC"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
base = [synth.make_barcoded_read(11, 5, 180, 4000 + i, SB, EB, rc=bool(i & 1), margin=5.0, flank=(20, 60)) for i in range(64)]
posts = [base[i % 64]["post"] for i in range(n)]
with pkg.Decoder(11, 5, 180, list_size=8, max_deviation=20, max_slots=1) as dec:
    dev, off = dec.upload(posts)
    dec.locate_payload_resident(dev, off, SB, EB)          # warm-up
    t0 = time.time(); loc = dec.locate_payload_resident(dev, off, SB, EB); dt = time.time() - t0
    t1 = time.time(); dec.basecall_resident(dev, off); dtb = time.time() - t1
    dec.free(dev)
blocks = int(off[-1])
print(json.dumps({"stage": "locate_payload (basecall + barcode search, both orientations)", "reads": n, "blocks": blocks,
                  "seconds": dt, "reads_per_s": n / dt, "post_GBps": blocks * 160 / dt / 1e9,
                  "basecall_only_seconds_incl_copy_back": dtb, "ok_reads": sum(x["ok"] for x in loc)}))
