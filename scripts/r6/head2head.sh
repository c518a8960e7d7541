#!/bin/bash
# round 6: libraries head to head on ONE box, the driver's command shape (shorter): scripts/r6/head2head.sh lib1 lib2 ... (default = the product library)
out=gpurun_out/r6/h2h; mkdir -p $out
for rep in 1 2; do
for v in "$@"; do
  lib=variants/$v.so; [ $v = default ] && lib=nanopore_dna_storage_amd/liblva_hip.so
  LVA_LIB_PATH=$lib python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-cross-check > $out/${v}_$rep.json 2> $out/${v}_$rep.err || tail -3 $out/${v}_$rep.err
  python3 - <<PY
import json
j=json.loads(open("$out/${v}_$rep.json").read().strip().splitlines()[-1])
print("%-10s run $rep  reads/s %.2f  launch %.3f ms  frac %.3f  build %s" % ("$v", j["value"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"], j["library"]["build_id"]))
PY
done; done
