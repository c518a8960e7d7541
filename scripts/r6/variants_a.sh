#!/bin/bash
# round 6, experiment A: where do the three lazy instances spend their time?  Variants built in the build container
# (scripts/r6/build_variants_a.sh), run here: kernel trace of a short benchmark run per variant.
export TMPDIR=/tmp
out=gpurun_out/r6/varA; mkdir -p $out
for v in "$@"; do
  lib=variants/$v.so; [ $v = default ] && lib=nanopore_dna_storage_amd/liblva_hip.so
  LVA_LIB_PATH=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_$v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cross-check > $out/$v.log 2>&1 || echo "$v failed"
  cat $out/t_$v/*/*kernel_stats.csv > $out/$v.csv 2>/dev/null; rm -rf $out/t_$v
  python3 - <<PY
import csv, json
rows = {r['Name'][:34]: float(r['AverageNs'])/1e6 for r in csv.DictReader(open("$out/$v.csv")) if 'lva_step_lazy' in r['Name']}
j = [l for l in open("$out/$v.log") if l.startswith('{')]
rs = json.loads(j[-1])["value"] if j else -1
print("%-14s reads/s %6.2f  " % ("$v", rs) + "  ".join("%s %.3f" % (k[-10:], v) for k, v in sorted(rows.items())))
PY
done
