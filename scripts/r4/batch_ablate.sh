#!/bin/bash
# round 4: where the two lva_step_lazy instances spend their time now that a launch runs one of them (timing builds made by
# scripts/r4/variant_from_patch.sh: results NOT exact) -- kernel averages from rocprofv3 traces
for v in default abl_noverify abl_nooutput abl_cap8 abl_nomerge; do
  lib=default; [ $v != default ] && lib=variants/$v.so
  bash scripts/r4/trace.sh $v $lib > /dev/null 2>&1
  python3 - gpurun_out/r4/${v}_kernel_stats.csv $v <<'PY'
import csv,sys
rows={r["Name"][:40]:float(r["AverageNs"])/1e6 for r in csv.DictReader(open(sys.argv[1])) if "lva_step_lazy<" in r["Name"] or "fixup_lazy" in r["Name"]}
print("%-14s" % sys.argv[2], "  ".join("%s %.3f ms" % (k.split("lva::")[-1][:26], v) for k, v in sorted(rows.items())))
PY
done
