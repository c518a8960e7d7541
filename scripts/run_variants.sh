#!/bin/bash
# usage: scripts/run_variants.sh OUTDIR "bench flags" variant1 variant2 ...   (variant "default" = the in-tree library)
# Runs bench.py once per library variant (variants/NAME.so via LVA_LIB_PATH) and collects the JSON lines.
out=$1; flags=$2; shift 2
mkdir -p $out
for v in "$@"; do
  if [ "$v" = default ]; then unset LVA_LIB_PATH; else export LVA_LIB_PATH=$PWD/variants/$v.so; fi
  python bench.py $flags --no-cpu-baseline > $out/$v.json 2> $out/$v.err
  python - "$out/$v.json" "$v" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = j["roofline"]
    print("%-10s value %8.3f reads/s  kernel %.3f ms/launch  pair %.3f  alg %.0f GB/s (%.3f)  e2e %.0f GB/s  fix %s" % (
        sys.argv[2], j["value"], r["avg_launch_ms"], (r["pair"] or {}).get("avg_launch_ms", 0), r["achieved"], r["frac"],
        r["end_to_end_achieved"], j["config"]["fixup_reason"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
