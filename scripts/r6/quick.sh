#!/bin/bash
# round 6 experiment loop (GPU box, repo root): lazy parity tests, a kernel trace of a short benchmark run, the same run untraced.
#   scripts/r6/quick.sh <tag> [notest]
export TMPDIR=/tmp
tag=$1; out=gpurun_out/r6/$tag; mkdir -p $out
if [ "$2" != "notest" ]; then
  timeout 900 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_headline.py -m gpu -x -q 2>&1 | tail -5 > $out/tests.log; cat $out/tests.log
fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cross-check > $out/trace.log 2>&1 || echo "trace failed"
cat $out/trace/*/*kernel_stats.csv > $out/kernel_stats.csv 2>/dev/null; rm -rf $out/trace
head -8 $out/kernel_stats.csv | cut -d, -f1-5 | cut -c1-150
python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --cross-check-reads 32 > $out/bench.json 2> $out/bench.err || tail -5 $out/bench.err
python3 - <<PY
import json
j=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print("reads/s %.2f  launch %.3f ms  frac %.3f  fixups %d %s  %s" % (j["value"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"], j["config"]["fixup_states"], j["config"]["fixup_reason"], j["config"].get("cross_check")))
PY
