#!/bin/bash
# round 6 profile evidence on the FINAL library (GPU box, repo root; copy gpurun_out/r6prof/r6_* into profiles/).  ONE set:
#   1. rocprofv3 --kernel-trace --stats of the driver's bench command (configs[1] + the extra_configs legs: configs[3], configs[4])
#   2. the same command without the profiler (the line the driver's BENCH_r06 should reproduce)
# Counters: the final library is byte-identical to round 5's (build bcc33b9ee275), whose PMC passes are committed
# (profiles/r5_lazy128_pmc_summary.txt, r5_big64_pmc_summary.txt); scripts/r6/make_traffic_json.py derives profiles/r6_traffic.json.
export TMPDIR=/tmp
out=gpurun_out/r6prof; mkdir -p $out
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/trace.log 2>&1 || echo "trace failed"
cat $out/trace/*/*kernel_stats.csv > $out/r6_default_kernel_stats.csv 2>/dev/null
grep '^{' $out/trace.log | tail -1 > $out/r6_default_bench_under_trace.json
rm -rf $out/trace
head -6 $out/r6_default_kernel_stats.csv | cut -c1-60,150-230
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/r6_driver_bench.json 2> $out/driver.err || tail -3 $out/driver.err
python3 - <<PY
import json
j=json.loads(open("$out/r6_driver_bench.json").read().strip().splitlines()[-1])
print("driver: reads/s %.2f  launch %.3f ms  frac %.3f  traffic %s" % (j["value"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"], j["roofline"]["traffic"]))
for e in j.get("extra_configs", []): print("  %s reads/s %.2f launch %.3f frac %.3f traffic %s" % (e["workload"][:10], e["reads_s"], e["avg_launch_ms"], e["frac"], e.get("traffic")))
print("cpu:", j.get("cpu_baseline"))
PY
