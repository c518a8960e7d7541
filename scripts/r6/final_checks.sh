#!/bin/bash
# round 6, last GPU call: the new L = 1 tile-order test, a soak of the final library beyond the driver-run suite (random
# configurations against the CPU oracle), and the driver's command without the profiler (traffic from the counters of this build).
out=gpurun_out/r6final; mkdir -p $out
export LVA_TESTING=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "xcd_aware" 2>&1 | tail -3 > $out/new_test.log; cat $out/new_test.log
timeout 500 python scripts/fuzz_gpu_vs_oracle.py 5602 140 > $out/fuzz_small.log 2>&1; tail -1 $out/fuzz_small.log
timeout 300 python scripts/fuzz_m11.py 5603 8 > $out/fuzz_m11.log 2>&1; tail -1 $out/fuzz_m11.log
grep -c MISMATCH $out/fuzz_small.log $out/fuzz_m11.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/r6f_driver_bench.json 2> $out/driver.err || tail -3 $out/driver.err
python3 - <<PY
import json
j=json.loads(open("$out/r6f_driver_bench.json").read().strip().splitlines()[-1])
print("driver: reads/s %.2f  launch %.3f ms  frac %.3f  frac_moved %.3f traffic %s" % (j["value"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"], j["roofline"]["frac_moved"], j["roofline"]["traffic"]))
for e in j.get("extra_configs", []): print("  %s reads/s %.2f launch %.3f frac %.3f traffic %s" % (e["workload"][:10], e["reads_s"], e["avg_launch_ms"], e["frac"], e.get("traffic")))
print("cpu:", j.get("cpu_baseline"))
PY
