#!/bin/bash
# round 4 soak on the final library: random configurations against the CPU oracle beyond the driver-run suite
out=gpurun_out/r4soak; mkdir -p $out
timeout 700 python scripts/fuzz_big.py 4104 40 > $out/fuzz_big.log 2>&1; tail -1 $out/fuzz_big.log
timeout 500 python scripts/fuzz_gpu_vs_oracle.py 4105 80 > $out/fuzz_small.log 2>&1; tail -1 $out/fuzz_small.log
timeout 500 python scripts/fuzz_m11.py 4106 8 > $out/fuzz_m11.log 2>&1; tail -1 $out/fuzz_m11.log
grep -c MISMATCH $out/*.log
