#!/bin/bash
# round 5 soak on the final library (records of 16 / 24 / 32 bytes by position): random configurations against the CPU oracle
# beyond the driver-run suite; the big-list draw is the large one
out=gpurun_out/r5soak; mkdir -p $out
timeout 900 python scripts/fuzz_big.py 5301 70 > $out/fuzz_big.log 2>&1; tail -1 $out/fuzz_big.log
timeout 400 python scripts/fuzz_gpu_vs_oracle.py 5302 100 > $out/fuzz_small.log 2>&1; tail -1 $out/fuzz_small.log
timeout 300 python scripts/fuzz_m11.py 5303 8 > $out/fuzz_m11.log 2>&1; tail -1 $out/fuzz_m11.log
grep -c MISMATCH $out/*.log
