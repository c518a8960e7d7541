#!/bin/bash
# round 4, final library (compact lists, band table clipped at t + 2, record layout): random configurations against the CPU oracle
out=gpurun_out/r4soak2; mkdir -p $out
timeout 420 python scripts/fuzz_gpu_vs_oracle.py 4207 140 > $out/fuzz_small.log 2>&1; tail -1 $out/fuzz_small.log
timeout 300 python scripts/fuzz_m11.py 4208 10 > $out/fuzz_m11.log 2>&1; tail -1 $out/fuzz_m11.log
timeout 300 python scripts/fuzz_big.py 4209 24 > $out/fuzz_big.log 2>&1; tail -1 $out/fuzz_big.log
grep -c MISMATCH $out/*.log
