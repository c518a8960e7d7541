#!/bin/bash
# round 3 soak of the FINAL library against the CPU oracle (beyond the test suite): small trellises incl. the overflow path, m=11/m=14, big lists
out=gpurun_out/r3soak2; mkdir -p $out
timeout 700 python scripts/fuzz_gpu_vs_oracle.py 41 200 > $out/fuzz_small.log 2>&1; tail -1 $out/fuzz_small.log
LVA_WORK_CAP=16 timeout 400 python scripts/fuzz_gpu_vs_oracle.py 42 80 > $out/fuzz_small_overflow.log 2>&1; tail -1 $out/fuzz_small_overflow.log
timeout 900 python scripts/fuzz_m11.py 43 16 > $out/fuzz_m11.log 2>&1; tail -1 $out/fuzz_m11.log
timeout 500 python scripts/fuzz_big.py 44 16 > $out/fuzz_big.log 2>&1; tail -1 $out/fuzz_big.log
grep -h MISMATCH $out/*.log | head
