#!/bin/bash
# round 3, GPU batch I: confirmation of fingerprint matches as a loop over the entries that have one (vl1: odd instance, vl2: both)
out=gpurun_out/r3i; mkdir -p $out
for v in vl1 vl2; do
  LVA_LIB_PATH=$PWD/variants/$v.so python -m pytest tests/test_gpu_lazy.py tests/test_gpu_fuzz_m11.py -m gpu -x -q > $out/tests_$v.log 2>&1
  echo "$v: $(tail -1 $out/tests_$v.log)"
done
bash scripts/run_variants.sh $out "--steps 6 --warmup 2 --no-cross-check" default vl1 vl2
