#!/bin/bash
# round 3, GPU batch M: backend scheduling strategies (-mllvm -amdgpu-sched-strategy=...)
out=gpurun_out/r3m; mkdir -p $out
for v in s_ilp s_mem s_minreg; do
  LVA_LIB_PATH=$PWD/variants/$v.so python -m pytest tests/test_gpu_lazy.py -m gpu -x -q -k "golden or tie or turnover" > $out/tests_$v.log 2>&1
  echo "$v: $(tail -1 $out/tests_$v.log)"
done
bash scripts/run_variants.sh $out "--steps 6 --warmup 2 --no-cross-check" default s_ilp s_mem s_minreg
