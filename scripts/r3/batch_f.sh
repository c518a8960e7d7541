#!/bin/bash
# round 3, GPU batch F: anchor instance held to 64 registers (four workgroups per CU instead of three; one cold 8-byte spill)
out=gpurun_out/r3f; mkdir -p $out
for v in a8 a8late; do
  LVA_LIB_PATH=$PWD/variants/$v.so python -m pytest tests/test_gpu_lazy.py -m gpu -x -q > $out/tests_$v.log 2>&1
  echo "$v: $(tail -1 $out/tests_$v.log)"
done
bash scripts/run_variants.sh $out "--steps 6 --warmup 2 --no-cross-check" default a8 a8late
bash scripts/run_variants.sh $out/m14 "--mem-conv 14 --rate 7 --slots 8 --steps 2 --warmup 1 --pool 32 --no-cross-check" default a8
bash scripts/run_variants.sh $out/m8 "--mem-conv 8 --rate 3 --msg-len 164 --steps 3 --warmup 1 --pool 1024 --no-cross-check" default a8
