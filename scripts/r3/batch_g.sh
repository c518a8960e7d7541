#!/bin/bash
# round 3, GPU batch G: fused anchor+odd launch (VALU-bound and TA-bound workgroups share the CUs); per-lane eq bitmask in the merge loop
out=gpurun_out/r3g; mkdir -p $out
for v in fused8 fused7 eqbits fused8eq; do
  LVA_LIB_PATH=$PWD/variants/$v.so python -m pytest tests/test_gpu_lazy.py tests/test_gpu_fuzz_m11.py -m gpu -x -q > $out/tests_$v.log 2>&1
  echo "$v: $(tail -1 $out/tests_$v.log)"
done
bash scripts/run_variants.sh $out "--steps 6 --warmup 2 --no-cross-check" default fused8 fused7 eqbits fused8eq
bash scripts/run_variants.sh $out/m14 "--mem-conv 14 --rate 7 --slots 8 --steps 2 --warmup 1 --pool 32 --no-cross-check" default fused8 fused7
bash scripts/run_variants.sh $out/m8 "--mem-conv 8 --rate 3 --msg-len 164 --steps 3 --warmup 1 --pool 1024 --no-cross-check" default fused8 fused7
