#!/bin/bash
# round 3, GPU batch D: loads without control flow -- verification pass (vp), anchor output phase (anb), big-list output phase (bignb)
out=gpurun_out/r3d; mkdir -p $out
for v in vp vp_anb2 vp_anb4 anb4; do
  LVA_LIB_PATH=$PWD/variants/$v.so python -m pytest tests/test_gpu_lazy.py tests/test_gpu_fuzz_m11.py -m gpu -x -q > $out/tests_$v.log 2>&1
  echo "$v: $(tail -1 $out/tests_$v.log)"
done
bash scripts/run_variants.sh $out "--steps 6 --warmup 2 --no-cross-check" default vp vp_anb2 vp_anb4 anb4
for v in bignb bignb8; do
  LVA_LIB_PATH=$PWD/variants/$v.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_random.py -m gpu -x -q -k "big or long_and_odd or L64 or L16 or hundreds or random" > $out/tests_$v.log 2>&1
  echo "$v: $(tail -1 $out/tests_$v.log)"
done
bash scripts/run_variants.sh $out/L64 "--list-size 64 --slots 8 --steps 1 --warmup 1 --pool 16 --no-cross-check" default bignb bignb8
bash scripts/run_variants.sh $out/m8L64 "--mem-conv 8 --rate 3 --msg-len 164 --list-size 64 --slots 32 --steps 1 --warmup 1 --pool 64 --no-cross-check" default bignb
