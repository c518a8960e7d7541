#!/bin/bash
# round 3, GPU batch E: 32-conv tiles for the lazy pair; big-list merge v2 (queue of rejected candidates, XOR/min de-duplication)
out=gpurun_out/r3e; mkdir -p $out
LVA_LIB_PATH=$PWD/variants/ts32.so python -m pytest tests/test_gpu_lazy.py tests/test_gpu_fuzz_m11.py -m gpu -x -q > $out/tests_ts32.log 2>&1
echo "ts32: $(tail -1 $out/tests_ts32.log)"
bash scripts/run_variants.sh $out "--steps 6 --warmup 2 --no-cross-check" default ts32
for v in bigv2 bigv2nb; do
  LVA_LIB_PATH=$PWD/variants/$v.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_random.py -m gpu -x -q -k "big or long_and_odd or L64 or L16 or hundreds or random" > $out/tests_$v.log 2>&1
  echo "$v: $(tail -1 $out/tests_$v.log)"
done
bash scripts/run_variants.sh $out/L64 "--list-size 64 --slots 8 --steps 1 --warmup 1 --pool 16 --no-cross-check" default bigv2 bigv2nb
bash scripts/run_variants.sh $out/m8L64 "--mem-conv 8 --rate 3 --msg-len 164 --list-size 64 --slots 32 --steps 1 --warmup 1 --pool 64 --no-cross-check" default bigv2
bash scripts/run_variants.sh $out/m8L16 "--mem-conv 8 --rate 3 --msg-len 164 --list-size 16 --slots 64 --steps 1 --warmup 1 --pool 128 --no-cross-check" default bigv2
