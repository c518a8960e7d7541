#!/bin/bash
# round 3, GPU batch C: full GPU suite on the default library (funnel-shift push + global loads + lazy at m=14 + overflow fallback),
# headline bench, m=14 bench, m=6 L=1 positions-per-workgroup variants
out=gpurun_out/r3c; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/tests_default.log 2>&1
tail -4 $out/tests_default.log
bash scripts/run_variants.sh $out "--steps 6 --warmup 2 --no-cross-check" default
bash scripts/run_variants.sh $out/m14 "--mem-conv 14 --rate 7 --slots 8 --steps 2 --warmup 1 --pool 32 --no-cross-check" default
for v in ppw2 ppw4 ppw8; do
  LVA_LIB_PATH=$PWD/variants/$v.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q -k "L1 or l1 or exact_kernel_matches or fast_kernel_matches or cfg1" > $out/tests_$v.log 2>&1
  echo "$v: $(tail -1 $out/tests_$v.log)"
done
bash scripts/run_variants.sh $out/m6 "--mem-conv 6 --rate 1 --list-size 1 --steps 3 --warmup 1 --pool 4096 --no-cross-check" default ppw2 ppw4 ppw8
bash scripts/run_variants.sh $out/m8l1 "--mem-conv 8 --rate 3 --msg-len 164 --list-size 1 --steps 3 --warmup 1 --pool 1024 --no-cross-check" default ppw4
