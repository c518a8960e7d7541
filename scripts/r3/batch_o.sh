#!/bin/bash
# round 3, GPU batch O: odd instance, confirmations with both messages requested together behind uniform branches
out=gpurun_out/r3o; mkdir -p $out
LVA_LIB_PATH=$PWD/variants/vpair.so python -m pytest tests/test_gpu_lazy.py tests/test_gpu_fuzz_m11.py tests/test_gpu_parity.py -m gpu -x -q > $out/tests_vpair.log 2>&1
echo "vpair: $(tail -1 $out/tests_vpair.log)"
bash scripts/run_variants.sh $out "--steps 6 --warmup 2" default vpair
bash scripts/run_variants.sh $out/m8 "--mem-conv 8 --rate 3 --msg-len 164 --steps 3 --warmup 1 --pool 1024 --no-cross-check" default vpair
