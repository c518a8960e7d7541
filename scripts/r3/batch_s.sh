#!/bin/bash
# round 3: lazy kernels with the slot record in one load and the target tables requested ahead of the staging wait
out=gpurun_out/r3s; mkdir -p $out
LVA_LIB_PATH=$PWD/variants/hoist.so timeout 900 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_fuzz_m11.py -m gpu -x -q > $out/tests_hoist.log 2>&1
echo "hoist: $(tail -1 $out/tests_hoist.log)"
bash scripts/run_variants.sh $out "--steps 6 --warmup 2 --no-cross-check" default hoist default hoist
bash scripts/run_variants.sh $out/m14 "--mem-conv 14 --rate 7 --slots 8 --steps 2 --warmup 1 --pool 32 --no-cross-check" default hoist
bash scripts/run_variants.sh $out/m8 "--mem-conv 8 --rate 3 --msg-len 164 --steps 3 --warmup 1 --pool 1024 --no-cross-check" default hoist
