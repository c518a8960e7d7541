#!/bin/bash
# round 3, GPU batch L: big-list kernel with one match slot per entry + 4 entries in flight (120 registers, 33 KB LDS: 4 workgroups per CU);
# lazy anchor with one entry in flight
out=gpurun_out/r3l; mkdir -p $out
LVA_LIB_PATH=$PWD/variants/big1r4.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_random.py -m gpu -x -q -k "big or long_and_odd or L64 or L16 or hundreds or random" > $out/tests_big1r4.log 2>&1
echo "big1r4: $(tail -1 $out/tests_big1r4.log)"
bash scripts/run_variants.sh $out/L64 "--list-size 64 --slots 8 --steps 1 --warmup 1 --pool 16 --no-cross-check" default big1r4
bash scripts/run_variants.sh $out/m8L64 "--mem-conv 8 --rate 3 --msg-len 164 --list-size 64 --slots 32 --steps 1 --warmup 1 --pool 64 --no-cross-check" default big1r4
bash scripts/run_variants.sh $out/m8L16 "--mem-conv 8 --rate 3 --msg-len 164 --list-size 16 --slots 64 --steps 1 --warmup 1 --pool 128 --no-cross-check" default big1r4
LVA_LIB_PATH=$PWD/variants/gb1.so python -m pytest tests/test_gpu_lazy.py -m gpu -x -q > $out/tests_gb1.log 2>&1
echo "gb1: $(tail -1 $out/tests_gb1.log)"
bash scripts/run_variants.sh $out "--steps 6 --warmup 2 --no-cross-check" default gb1
