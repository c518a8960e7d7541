#!/bin/bash
# round 3, GPU batch A: lazy-mode overflow fallback tests; deferred tie detection (variant) parity + bench; m=14 lazy variants
out=gpurun_out/r3a; mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_lazy.py -m gpu -x -q > $out/tests_default.log 2>&1
tail -3 $out/tests_default.log
LVA_LIB_PATH=$PWD/variants/defer.so python -m pytest tests/test_gpu_lazy.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz_m11.py -m gpu -x -q -k "not m14 and not L64" > $out/tests_defer.log 2>&1
tail -3 $out/tests_defer.log
bash scripts/run_variants.sh $out "--steps 6 --warmup 2" default defer
echo "--- m14: kernel 2, kernel 4 (2 entries in flight), kernel 4 (1 entry in flight)"
bash scripts/run_variants.sh $out/m14_k2 "--mem-conv 14 --rate 7 --slots 8 --steps 2 --warmup 1 --kernel 2 --pool 32" default
bash scripts/run_variants.sh $out/m14_k4 "--mem-conv 14 --rate 7 --slots 8 --steps 2 --warmup 1 --kernel 4 --pool 32" default gb1
