#!/bin/bash
# round 3: lazy_ctx on the position record
out=gpurun_out/r3t; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_fuzz_m11.py -m gpu -x -q > $out/tests.log 2>&1
echo "posrec in lazy_ctx: $(tail -1 $out/tests.log)"
bash scripts/run_variants.sh $out "--steps 6 --warmup 2 --no-cross-check" default noposrec default noposrec
bash scripts/run_variants.sh $out/m14 "--mem-conv 14 --rate 7 --slots 8 --steps 2 --warmup 1 --pool 32 --no-cross-check" default noposrec
