#!/bin/bash
# round 3: anchor instance keeps the predecessor word of the source conv state from tile_target
out=gpurun_out/r3z; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_fuzz_m11.py -m gpu -x -q > $out/tests.log 2>&1
echo "pk1 kept: $(tail -1 $out/tests.log)"
bash scripts/run_variants.sh $out "--steps 6 --warmup 2 --no-cross-check" default nopk1 default nopk1
bash scripts/run_variants.sh $out/m14 "--mem-conv 14 --rate 7 --slots 8 --steps 2 --warmup 1 --pool 32 --no-cross-check" default nopk1
