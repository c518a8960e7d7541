#!/bin/bash
# round 3: non-temporal loads for what a step reads once (staging; + stay lists)
out=gpurun_out/r3nt; mkdir -p $out
LVA_LIB_PATH=$PWD/variants/nt2.so timeout 600 python -m pytest tests/test_gpu_lazy.py -m gpu -x -q > $out/tests.log 2>&1; echo "nt2: $(tail -1 $out/tests.log)"
bash scripts/run_variants.sh $out "--steps 6 --warmup 2 --no-cross-check" default nt1 nt2 default nt1 nt2
