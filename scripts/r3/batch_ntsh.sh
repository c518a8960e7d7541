#!/bin/bash
# round 3: (score, fingerprint) stores of the merge loop non-temporal
out=gpurun_out/r3ntsh; mkdir -p $out
LVA_LIB_PATH=$PWD/variants/ntsh.so timeout 600 python -m pytest tests/test_gpu_lazy.py -m gpu -x -q > $out/tests.log 2>&1; echo "ntsh: $(tail -1 $out/tests.log)"
bash scripts/run_variants.sh $out "--steps 6 --warmup 2 --no-cross-check" default ntsh default ntsh
