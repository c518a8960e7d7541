#!/bin/bash
# round 3, GPU batch H: timing-only ablations of the lazy pair that keep the data valid: abl6 = the merge once more without
# its stores, abl7 = the output phase once more (same stores again); both decode correctly (parity run included)
out=gpurun_out/r3h2; mkdir -p $out
for v in abl6 abl7; do
  LVA_LIB_PATH=$PWD/variants/$v.so python -m pytest tests/test_gpu_lazy.py -m gpu -x -q -k "golden or tie" > $out/tests_$v.log 2>&1
  echo "$v: $(tail -1 $out/tests_$v.log)"
done
bash scripts/run_variants.sh $out "--steps 4 --warmup 1 --no-cross-check" default abl6 abl7
