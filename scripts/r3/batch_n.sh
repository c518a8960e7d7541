#!/bin/bash
out=gpurun_out/r3n; mkdir -p $out
bash scripts/run_variants.sh $out "--steps 6 --warmup 2" default b100 b50 trk relax
