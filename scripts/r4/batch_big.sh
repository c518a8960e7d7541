#!/bin/bash
# round 4: cheap experiments on the big-list kernel (L = 64): read slots 4 / 8 / 16, 16-conv tiles, non-temporal message loads
out=gpurun_out/r4big; mkdir -p $out
F="--list-size 64 --steps 1 --warmup 1 --pool 32 --no-cross-check"
bash scripts/run_variants.sh $out/s8 "$F --slots 8" default tsb16 ntmsg
bash scripts/run_variants.sh $out/s4 "$F --slots 4" default
bash scripts/run_variants.sh $out/s16 "$F --slots 16" default tsb16
