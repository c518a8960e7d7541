#!/bin/bash
# round 3: wavefront priority, second batch (3 = output phase at priority 3, 4 = staging 2 / merge 0 / output 1, 5 = odd instance only, 6 = anchor only)
out=gpurun_out/r3v; mkdir -p $out
bash scripts/run_variants.sh $out "--steps 6 --warmup 2 --no-cross-check" default prio1 prio3 prio4 prio5 prio6 default prio1 prio3 prio4 prio5 prio6
bash scripts/run_variants.sh $out/m14 "--mem-conv 14 --rate 7 --slots 8 --steps 2 --warmup 1 --pool 32 --no-cross-check" default prio3 prio4 prio5 prio6
