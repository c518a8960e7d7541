#!/bin/bash
# round 3: L = 1 kernel, paired/hoisted body against the sequential body (both on the position record), rate 1/2 and rate 5/6
out=gpurun_out/r3y; mkdir -p $out
bash scripts/run_variants.sh $out/m6 "--mem-conv 6 --rate 1 --list-size 1 --steps 3 --warmup 1 --pool 4096 --no-cross-check" default nopair default nopair
bash scripts/run_variants.sh $out/m11r1 "--mem-conv 11 --rate 1 --list-size 1 --steps 2 --warmup 1 --no-cross-check" default nopair
bash scripts/run_variants.sh $out/m11 "--list-size 1 --steps 2 --warmup 1 --no-cross-check" default nopair
