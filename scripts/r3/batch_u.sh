#!/bin/bash
# round 3: wavefront priority (s_setprio) around the merge / the output phase, lazy pair and big-list kernel
out=gpurun_out/r3u; mkdir -p $out
bash scripts/run_variants.sh $out "--steps 6 --warmup 2 --no-cross-check" default prio1 prio2 default prio1 prio2
bash scripts/run_variants.sh $out/big "--list-size 64 --slots 8 --steps 1 --warmup 1 --pool 16 --no-cross-check" default bprio1 bprio2
bash scripts/run_variants.sh $out/m14 "--mem-conv 14 --rate 7 --slots 8 --steps 2 --warmup 1 --pool 32 --no-cross-check" default prio1 prio2
