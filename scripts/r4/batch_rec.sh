#!/bin/bash
# round 4: record layout of the big-list kernel against the plane layout (variants/norec.so) at list sizes 20 .. 64
out=gpurun_out/r4rec; mkdir -p $out
for L in 20 32 48 64; do
  bash scripts/run_variants.sh $out/L$L "--list-size $L --slots 16 --steps 1 --warmup 1 --pool 32 --no-cross-check" default norec
done
bash scripts/run_variants.sh $out/m8L64 "--mem-conv 8 --rate 3 --msg-len 164 --list-size 64 --slots 32 --steps 1 --warmup 1 --pool 64 --no-cross-check" default norec
