#!/bin/bash
out=gpurun_out/r5cmp2; mkdir -p $out
for L in 32 48 64; do
  bash scripts/run_variants.sh $out/L$L "--list-size $L --slots 8 --steps 1 --warmup 0 --pool 8 --cross-check-reads 1" default norec
done
bash scripts/run_variants.sh $out/m8L64 "--mem-conv 8 --rate 3 --msg-len 164 --list-size 64 --slots 32 --steps 1 --warmup 0 --pool 64 --cross-check-reads 2" default norec
