#!/bin/bash
# round 3: the lazy pair on codes with other shares of one-bit steps (rate 1/2: all of them; 2/3; 5/6: a third)
out=gpurun_out/r3rates; mkdir -p $out
for r in 1 2 5; do
  bash scripts/run_variants.sh $out/r$r "--mem-conv 11 --rate $r --steps 2 --warmup 1 --pool 128 --no-cross-check" default
done
bash scripts/run_variants.sh $out/m8r1 "--mem-conv 8 --rate 1 --msg-len 100 --steps 2 --warmup 1 --pool 512 --no-cross-check" default
bash scripts/run_variants.sh $out/m8r5 "--mem-conv 8 --rate 5 --msg-len 100 --steps 2 --warmup 1 --pool 512 --no-cross-check" default
