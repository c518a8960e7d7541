#!/bin/bash
bash scripts/run_variants.sh gpurun_out/r5v7 "--list-size 64 --slots 8 --steps 1 --warmup 0 --pool 8 --cross-check-reads 1" default out8 out8w3
bash scripts/run_variants.sh gpurun_out/r5v7b "--mem-conv 8 --rate 3 --msg-len 164 --list-size 64 --slots 32 --steps 1 --warmup 0 --pool 64 --cross-check-reads 2" default out8
