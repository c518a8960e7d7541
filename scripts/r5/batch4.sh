#!/bin/bash
bash scripts/run_variants.sh gpurun_out/r5v4 "--list-size 64 --slots 8 --steps 1 --warmup 0 --pool 8 --cross-check-reads 1" default stag1 stag2
