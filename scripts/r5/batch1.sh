#!/bin/bash
bash scripts/run_variants.sh gpurun_out/r5v1 "--list-size 64 --slots 8 --steps 1 --warmup 0 --pool 8 --no-cross-check" default condst nostore nomsg
bash scripts/r5/pmc.sh big64a "lva_step_list" --list-size 64 --slots 8 --pool 8
