#!/bin/bash
export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "Name:[ ]*[A-Za-z0-9_]*" | sed 's/Name:[ ]*//' | sort -u > gpurun_out/r5pmc/avail_counters.txt
wc -l gpurun_out/r5pmc/avail_counters.txt
bash scripts/r5/pmc.sh big64b "lva_step_big_rec" --list-size 64 --slots 8 --pool 8
