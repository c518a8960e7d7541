#!/bin/bash
# Round-2 rocprofv3 evidence.  bash scripts/prof_r2.sh   (on the GPU box; writes gpurun_out/r2_*)
#   default : the DEFAULT bench command (python3 bench.py) under --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE
#             in separate --pmc passes (same command without the CPU legs)
#   big64   : m=11 r=5/6 L=64 (configs[4]), 8 slots
#   m14     : m=14 r=7/8 L=8 (configs[3]), 8 slots
#   m8, m6  : the small trellises with their default slot counts (256 / 1024)
# The program itself follows "--" (no env / bash -c hop).
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
run() { name=$1; shift; rm -rf gpurun_out/r2_$name; timeout 900 rocprofv3 "$@" > gpurun_out/r2_$name.log 2>&1 || echo "$name failed"; }
stats() { name=$1; shift; run $name --kernel-trace --stats --output-format csv -d gpurun_out/r2_$name -- python3 bench.py "$@"; 
          cat gpurun_out/r2_$name/*/*kernel_stats.csv > gpurun_out/r2_${name}_kernel_stats.csv; grep '^{' gpurun_out/r2_$name.log | tail -1 > gpurun_out/r2_${name}_bench_under_trace.json;
          cut -c1-160 gpurun_out/r2_${name}_kernel_stats.csv | head -5; }
stats default
run default_fetch --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2_default_fetch -- python3 bench.py --no-cpu-baseline
run default_write --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2_default_write -- python3 bench.py --no-cpu-baseline
python3 scripts/pmc_summary.py gpurun_out/r2_default_fetch gpurun_out/r2_default_write > gpurun_out/r2_default_pmc_summary.txt 2>&1
grep '^{' gpurun_out/r2_default_fetch.log | tail -1 > gpurun_out/r2_default_bench_under_pmc.json
stats big64 --list-size 64 --slots 8 --reads-per-step 8 --pool 8 --steps 2 --warmup 1 --resident --no-cpu-baseline
stats m14 --mem-conv 14 --rate 7 --list-size 8 --slots 8 --reads-per-step 8 --pool 8 --steps 1 --warmup 1 --resident --no-cpu-baseline
stats m8 --mem-conv 8 --rate 3 --msg-len 164 --list-size 8 --reads-per-step 1024 --pool 1024 --steps 2 --warmup 1 --resident --no-cpu-baseline
stats m6 --mem-conv 6 --rate 1 --msg-len 180 --list-size 1 --reads-per-step 4096 --pool 4096 --steps 2 --warmup 1 --resident --no-cpu-baseline
grep -A2 "step_fast" gpurun_out/r2_default_pmc_summary.txt | head -12
