#!/bin/bash
# Re-collect the round-2 evidence that depends on the final kernels / defaults (64 slots at m >= 11, lva_step_acs,
# lva_prepare_step): default bench under --kernel-trace --stats, FETCH/WRITE_SIZE passes, m=6 and m=8 stats.
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
run() { name=$1; shift; rm -rf gpurun_out/r2_$name; timeout 1200 rocprofv3 "$@" > gpurun_out/r2_$name.log 2>&1 || echo "$name failed"; }
stats() { name=$1; shift; run $name --kernel-trace --stats --output-format csv -d gpurun_out/r2_$name -- python3 bench.py "$@";
          cat gpurun_out/r2_$name/*/*kernel_stats.csv > gpurun_out/r2_${name}_kernel_stats.csv; grep '^{' gpurun_out/r2_$name.log | tail -1 > gpurun_out/r2_${name}_bench_under_trace.json;
          cut -c1-160 gpurun_out/r2_${name}_kernel_stats.csv | head -6; }
stats default
bash scripts/pmc_r2.sh
stats m8 --mem-conv 8 --rate 3 --msg-len 164 --list-size 8 --reads-per-step 1024 --pool 1024 --steps 2 --warmup 1 --resident --no-cpu-baseline
stats m6 --mem-conv 6 --rate 1 --msg-len 180 --list-size 1 --reads-per-step 4096 --pool 4096 --steps 2 --warmup 1 --resident --no-cpu-baseline
stats m14 --mem-conv 14 --rate 7 --list-size 8 --slots 8 --reads-per-step 8 --pool 8 --steps 1 --warmup 1 --resident --no-cpu-baseline
