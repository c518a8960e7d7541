#!/bin/bash
# Re-collect the round-2 evidence that depends on the final merge loop (lva_step_lazy pair, lva_step_fast): default bench,
# m=8 and m=14 under --kernel-trace --stats, then the driver's bench command unprofiled.   bash scripts/prof_r2_final.sh
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
run() { name=$1; shift; rm -rf gpurun_out/r2_$name; timeout 900 rocprofv3 "$@" > gpurun_out/r2_$name.log 2>&1 || echo "$name failed"; }
stats() { name=$1; shift; run $name --kernel-trace --stats --output-format csv -d gpurun_out/r2_$name -- python3 bench.py "$@";
          cat gpurun_out/r2_$name/*/*kernel_stats.csv > gpurun_out/r2_${name}_kernel_stats.csv; grep '^{' gpurun_out/r2_$name.log | tail -1 > gpurun_out/r2_${name}_bench_under_trace.json;
          cut -c1-160 gpurun_out/r2_${name}_kernel_stats.csv | head -5; }
stats default
stats m8 --mem-conv 8 --rate 3 --msg-len 164 --list-size 8 --reads-per-step 1024 --pool 1024 --steps 2 --warmup 1 --resident --no-cpu-baseline
stats m14 --mem-conv 14 --rate 7 --list-size 8 --slots 8 --reads-per-step 8 --pool 8 --steps 1 --warmup 1 --resident --no-cpu-baseline
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_driver_like.json 2> gpurun_out/r2_driver_like.err; cut -c1-300 gpurun_out/r2_driver_like.json
