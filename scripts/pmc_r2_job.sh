#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the lazy-message kernels, separate --pmc passes, on a SMALL run of the benchmark shape
# (8 slots, 16 reads: the 32- and 64-slot passes did not come back from rocprofv3 within 700 s with four kernels per launch).
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
B="--steps 1 --warmup 0 --slots 8 --reads-per-step 16 --pool 16 --no-cpu-baseline --no-launch-events"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/r2_pmc_$c
  s=$(date +%s)
  timeout 240 rocprofv3 --pmc $c --output-format csv -d gpurun_out/r2_pmc_$c -- python3 bench.py $B > gpurun_out/r2_pmc_$c.log 2>&1; echo "$c rc=$? $(( $(date +%s)-s )) s"
done
python3 scripts/pmc_summary.py gpurun_out/r2_pmc_FETCH_SIZE gpurun_out/r2_pmc_WRITE_SIZE > gpurun_out/r2_default_pmc_summary.txt 2>&1
grep '^{' gpurun_out/r2_pmc_FETCH_SIZE.log | tail -1 > gpurun_out/r2_default_bench_under_pmc.json
cut -c1-150 gpurun_out/r2_default_pmc_summary.txt
