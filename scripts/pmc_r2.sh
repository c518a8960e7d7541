#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the dominant kernel on the benchmark shape, separate --pmc passes (MI355X_MICROARCH.md, HBM):
# one step of 128 reads through 32 slots (2035 launches), no per-launch events.   bash scripts/pmc_r2.sh [extra bench flags]
# (32 slots on purpose: with the default 64 slots the WRITE_SIZE pass did not come back from rocprofv3 within 15 minutes,
#  twice; the FETCH_SIZE pass did: 5.20 GB raw per launch = twice the 32-slot figure, as expected)
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
B="--steps 1 --warmup 0 --slots 32 --reads-per-step 128 --no-cpu-baseline --no-launch-events $*"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/r2_pmc_$c
  timeout 900 rocprofv3 --pmc $c --output-format csv -d gpurun_out/r2_pmc_$c -- python3 bench.py $B > gpurun_out/r2_pmc_$c.log 2>&1 || echo "$c failed"
done
python3 scripts/pmc_summary.py gpurun_out/r2_pmc_FETCH_SIZE gpurun_out/r2_pmc_WRITE_SIZE > gpurun_out/r2_default_pmc_summary.txt 2>&1
grep '^{' gpurun_out/r2_pmc_FETCH_SIZE.log | tail -1 > gpurun_out/r2_default_bench_under_pmc.json
grep -B1 -A3 "lva_step_fast" gpurun_out/r2_default_pmc_summary.txt | head -20
