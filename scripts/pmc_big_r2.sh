#!/bin/bash
# memory-path counters of lva_step_big<64,3> (m=11 r=5/6 L=64, 8 slots, one pass of 8 reads) for a library variant:
#   bash scripts/pmc_big_r2.sh TAG [variant]      -> gpurun_out/r2_bigpmc_TAG.txt
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
TAG=$1
if [ -n "$2" ] && [ "$2" != default ]; then export LVA_LIB_PATH=$PWD/variants/$2.so; fi
B="--list-size 64 --steps 1 --warmup 0 --slots 8 --reads-per-step 8 --pool 8 --no-cpu-baseline --no-launch-events --resident"
dirs=""
run() { name=$1; shift; rm -rf gpurun_out/bp_$name; timeout 200 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/bp_$name -- python3 bench.py $B > gpurun_out/bp_$name.log 2>&1 || echo "$name failed"; dirs="$dirs gpurun_out/bp_$name"; }
run a FETCH_SIZE
run b WRITE_SIZE
run c TCC_HIT_sum TCC_MISS_sum
run d TCC_REQ_sum TCC_READ_sum
run e TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum
run f TA_TA_BUSY_sum GRBM_GUI_ACTIVE
run g SQ_INSTS_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES
python3 scripts/pmc_summary.py $dirs 2>&1 | grep -A4 "lva_step_big" | grep -v "^--" > gpurun_out/r2_bigpmc_$TAG.txt
cat gpurun_out/r2_bigpmc_$TAG.txt | cut -c1-120
