#!/bin/bash
# round 3 profile evidence on the FINAL library (run from the repo root on the GPU box; copy gpurun_out/r3prof/* summaries into profiles/):
#   1. rocprofv3 --kernel-trace --stats of the driver's bench command
#   2. FETCH_SIZE / WRITE_SIZE (separate --pmc passes) and SQ / TA counters at the DEFAULT 64 slots, one step of 64 reads,
#      restricted to the two lva_step_lazy instances so that the passes come back
export TMPDIR=/tmp
out=gpurun_out/r3prof; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/trace.log 2>&1 || echo "trace failed"
cat $out/trace/*/*kernel_stats.csv > $out/r3_default_kernel_stats.csv 2>/dev/null
grep '^{' $out/trace.log | tail -1 > $out/r3_default_bench_under_trace.json
B="python3 bench.py --steps 1 --warmup 0 --reads-per-step 64 --pool 64 --no-cpu-baseline --no-launch-events --no-cross-check"
run() { name=$1; shift; s=$(date +%s); timeout 600 rocprofv3 --kernel-include-regex "lva_step_lazy" "$@" --output-format csv -d $out/$name -- $B > $out/$name.log 2>&1; echo "$name rc=$? $(( $(date +%s)-s )) s"; }
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run ta --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
python3 scripts/pmc_summary.py $out/fetch $out/write $out/sq1 $out/sq2 $out/ta > $out/r3_lazy64_pmc_summary.txt 2>&1
grep '^{' $out/fetch.log | tail -1 > $out/r3_lazy64_bench_under_pmc.json
cut -c1-160 $out/r3_lazy64_pmc_summary.txt | grep -v "^$" | head -80
