#!/bin/bash
# round 5 profile evidence on the FINAL library (GPU box, repo root; copy gpurun_out/r5prof/r5_* into profiles/):
#   1. rocprofv3 --kernel-trace --stats of the driver's bench command
#   2. FETCH_SIZE / WRITE_SIZE (separate --pmc passes) and SQ / TA counters at the DEFAULT 64 slots, one step of 64 reads,
#      restricted to the lva_step_lazy instances
#   3. kernel traces of the other configurations
export TMPDIR=/tmp
out=gpurun_out/r5prof; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/trace.log 2>&1 || echo "trace failed"
cat $out/trace/*/*kernel_stats.csv > $out/r5_default_kernel_stats.csv 2>/dev/null
grep '^{' $out/trace.log | tail -1 > $out/r5_default_bench_under_trace.json
rm -rf $out/trace
B="python3 bench.py --steps 1 --warmup 0 --reads-per-step 128 --pool 128 --no-cpu-baseline --no-launch-events --no-cross-check"
run() { name=$1; shift; s=$(date +%s); timeout 600 rocprofv3 --kernel-include-regex "lva_step_lazy" "$@" --output-format csv -d $out/$name -- $B > $out/$name.log 2>&1; echo "$name rc=$? $(( $(date +%s)-s )) s"; }
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run ta --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
python3 scripts/pmc_summary.py $out/fetch $out/write $out/sq1 $out/sq2 $out/ta > $out/r5_lazy128_pmc_summary.txt 2>&1
grep '^{' $out/fetch.log | tail -1 > $out/r5_lazy128_bench_under_pmc.json
python3 scripts/r5/make_traffic_json.py $out $out/r5_traffic.json
tr() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_$name -- python3 bench.py "$@" --no-cpu-baseline --no-cross-check > $out/t_$name.log 2>&1; cat $out/t_$name/*/*kernel_stats.csv > $out/r5_${name}_kernel_stats.csv; grep '^{' $out/t_$name.log | tail -1 > $out/r5_${name}_bench_under_trace.json; rm -rf $out/t_$name; head -3 $out/r5_${name}_kernel_stats.csv | cut -c1-70,160-230; }
tr m14 --mem-conv 14 --rate 7 --slots 8 --steps 2 --warmup 1 --pool 32
tr big64 --list-size 64 --slots 8 --steps 1 --warmup 1 --pool 16
tr m8 --mem-conv 8 --rate 3 --msg-len 164 --steps 3 --warmup 1 --pool 1024
tr m8L64 --mem-conv 8 --rate 3 --msg-len 164 --list-size 64 --slots 32 --steps 1 --warmup 1 --pool 64
tr m6 --mem-conv 6 --rate 1 --list-size 1 --steps 3 --warmup 1 --pool 8192
tr m11L1 --list-size 1 --steps 2 --warmup 1 --pool 512
# the same configurations without the profiler, and the rate table of DESIGN_HISTORY 4e (one-bit steps: all / half / a third)
nb() { name=$1; shift; python3 bench.py "$@" --no-cpu-baseline --no-cross-check > $out/r5_${name}_bench.json 2> $out/nb_$name.err; cut -c1-120 $out/r5_${name}_bench.json; }
nb m14 --mem-conv 14 --rate 7 --slots 8 --steps 2 --warmup 1 --pool 32
nb L64 --list-size 64 --slots 8 --steps 1 --warmup 1 --pool 16
nb m8 --mem-conv 8 --rate 3 --msg-len 164 --steps 3 --warmup 1 --pool 1024
nb m8L64 --mem-conv 8 --rate 3 --msg-len 164 --list-size 64 --slots 32 --steps 1 --warmup 1 --pool 64
nb m6 --mem-conv 6 --rate 1 --list-size 1 --steps 3 --warmup 1 --pool 8192
nb m11L1 --list-size 1 --steps 2 --warmup 1 --pool 512
for r in 1 2 5; do nb rate$r --mem-conv 11 --rate $r --steps 2 --warmup 1 --pool 128; done
# the big-list kernel's counters on the same (final) library -> gpurun_out/r5pmc/r5_big64_pmc_summary.txt, r5_big64_bench_under_pmc.json
bash scripts/r5/pmc.sh big64 "lva_step_big_rec" --list-size 64 --slots 8 --pool 8
cp gpurun_out/r5pmc/r5_big64_pmc_summary.txt gpurun_out/r5pmc/r5_big64_bench_under_pmc.json $out/ 2>/dev/null
