#!/bin/bash
# round 6 profile evidence on the FINAL library (GPU box, repo root; the r6f_* files are copied into profiles/, r6_traffic.json replaced).
# ONE set, after the last kernel change:
#   1. rocprofv3 --kernel-trace --stats of the driver's bench command (configs[1] + the extra_configs legs: configs[3], configs[4])
#   2. FETCH_SIZE / WRITE_SIZE (separate --pmc passes), SQ and TA / TCC counters at the DEFAULT 128 slots, one step of 128 reads,
#      restricted to the lva_step_lazy instances; the same five passes for lva_step_big_rec<64> (configs[4] shape, 8 slots)
#   3. profiles/r6_traffic.json (line traffic of both dominant kernels, named after the build it was measured on)
export TMPDIR=/tmp
out=gpurun_out/r6prof; mkdir -p $out
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/trace.log 2>&1 || echo "trace failed"
cat $out/trace/*/*kernel_stats.csv > $out/r6f_default_kernel_stats.csv 2>/dev/null
grep '^{' $out/trace.log | tail -1 > $out/r6f_default_bench_under_trace.json
rm -rf $out/trace
head -6 $out/r6f_default_kernel_stats.csv | cut -c1-60,150-230
python3 - <<PY
import json
j=json.loads(open("$out/r6f_default_bench_under_trace.json").read().strip().splitlines()[-1])
print("traced driver command: reads/s %.2f  launch %.3f ms  frac %.3f  build %s" % (j["value"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"], j["library"]["build_id"]))
for e in j.get("extra_configs", []): print("  %s reads/s %.2f launch %.3f frac %.3f" % (e["workload"][:10], e["reads_s"], e["avg_launch_ms"], e["frac"]))
print("cpu:", j.get("cpu_baseline"))
PY
B="python3 bench.py --steps 1 --warmup 0 --reads-per-step 128 --pool 128 --no-cpu-baseline --no-launch-events --no-cross-check --no-extra-configs"
run() { name=$1; shift; s=$(date +%s); timeout 600 rocprofv3 --kernel-include-regex "lva_step_lazy" "$@" --output-format csv -d $out/$name -- $B > $out/$name.log 2>&1; echo "$name rc=$? $(( $(date +%s)-s )) s"; }
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run ta --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
python3 scripts/pmc_summary.py $out/fetch $out/write $out/sq1 $out/sq2 $out/ta > $out/r6f_lazy128_pmc_summary.txt 2>&1
grep '^{' $out/fetch.log | tail -1 > $out/r6f_lazy128_bench_under_pmc.json
cp $out/r6f_lazy128_bench_under_pmc.json $out/r5_lazy128_bench_under_pmc.json      # (the name scripts/r5/make_traffic_json.py reads)
python3 scripts/r5/make_traffic_json.py $out $out/r6f_lazy_traffic.json > $out/lazy_traffic.log 2>&1 || tail -3 $out/lazy_traffic.log
BB="python3 bench.py --list-size 64 --slots 8 --pool 8 --steps 1 --warmup 0 --no-cpu-baseline --no-launch-events --no-cross-check --no-extra-configs"
runb() { name=$1; shift; s=$(date +%s); timeout 600 rocprofv3 --kernel-include-regex "lva_step_big_rec" "$@" --output-format csv -d $out/big_$name -- $BB > $out/big_$name.log 2>&1; echo "big $name rc=$? $(( $(date +%s)-s )) s"; }
runb fetch --pmc FETCH_SIZE
runb write --pmc WRITE_SIZE
runb sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
runb sq2 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
runb ta --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
python3 scripts/pmc_summary.py $out/big_fetch $out/big_write $out/big_sq1 $out/big_sq2 $out/big_ta > $out/r6f_big64_pmc_summary.txt 2>&1
grep '^{' $out/big_fetch.log | tail -1 > $out/r6f_big64_bench_under_pmc.json
python3 scripts/r6/make_traffic_json.py $out $out/r6_traffic.json r6f
rm -rf $out/fetch $out/write $out/sq1 $out/sq2 $out/ta $out/big_fetch $out/big_write $out/big_sq1 $out/big_sq2 $out/big_ta
ls -la $out | head -30
