# rocprofv3 evidence for the dominant kernel (run from the repo root on the GPU box):
#   bash scripts/pmc_r1.sh <tag>      -> gpurun_out/<tag>_*  (copy the summaries into profiles/)
export TMPDIR=/tmp
TAG=${1:-r1}
B="python3 bench.py --steps 1 --warmup 0 --slots 8 --reads-per-step 8 --pool 8 --no-cpu-baseline --no-launch-events"
run() { name=$1; shift; timeout 180 rocprofv3 "$@" --output-format csv -d gpurun_out/${TAG}_$name -- $B > gpurun_out/${TAG}_$name.log 2>&1 || echo "$name failed"; }
run trace --kernel-trace --stats
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch --pmc FETCH_SIZE TCC_HIT_sum
run write --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
run ta --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE
python3 scripts/pmc_summary.py gpurun_out/${TAG}_sq1 gpurun_out/${TAG}_sq2 gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_ta > gpurun_out/${TAG}_pmc_summary.txt 2>&1
cat gpurun_out/${TAG}_trace/*/*kernel_stats.csv > gpurun_out/${TAG}_kernel_stats.csv
grep -o '"value": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"achieved": [0-9.]*' gpurun_out/${TAG}_trace.log > gpurun_out/${TAG}_bench_under_trace.txt
