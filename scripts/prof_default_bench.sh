# rocprofv3 evidence for the DEFAULT bench command (python3 bench.py: 32 slots, 3 steps + 1 warm-up, cpu baseline leg included):
# kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes.   bash scripts/prof_default_bench.sh <tag>
export TMPDIR=/tmp
TAG=${1:-r1_default}
run() { name=$1; shift; rm -rf gpurun_out/${TAG}_$name; timeout 400 rocprofv3 "$@" --output-format csv -d gpurun_out/${TAG}_$name -- python3 bench.py > gpurun_out/${TAG}_$name.log 2>&1 || echo "$name failed"; }
run trace --kernel-trace --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
cat gpurun_out/${TAG}_trace/*/*kernel_stats.csv > gpurun_out/${TAG}_kernel_stats.csv
tail -1 gpurun_out/${TAG}_trace.log > gpurun_out/${TAG}_bench_under_trace.json
python3 scripts/pmc_summary.py gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write > gpurun_out/${TAG}_pmc_summary.txt 2>&1
cut -c1-180 gpurun_out/${TAG}_kernel_stats.csv | head -6; grep -A1 "step_fast" gpurun_out/${TAG}_pmc_summary.txt; cut -c1-300 gpurun_out/${TAG}_bench_under_trace.json
