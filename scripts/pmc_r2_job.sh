export TMPDIR=/tmp
B="--steps 1 --warmup 0 --slots 32 --reads-per-step 128 --no-cpu-baseline --no-launch-events"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/r2_pmc_$c
  s=$(date +%s)
  timeout 700 rocprofv3 --pmc $c --output-format csv -d gpurun_out/r2_pmc_$c -- python3 bench.py $B > gpurun_out/r2_pmc_$c.log 2>&1; echo "$c rc=$? $(( $(date +%s)-s )) s"
done
python3 scripts/pmc_summary.py gpurun_out/r2_pmc_FETCH_SIZE gpurun_out/r2_pmc_WRITE_SIZE > gpurun_out/r2_default_pmc_summary.txt 2>&1
grep '^{' gpurun_out/r2_pmc_FETCH_SIZE.log | tail -1 > gpurun_out/r2_default_bench_under_pmc.json
cat gpurun_out/r2_default_pmc_summary.txt | cut -c1-150
