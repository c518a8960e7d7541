export TMPDIR=/tmp
for s in 1 2 4; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sl$s -- python3 bench.py --steps 1 --warmup 0 --slots $s --reads-per-step 4 --no-cpu-baseline --check 0 > gpurun_out/sl$s.log 2>&1
echo "slots $s"; cat gpurun_out/sl$s/*/*kernel_stats.csv | grep "step_f" | cut -d, -f1-4 | cut -c1-30,100-
done
