#!/bin/bash
# the driver's bench command under rocprofv3 --kernel-trace --stats -> gpurun_out/r4prof/r4_default_kernel_stats.csv, r4_default_bench_under_trace.json
export TMPDIR=/tmp
out=gpurun_out/r4prof; mkdir -p $out
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/trace.log 2>&1 || echo "trace failed"
cat $out/trace/*/*kernel_stats.csv > $out/r4_default_kernel_stats.csv 2>/dev/null
grep '^{' $out/trace.log | tail -1 > $out/r4_default_bench_under_trace.json
rm -rf $out/trace
head -3 $out/r4_default_kernel_stats.csv | cut -c1-60,150-260
