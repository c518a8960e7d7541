#!/bin/bash
# usage (on the GPU box, from the repo root): bash scripts/r4/trace.sh NAME [LVA_LIB_PATH] [bench flags...]
# rocprofv3 kernel trace + stats of a short default bench run -> gpurun_out/r4/NAME_kernel_stats.csv, NAME_bench.json
export TMPDIR=/tmp
name=$1; lib=$2; shift 2
out=gpurun_out/r4; mkdir -p $out
if [ -n "$lib" ] && [ "$lib" != default ]; then export LVA_LIB_PATH=$PWD/$lib; else unset LVA_LIB_PATH; fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cross-check "$@" > $out/$name.log 2>&1 || echo "trace $name failed"
cat $out/trace_$name/*/*kernel_stats.csv > $out/${name}_kernel_stats.csv 2>/dev/null
grep '^{' $out/$name.log | tail -1 > $out/${name}_bench.json
rm -rf $out/trace_$name
head -8 $out/${name}_kernel_stats.csv | cut -c1-200
