#!/bin/bash
# round 3: kernel traces of the L = 1 configurations on the final library
export TMPDIR=/tmp
out=gpurun_out/r3prof3; mkdir -p $out
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- python3 bench.py "$@" --no-cpu-baseline --no-cross-check > $out/$name.log 2>&1; cat $out/$name/*/*kernel_stats.csv > $out/r3_${name}_kernel_stats.csv; grep '^{' $out/$name.log | tail -1 > $out/r3_${name}_bench_under_trace.json; head -3 $out/r3_${name}_kernel_stats.csv | cut -c1-70,160-230; }
run m6 --mem-conv 6 --rate 1 --list-size 1 --steps 3 --warmup 1 --pool 4096
run m11L1 --list-size 1 --steps 2 --warmup 1 --pool 512
for n in m6 m11L1; do f=m6; [ $n = m11L1 ] && a="--list-size 1 --steps 2 --warmup 1 --pool 512" || a="--mem-conv 6 --rate 1 --list-size 1 --steps 3 --warmup 1 --pool 4096"; python3 bench.py $a --no-cpu-baseline --no-cross-check 2>/dev/null | grep '^{' | tail -1 > $out/r3_${n}_bench.json; done
