#!/bin/bash
# round 3: kernel traces of the other configurations on the final library (copy the CSVs into profiles/)
export TMPDIR=/tmp
out=gpurun_out/r3prof2; mkdir -p $out
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- python3 bench.py "$@" --no-cpu-baseline --no-cross-check > $out/$name.log 2>&1; cat $out/$name/*/*kernel_stats.csv > $out/r3_${name}_kernel_stats.csv; grep '^{' $out/$name.log | tail -1 > $out/r3_${name}_bench_under_trace.json; head -4 $out/r3_${name}_kernel_stats.csv | cut -c1-70,160-230; }
run m14 --mem-conv 14 --rate 7 --slots 8 --steps 2 --warmup 1 --pool 32
run big64 --list-size 64 --slots 8 --steps 1 --warmup 1 --pool 16
run m8 --mem-conv 8 --rate 3 --msg-len 164 --steps 3 --warmup 1 --pool 1024
run m6 --mem-conv 6 --rate 1 --list-size 1 --steps 3 --warmup 1 --pool 4096
run m11L1 --list-size 1 --steps 2 --warmup 1 --pool 512
