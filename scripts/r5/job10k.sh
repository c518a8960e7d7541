#!/bin/bash
# configs[1] at its stated job size (10 000 reads in ONE lva_decode_batch call) is 3.4 % slower than the driver
# bench -- clocks over a 3.5-minute run, or slots that drift out of phase?  Kernel trace of the whole job (one row per launch) +
# clock / power samples beside it -> gpurun_out/r5job/: windows of 4000 launches, early to late.
export TMPDIR=/tmp
out=gpurun_out/r5job; mkdir -p $out
( while true; do echo "t=$(date +%s) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|Power' | tr -s ' ' | tr '\n' ';')"; sleep 5; done ) > $out/smi.log 2>&1 &
smi=$!
s=$(date +%s)
timeout 1200 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --total-reads 10000 --steps 1 --warmup 0 --no-cpu-baseline --no-cross-check > $out/job.log 2>&1
echo "rc=$? wall=$(( $(date +%s)-s )) s"
kill $smi
grep '^{' $out/job.log | tail -1 > $out/r5_10k_reads_under_trace.json
python3 scripts/r5/job10k_windows.py $out/trace $s > $out/r5_10k_windows.txt 2>&1
cat $out/r5_10k_windows.txt
rm -rf $out/trace
python3 bench.py --total-reads 10000 --steps 1 --warmup 0 --no-cpu-baseline --no-cross-check > $out/r5_10k_reads.json 2> $out/job2.err
cut -c1-160 $out/r5_10k_reads.json
