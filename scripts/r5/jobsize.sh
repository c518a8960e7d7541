#!/bin/bash
# VERDICT r4 item 5 (is configs[1] at its stated job size slower than the driver bench?): on ONE box, back to back, without a
# profiler: the driver's command, 10 000 reads in one lva_decode_batch call, and the same 10 000 reads as five calls of 2 000
# (per-call wall times: drift over a 3.5-minute run would show as slower late calls), with clock / power samples beside them.
out=gpurun_out/r5job; mkdir -p $out
( while true; do echo "t=$(date +%s) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | tr -s ' \t' ' ' | tr '\n' ';')"; sleep 10; done ) > $out/smi_jobsize.log 2>&1 &
smi=$!
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check > $out/r5_jobsize_driver_bench.json 2> $out/js1.err
python3 bench.py --total-reads 10000 --steps 1 --warmup 0 --no-cpu-baseline --no-cross-check > $out/r5_10k_reads.json 2> $out/js2.err
python3 bench.py --total-reads 2000 --steps 5 --warmup 0 --no-cpu-baseline --no-cross-check > $out/r5_5x2000_reads.json 2> $out/js3.err
kill $smi
python3 - $out <<'PY'
import json, sys
d = sys.argv[1]
for f in ("r5_jobsize_driver_bench.json", "r5_10k_reads.json", "r5_5x2000_reads.json"):
    j = json.loads([l for l in open(d + "/" + f) if l.startswith("{")][-1]); r = j["roofline"]
    print("%-32s %7.2f reads/s  kernel %.3f ms/launch  span %.3f  active slots %.1f  steps [ms] %s" % (
        f, j["value"], r["avg_launch_ms"], r["span"]["ms_per_launch"], j["config"]["mean_active_slots"], j["config"]["step_ms_rank0"][:6]))
PY
grep -c sclk $out/smi_jobsize.log; awk 'NR%6==2' $out/smi_jobsize.log | cut -c1-200 | head -12
