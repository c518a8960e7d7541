#!/bin/bash
# round 3: fix-up pass of the big-list kernel with its candidates in registers (was: scratch), wide message moves, 64 registers
out=gpurun_out/r3p; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "wave or tie or big_list or long_and_odd or overflow" > $out/tests.log 2>&1; tail -3 $out/tests.log
timeout 600 python -m pytest tests -x -q -m gpu -k "L64 or golden" > $out/tests2.log 2>&1; tail -3 $out/tests2.log
for i in 1 2; do
timeout 300 python bench.py --list-size 64 --slots 8 --steps 1 --warmup 1 --pool 16 --no-cpu-baseline --no-cross-check 2>&1 | grep '^{' | tail -1 > $out/big64_$i.json
python - <<PY
import json; d=json.load(open('$out/big64_$i.json')); r=d['roofline']; print('big64', d['value'], r['avg_launch_ms'], r['pair']['avg_launch_ms'], r['frac'])
PY
done
timeout 300 python bench.py --mem-conv 8 --rate 3 --msg-len 164 --list-size 64 --steps 1 --warmup 1 --no-cpu-baseline --no-cross-check 2>&1 | grep '^{' | tail -1 > $out/m8L64.json
python - <<PY
import json; d=json.load(open('$out/m8L64.json')); r=d['roofline']; print('m8L64', d['value'], r['avg_launch_ms'], r['pair']['avg_launch_ms'], r['frac'])
PY
