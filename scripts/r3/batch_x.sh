#!/bin/bash
# round 3: L = 1 kernel, paired message loads only where one of the two targets has a winner
out=gpurun_out/r3x; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q > $out/tests.log 2>&1; tail -2 $out/tests.log
b() { name=$1; shift; timeout 300 python bench.py "$@" --no-cpu-baseline --no-cross-check 2>&1 | grep '^{' | tail -1 > $out/$name.json
python - <<PY
import json; d=json.load(open('$out/$name.json')); r=d['roofline']; print('$name', round(d['value'],2), round(r['avg_launch_ms'],4), round(r['frac'],4))
PY
}
b m6 --mem-conv 6 --rate 1 --list-size 1 --steps 3 --warmup 1 --pool 4096
b m6b --mem-conv 6 --rate 1 --list-size 1 --steps 3 --warmup 1 --pool 4096
b m6c --mem-conv 6 --rate 1 --list-size 1 --steps 3 --warmup 1 --pool 4096
b m8L1 --mem-conv 8 --rate 3 --msg-len 164 --list-size 1 --steps 3 --warmup 1 --pool 1024
b m11L1 --list-size 1 --steps 2 --warmup 1
b m11L1b --list-size 1 --steps 2 --warmup 1
