#!/bin/bash
# round 3: L = 1 kernel with two-wavefront workgroups for rate-1/2 codes
out=gpurun_out/r3w; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q > $out/tests.log 2>&1; tail -2 $out/tests.log
b() { name=$1; shift; timeout 300 python bench.py "$@" --no-cpu-baseline --no-cross-check 2>&1 | grep '^{' | tail -1 > $out/$name.json
python - <<PY
import json; d=json.load(open('$out/$name.json')); r=d['roofline']; print('$name', round(d['value'],2), round(r['avg_launch_ms'],4), round(r['frac'],4))
PY
}
b m6 --mem-conv 6 --rate 1 --list-size 1 --steps 3 --warmup 1 --pool 4096
b m6b --mem-conv 6 --rate 1 --list-size 1 --steps 3 --warmup 1 --pool 4096
b m6_s512 --mem-conv 6 --rate 1 --list-size 1 --steps 3 --warmup 1 --pool 4096 --slots 512
b m11r1L1 --mem-conv 11 --rate 1 --list-size 1 --steps 2 --warmup 1
