#!/bin/bash
# round 3: m=6 L=1 slot-count sweep with the new L=1 kernel
out=gpurun_out/r3r; mkdir -p $out
b() { name=$1; shift; timeout 300 python bench.py "$@" --no-cpu-baseline --no-cross-check 2>&1 | grep '^{' | tail -1 > $out/$name.json
python - <<PY
import json; d=json.load(open('$out/$name.json')); r=d['roofline']; print('$name', round(d['value'],2), round(r['avg_launch_ms'],4), round(r['frac'],4), d['config'].get('mean_active_slots'), r.get('span'))
PY
}
for s in 64 128 256 512 1024 2048; do b m6_s$s --mem-conv 6 --rate 1 --list-size 1 --steps 3 --warmup 1 --pool 4096 --slots $s; done
for s in 128 256 512; do b m8_s$s --mem-conv 8 --rate 3 --msg-len 164 --list-size 1 --steps 3 --warmup 1 --pool 1024 --slots $s; done
