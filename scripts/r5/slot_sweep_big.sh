#!/bin/bash
# usage (GPU box, repo root): bash scripts/r5/slot_sweep_big.sh -- reads/s against the number of read slots for the configurations the
# profile set runs at 8 slots (configs[4]: m=11 L=64; configs[3]: m=14 L=8) -> gpurun_out/r5slots/
mkdir -p gpurun_out/r5slots
one() { tag=$1; s=$2; shift 2
  timeout 500 python3 bench.py "$@" --steps 1 --warmup 1 --slots $s --reads-per-step $((s*2)) --pool $((s*2)) --no-cpu-baseline --no-cross-check > gpurun_out/r5slots/$tag.s$s.json 2> gpurun_out/r5slots/$tag.s$s.err
  python3 -c "
import json,sys
j=json.loads([l for l in open('gpurun_out/r5slots/$tag.s$s.json') if l.startswith('{')][-1]); r=j['roofline']
print('$tag', $s, 'reads/s %.3f  launch %.4f ms  pair %.4f  frac %.3f  active %.1f' % (j['value'], r['avg_launch_ms'], (r.get('pair') or {}).get('avg_launch_ms') or 0, r['frac'], j['config']['mean_active_slots']))"
}
for s in 8 16 32; do one L64 $s --list-size 64; done
for s in 8 16 32; do one m14 $s --mem-conv 14 --rate 7; done
