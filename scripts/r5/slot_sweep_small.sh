#!/bin/bash
# usage (GPU box, repo root): bash scripts/r5/slot_sweep_small.sh -- reads/s against the number of read slots for the small trellises
# (m = 6: L = 1 and L = 8; m = 8 r = 3/4 L = 8) -> gpurun_out/r5m6/
mkdir -p gpurun_out/r5m6
one() { tag=$1; s=$2; shift 2
  python3 bench.py "$@" --steps 3 --warmup 1 --slots $s --reads-per-step $((s*4)) --pool 8192 --no-cpu-baseline > gpurun_out/r5m6/$tag.s$s.json 2> gpurun_out/r5m6/$tag.s$s.err
  python3 -c "
import json,sys
j=json.loads([l for l in open('gpurun_out/r5m6/$tag.s$s.json') if l.startswith('{')][-1]); r=j['roofline']
print('$tag', $s, 'reads/s %.1f  launch %.4f ms  frac %.3f  end to end %.0f GB/s' % (j['value'], r['avg_launch_ms'], r['frac'], r['end_to_end_achieved']))"
}
for s in 1024 2048 4096 8192; do one m6L1 $s --mem-conv 6 --rate 1 --list-size 1; done
for s in 1024 2048 4096; do one m6L8 $s --mem-conv 6 --rate 1 --list-size 8; done
for s in 256 512; do one m8L8 $s --mem-conv 8 --rate 3 --msg-len 164; done
for s in 64 128; do one m11L1 $s --list-size 1; done
