#!/bin/bash
# round 4: the benchmark shape at 48 / 64 / 96 / 128 read slots (phase-aligned launches)
out=gpurun_out/r4slots; mkdir -p $out
for s in 48 64 96 128; do
  python bench.py --slots $s --reads-per-step 256 --pool 512 --steps 3 --warmup 1 --no-cpu-baseline --no-cross-check > $out/s$s.json 2> $out/s$s.err
  python - $out/s$s.json $s <<'PY'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=j["roofline"]
print("slots %s: %.2f reads/s  %.3f ms/launch  frac %.3f  active %.1f" % (sys.argv[2], j["value"], r["avg_launch_ms"], r["frac"], j["config"]["mean_active_slots"]))
PY
done
