#!/bin/bash
# bench lines of the other configurations on the final library (no profiler) -> gpurun_out/r4prof/r4_*_bench.json
out=gpurun_out/r4prof; mkdir -p $out
nb() { name=$1; shift; python3 bench.py "$@" --no-cpu-baseline --no-cross-check > $out/r4_${name}_bench.json 2> $out/nb_$name.err; python3 - $out/r4_${name}_bench.json $name <<'PY'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=j["roofline"]
print("%-6s %9.3f reads/s  %.4f ms  frac %.4f  %d GB/s" % (sys.argv[2], j["value"], r["avg_launch_ms"], r["frac"], r["achieved"]))
PY
}
nb m14 --mem-conv 14 --rate 7 --slots 8 --steps 2 --warmup 1 --pool 32
nb m8 --mem-conv 8 --rate 3 --msg-len 164 --steps 3 --warmup 1 --pool 1024
nb m6 --mem-conv 6 --rate 1 --list-size 1 --steps 3 --warmup 1 --pool 4096
nb m11L1 --list-size 1 --steps 2 --warmup 1 --pool 512
nb rate1 --mem-conv 11 --rate 1 --steps 2 --warmup 1 --pool 128
nb L64 --list-size 64 --slots 8 --steps 1 --warmup 1 --pool 16
