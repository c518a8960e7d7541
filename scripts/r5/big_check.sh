#!/bin/bash
# round 5: parity of the list kernel (three message planes, L >= 32) + its bench lines -> gpurun_out/r5big
out=gpurun_out/r5big; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_golden.py -m gpu -x -q -k "big_list or list64 or long_and_odd or L64 or L32 or L100 or instances" > $out/pytest.log 2>&1
tail -5 $out/pytest.log
nb() { name=$1; shift; timeout 600 python3 bench.py "$@" --no-cpu-baseline > $out/${name}_bench.json 2> $out/nb_$name.err; python3 - $out/${name}_bench.json $name <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=j["roofline"]
    print("%-6s %9.3f reads/s  %.4f ms  frac %.4f  %d GB/s" % (sys.argv[2], j["value"], r["avg_launch_ms"], r["frac"], r["achieved"]))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
nb L64 --list-size 64 --slots 8 --steps 1 --warmup 1 --pool 16 --cross-check-reads 2
nb m8L64 --mem-conv 8 --rate 3 --msg-len 164 --list-size 64 --slots 32 --steps 1 --warmup 1 --pool 64 --cross-check-reads 4
nb L32 --list-size 32 --slots 16 --steps 1 --warmup 1 --pool 32 --no-cross-check
