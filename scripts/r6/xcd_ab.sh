#!/bin/bash
# round 6: XCD-aware tile order (PosRec::xs) against the plain order, SAME library, one box: parity of the kernels it touches first,
# then the driver's command shape (shorter) with its extra configurations, plus a one-bit-step code and the L = 1 kernel.
out=gpurun_out/r6/xcd; mkdir -p $out
export LVA_TESTING=1
if [ "$1" != "notest" ]; then
  timeout 1200 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_instances.py -m gpu -x -q 2>&1 | tail -5 > $out/tests.log; cat $out/tests.log
fi
show() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    line = "%-22s reads/s %9.2f  launch %.3f ms  frac %.3f" % (sys.argv[2], j["value"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"])
    for e in j.get("extra_configs") or []:
        line += "  | %s %.2f reads/s %.3f ms" % (e["workload"].split(":")[0], e["reads_s"], e["avg_launch_ms"])
    print(line)
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for rep in 1 2; do for v in xcd plain; do
  if [ $v = plain ]; then export LVA_NO_XCD_ORDER=1; else unset LVA_NO_XCD_ORDER; fi
  timeout 400 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-cross-check > $out/${v}_$rep.json 2> $out/${v}_$rep.err || tail -3 $out/${v}_$rep.err
  show $out/${v}_$rep.json "headline $v $rep"
done; done
for v in xcd plain; do
  if [ $v = plain ]; then export LVA_NO_XCD_ORDER=1; else unset LVA_NO_XCD_ORDER; fi
  timeout 300 python3 bench.py --mem-conv 11 --rate 1 --steps 2 --warmup 1 --pool 128 --no-cpu-baseline --no-cross-check --no-extra-configs > $out/rate1_$v.json 2> $out/rate1_$v.err || tail -3 $out/rate1_$v.err
  show $out/rate1_$v.json "m11 rate 1/2 L8 $v"
  timeout 300 python3 bench.py --list-size 1 --steps 2 --warmup 1 --no-cpu-baseline --no-cross-check --no-extra-configs > $out/L1_$v.json 2> $out/L1_$v.err || tail -3 $out/L1_$v.err
  show $out/L1_$v.json "m11 L1 $v"
done
