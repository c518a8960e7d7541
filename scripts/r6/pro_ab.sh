#!/bin/bash
# round 6: prologue variants of lva_step_lazy (product = flat staging; variants/pro1.so: + back-pointer bytes and posteriors in the same
# batch; variants/pro2.so: + both table words of the target before the barrier) against round 5's library, one box.
out=gpurun_out/r6/pro; mkdir -p $out
export LVA_TESTING=1
LVA_LIB_PATH=variants/pro2.so timeout 900 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_instances.py -m gpu -x -q 2>&1 | tail -5 > $out/tests_pro2.log; cat $out/tests_pro2.log
show() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s reads/s %9.2f  launch %.3f ms  frac %.3f  build %s" % (sys.argv[2], j["value"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"], j["library"]["build_id"]))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for v in r5 default pro1 pro2; do
  lib=variants/$v.so; [ $v = default ] && lib=nanopore_dna_storage_amd/liblva_hip.so
  LVA_LIB_PATH=$lib timeout 300 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-cross-check --no-extra-configs > $out/${v}.json 2> $out/${v}.err || tail -3 $out/${v}.err
  show $out/${v}.json "headline $v"
done
for o in xcd plain; do
  if [ $o = plain ]; then export LVA_NO_XCD_ORDER=1; else unset LVA_NO_XCD_ORDER; fi
  timeout 300 python3 bench.py --mem-conv 14 --rate 7 --list-size 1 --steps 2 --warmup 1 --pool 64 --reads-per-step 64 --no-cpu-baseline --no-cross-check --no-extra-configs > $out/m14L1_$o.json 2> $out/m14L1_$o.err || tail -3 $out/m14L1_$o.err
  show $out/m14L1_$o.json "m14 r7 L1 $o"
done
