#!/bin/bash
# round 6: first-phase variants of lva_step_lazy (variants/*.so built with -DLVA_STAGE4 / -DLVA_HOIST), each with the XCD-aware tile
# order and with the plain one, one box.  Parity of the most aggressive variant first.
out=gpurun_out/r6/hoist; mkdir -p $out
export LVA_TESTING=1
LVA_LIB_PATH=variants/both.so timeout 900 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_instances.py -m gpu -x -q 2>&1 | tail -5 > $out/tests_both.log; cat $out/tests_both.log
show() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s reads/s %9.2f  launch %.3f ms  frac %.3f  build %s" % (sys.argv[2], j["value"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"], j["library"]["build_id"]))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for v in default stage4 hoist both; do for o in xcd plain; do
  lib=variants/$v.so; [ $v = default ] && lib=nanopore_dna_storage_amd/liblva_hip.so
  if [ $o = plain ]; then export LVA_NO_XCD_ORDER=1; else unset LVA_NO_XCD_ORDER; fi
  LVA_LIB_PATH=$lib timeout 300 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-cross-check --no-extra-configs > $out/${v}_$o.json 2> $out/${v}_$o.err || tail -3 $out/${v}_$o.err
  show $out/${v}_$o.json "$v $o"
done; done
