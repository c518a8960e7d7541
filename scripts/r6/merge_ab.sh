#!/bin/bash
# round 6: the merge loop with tagged fingerprints (one unsigned minimum instead of eight compare + select pairs, 8-bit match records,
# the one-match-per-entry test behind the loop) against the library before it (variants/final1.so), one box.  Parity first.
out=gpurun_out/r6/merge; mkdir -p $out
export LVA_TESTING=1
timeout 900 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_instances.py tests/test_gpu_lazy_wide.py -m gpu -x -q 2>&1 | tail -5 > $out/tests.log; cat $out/tests.log
run() { name=$1; shift; for v in final1 default; do
  lib=variants/$v.so; [ $v = default ] && lib=nanopore_dna_storage_amd/liblva_hip.so
  LVA_LIB_PATH=$lib timeout 300 python3 bench.py "$@" --no-cpu-baseline --no-cross-check --no-extra-configs > $out/${name}_$v.json 2> $out/${name}_$v.err || tail -2 $out/${name}_$v.err
  python3 - <<PY
import json
try:
    j=json.loads(open("$out/${name}_$v.json").read().strip().splitlines()[-1])
    print("%-14s %-8s reads/s %9.2f  launch %.3f ms  frac %.3f  fixups %s" % ("$name", "$v", j["value"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"], j["config"]["fixup_reason"]))
except Exception as e: print("$name $v failed", e)
PY
done; }
run headline --steps 4 --warmup 1
run m11_L4 --list-size 4 --steps 2 --warmup 1
run m8_r3_L8 --mem-conv 8 --rate 3 --msg-len 164 --steps 3 --warmup 1 --pool 1024
run m11_rate1 --mem-conv 11 --rate 1 --steps 2 --warmup 1 --pool 128
