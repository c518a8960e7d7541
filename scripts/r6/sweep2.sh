#!/bin/bash
# round 6: the product library (one-round-trip first phase in lva_step_lazy; XCD-aware tile order in the L = 1 kernel) against round 5's
# (variants/r5.so) on the other shapes, one box.
out=gpurun_out/r6/sweep2; mkdir -p $out
export LVA_TESTING=1
run() { name=$1; shift; for v in r5 default; do
  lib=variants/$v.so; [ $v = default ] && lib=nanopore_dna_storage_amd/liblva_hip.so
  LVA_LIB_PATH=$lib timeout 300 python3 bench.py "$@" --no-cpu-baseline --no-cross-check --no-extra-configs > $out/${name}_$v.json 2> $out/${name}_$v.err || tail -2 $out/${name}_$v.err
  python3 - <<PY
import json
try:
    j=json.loads(open("$out/${name}_$v.json").read().strip().splitlines()[-1])
    print("%-14s %-8s reads/s %9.2f  launch %.3f ms  frac %.3f  slots %d" % ("$name", "$v", j["value"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"], j["config"]["slots"]))
except Exception as e: print("$name $v failed", e)
PY
done; }
run m14_r7_L8 --mem-conv 14 --rate 7 --steps 1 --warmup 1 --pool 64 --reads-per-step 64
run m14_r7_L1 --mem-conv 14 --rate 7 --list-size 1 --steps 2 --warmup 1 --pool 64 --reads-per-step 64
run m11_L1 --list-size 1 --steps 2 --warmup 1
run m8_r3_L8 --mem-conv 8 --rate 3 --msg-len 164 --steps 3 --warmup 1 --pool 1024
run m11_rate1 --mem-conv 11 --rate 1 --steps 2 --warmup 1 --pool 128
run m11_L4 --list-size 4 --steps 2 --warmup 1
run m11_L2 --list-size 2 --steps 2 --warmup 1
run m6_r1_L8 --mem-conv 6 --rate 1 --list-size 8 --steps 3 --warmup 1 --pool 4096
