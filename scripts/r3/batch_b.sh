#!/bin/bash
# round 3, GPU batch B: new host-side tests (chain, forced RCCL group, chunked driver) on the default library;
# merge-loop / output-phase variants: parity (lazy suite) then bench
out=gpurun_out/r3b; mkdir -p $out
python -m pytest tests/test_gpu_chain.py tests/test_gpu_multi.py tests/test_gpu_golden.py -m gpu -x -q > $out/tests_default.log 2>&1
tail -5 $out/tests_default.log
for v in hoist dedupasm pushvar gb1; do
  LVA_LIB_PATH=$PWD/variants/$v.so python -m pytest tests/test_gpu_lazy.py tests/test_gpu_fuzz_m11.py -m gpu -x -q > $out/tests_$v.log 2>&1
  echo "$v: $(tail -1 $out/tests_$v.log)"
done
bash scripts/run_variants.sh $out "--steps 6 --warmup 2 --no-cross-check" default hoist dedupasm pushvar gb1
