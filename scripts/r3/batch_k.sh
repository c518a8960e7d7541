#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3k; mkdir -p $out
B="python3 bench.py --steps 1 --warmup 0 --reads-per-step 64 --pool 64 --no-cpu-baseline --no-launch-events --no-cross-check"
for v in default noproof; do
  if [ "$v" = default ]; then unset LVA_LIB_PATH; else export LVA_LIB_PATH=$PWD/variants/$v.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$v -- $B > $out/trace_$v.log 2>&1
  echo "== $v"; cat $out/trace_$v/*/*kernel_stats.csv | head -4 | cut -c1-60,150-260
  timeout 300 rocprofv3 --kernel-include-regex "lva_step_lazy" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $out/sq_$v -- $B > $out/sq_$v.log 2>&1
  python3 scripts/pmc_summary.py $out/sq_$v | grep -v "^$" | cut -c1-100
done
