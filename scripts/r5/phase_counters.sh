#!/bin/bash
# usage (GPU box, repo root): bash scripts/r5/phase_counters.sh -- L2 hits / misses, FETCH and WRITE of lva_step_big_rec<64> per PHASE:
# the library as it is, with the output phase executed twice and with the merge executed twice (variants/out2.so, variants/merge2.so:
# results identical); the differences are what each phase costs the memory system.  -> gpurun_out/r5phase/
export TMPDIR=/tmp
out=gpurun_out/r5phase; mkdir -p $out
B="python3 bench.py --list-size 64 --slots 8 --pool 8 --steps 1 --warmup 0 --no-cpu-baseline --no-launch-events --no-cross-check"
for v in default out2 merge2; do
  if [ "$v" = default ]; then unset LVA_LIB_PATH; else export LVA_LIB_PATH=$PWD/variants/$v.so; fi
  for c in "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_WAVES" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    n=$(echo $c | cut -d' ' -f1)
    s=$(date +%s)
    timeout 400 rocprofv3 --kernel-include-regex "lva_step_big_rec" --pmc $c --output-format csv -d $out/${v}_$n -- $B > $out/${v}_$n.log 2>&1
    echo "$v $n rc=$? $(( $(date +%s)-s )) s"
  done
  python3 scripts/pmc_summary.py $out/${v}_TCC_HIT_sum $out/${v}_FETCH_SIZE $out/${v}_WRITE_SIZE $out/${v}_SQ_INSTS_VMEM_RD $out/${v}_TCP_TCC_READ_REQ_sum > $out/r5_big64_phase_${v}_pmc_summary.txt 2>&1
  grep -v "^gpurun_out\|^k \|^  *mean" $out/r5_big64_phase_${v}_pmc_summary.txt | cut -c1-110
done
unset LVA_LIB_PATH
