#!/bin/bash
# round 3: counters of the dominant kernels of configs[3] (m=14 L=8) and configs[4] (m=11 L=64) on the final library
export TMPDIR=/tmp
out=gpurun_out/r3pmc2; mkdir -p $out
pass() { tag=$1; re=$2; shift 2; B="python3 bench.py $* --steps 1 --warmup 0 --no-cpu-baseline --no-launch-events --no-cross-check"
  run() { name=$1; shift; s=$(date +%s); timeout 600 rocprofv3 --kernel-include-regex "$re" "$@" --output-format csv -d $out/${tag}_$name -- $B > $out/${tag}_$name.log 2>&1; echo "$tag $name rc=$? $(( $(date +%s)-s )) s"; }
  run fetch --pmc FETCH_SIZE
  run write --pmc WRITE_SIZE
  run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  run sq2 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  run ta --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
  python3 scripts/pmc_summary.py $out/${tag}_fetch $out/${tag}_write $out/${tag}_sq1 $out/${tag}_sq2 $out/${tag}_ta > $out/r3_${tag}_pmc_summary.txt 2>&1
  grep '^{' $out/${tag}_fetch.log | tail -1 > $out/r3_${tag}_bench_under_pmc.json
}
pass m14 "lva_step_lazy" --mem-conv 14 --rate 7 --slots 8 --reads-per-step 8 --pool 8
pass big64 "lva_step_big|lva_step_fixup_wave" --list-size 64 --slots 8 --reads-per-step 8 --pool 8
grep -h "lva_step\|SIZE\|TA_TA\|GUI\|ACTIVE_INST_VALU \|INSTS_VALU\|VMEM" $out/r3_m14_pmc_summary.txt $out/r3_big64_pmc_summary.txt | cut -c1-110 | head -60
