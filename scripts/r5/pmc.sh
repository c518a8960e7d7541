#!/bin/bash
# usage (GPU box, repo root): bash scripts/r5/pmc.sh TAG "kernel regex" bench flags...
# separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ, TA/TCC) restricted to the named kernels -> gpurun_out/r5pmc/r5_TAG_pmc_summary.txt
export TMPDIR=/tmp
out=gpurun_out/r5pmc; mkdir -p $out
tag=$1; re=$2; shift 2
B="python3 bench.py $* --steps 1 --warmup 0 --no-cpu-baseline --no-launch-events --no-cross-check"
run() { name=$1; shift; s=$(date +%s); timeout 600 rocprofv3 --kernel-include-regex "$re" "$@" --output-format csv -d $out/${tag}_$name -- $B > $out/${tag}_$name.log 2>&1; echo "$tag $name rc=$? $(( $(date +%s)-s )) s"; }
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run ta --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
python3 scripts/pmc_summary.py $out/${tag}_fetch $out/${tag}_write $out/${tag}_sq1 $out/${tag}_sq2 $out/${tag}_ta > $out/r5_${tag}_pmc_summary.txt 2>&1
grep '^{' $out/${tag}_fetch.log | tail -1 > $out/r5_${tag}_bench_under_pmc.json
grep -v "^gpurun_out\|^k \|^  *mean" $out/r5_${tag}_pmc_summary.txt | cut -c1-110 | grep -i "big\|lazy\|acs\|list" | head -60
