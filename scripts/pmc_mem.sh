export TMPDIR=/tmp
B="python3 bench.py --steps 1 --warmup 0 --slots 8 --reads-per-step 8 --no-cpu-baseline --check 0"
run() { name=$1; shift; timeout 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$name -- $B > gpurun_out/$name.log 2>&1 || echo "$name failed"; }
run m_a TA_TA_BUSY_sum GRBM_GUI_ACTIVE
run m_b TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run m_c TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum
run m_d TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run m_e TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
run m_f TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum
python3 scripts/pmc_summary.py gpurun_out/m_a gpurun_out/m_b gpurun_out/m_c gpurun_out/m_d gpurun_out/m_e gpurun_out/m_f 2>&1 | grep -A3 "lva_step_fast"
