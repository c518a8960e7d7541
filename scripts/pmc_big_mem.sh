# memory-path counters of lva_step_big<64,3> (m=11 r=5/6 L=64, 8 slots): two counters per pass, each pass under its own timeout
export TMPDIR=/tmp
B="python3 bench.py --list-size 64 --steps 1 --warmup 0 --slots 8 --reads-per-step 8 --no-cpu-baseline"
run() { name=$1; shift; rm -rf gpurun_out/$name; timeout 150 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$name -- $B > gpurun_out/$name.log 2>&1 || echo "$name failed"; }
run g_a TCC_HIT_sum TCC_MISS_sum
run g_b TCC_REQ_sum TCC_READ_sum
run g_c TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum
run g_d TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum
run g_e TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum
run g_f TA_TA_BUSY_sum GRBM_GUI_ACTIVE
run g_g TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum
python3 scripts/pmc_summary.py gpurun_out/g_a gpurun_out/g_b gpurun_out/g_c gpurun_out/g_d gpurun_out/g_e gpurun_out/g_f gpurun_out/g_g 2>&1 | grep -A2 "lva_step_big"
