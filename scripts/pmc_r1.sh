export TMPDIR=/tmp
B="python3 bench.py --steps 1 --warmup 0 --slots 8 --reads-per-step 8 --no-cpu-baseline --check 0"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_trace -- $B > gpurun_out/p_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/p_a -- $B > gpurun_out/p_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/p_b -- $B > gpurun_out/p_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d gpurun_out/p_c -- $B > gpurun_out/p_c.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --output-format csv -d gpurun_out/p_d -- $B > gpurun_out/p_d.log 2>&1
ls gpurun_out/p_*/*/ | head -30
