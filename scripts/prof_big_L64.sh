# rocprofv3 kernel stats + SQ/TCC counters of the big-list kernel (m=11 r=5/6 L=64, 8 slots, 8 reads)
export TMPDIR=/tmp
B="python3 bench.py --list-size 64 --steps 1 --warmup 0 --slots 8 --reads-per-step 8 --no-cpu-baseline"
rm -rf gpurun_out/b64_t gpurun_out/b64_p gpurun_out/b64_m
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b64_t -- $B > gpurun_out/b64_t.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/b64_p -- $B > gpurun_out/b64_p.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/b64_q -- $B > gpurun_out/b64_q.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/b64_m -- $B > gpurun_out/b64_m.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/b64_w -- $B > gpurun_out/b64_w.log 2>&1
cat gpurun_out/b64_t/*/*kernel_stats.csv | cut -c1-200
tail -1 gpurun_out/b64_t.log | cut -c1-400
for d in b64_p b64_q b64_m b64_w; do python3 scripts/pmc_summary.py gpurun_out/$d | grep -A10 "step_big" | head -12; done
