export TMPDIR=/tmp
B="python3 bench.py --list-size 64 --steps 1 --warmup 0 --slots 1 --reads-per-step 1 --no-cpu-baseline --check 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x64_t -- $B > gpurun_out/x64_t.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/x64_p -- $B > gpurun_out/x64_p.log 2>&1
cat gpurun_out/x64_t/*/*kernel_stats.csv | cut -c1-160
python3 scripts/pmc_summary.py gpurun_out/x64_p | grep -A8 "step_exact"
grep lva_step_exact gpurun_out/x64_t/*/*kernel_trace.csv | head -2 | cut -d, -f8- | cut -c60-200
