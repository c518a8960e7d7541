#!/bin/bash
# the whole GPU suite with per-test durations -> gpurun_out/r5suite/
out=gpurun_out/r5suite; mkdir -p $out
s=$(date +%s); echo "cpus $(nproc)"
timeout 1500 python3 -m pytest tests -m gpu -q --durations=60 "$@" > $out/pytest.log 2>&1
echo "rc=$? wall=$(( $(date +%s)-s )) s"
tail -75 $out/pytest.log
