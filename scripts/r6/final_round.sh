#!/bin/bash
# round 6, the last GPU call: the whole GPU suite on the final library, then (only if it is green) the profile set of scripts/r6/prof_final2.sh
mkdir -p gpurun_out
timeout 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r6_final_gputests.log; cat gpurun_out/r6_final_gputests.log
if grep -q " passed" gpurun_out/r6_final_gputests.log && ! grep -q "failed\|error" gpurun_out/r6_final_gputests.log; then
  bash scripts/r6/prof_final2.sh
else
  echo "suite not green: no profile"
fi
