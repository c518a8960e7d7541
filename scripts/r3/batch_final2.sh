#!/bin/bash
# round 3, last call: the whole GPU suite + smoke on the final library, the profile evidence of the default bench, a second soak
out=gpurun_out/r3final2; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; tail -2 $out/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
bash scripts/r3/prof_final.sh > $out/prof.log 2>&1; tail -3 $out/prof.log
bash scripts/r3/soak2.sh
