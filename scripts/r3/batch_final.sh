#!/bin/bash
# round 3: the whole GPU suite + smoke on the final library, then the profile evidence (scripts/r3/prof_final.sh), then the other configurations
out=gpurun_out/r3final; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; tail -3 $out/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
bash scripts/r3/prof_final.sh > $out/prof.log 2>&1; tail -5 $out/prof.log
bash scripts/run_variants.sh $out/m14 "--mem-conv 14 --rate 7 --slots 8 --steps 2 --warmup 1 --pool 32 --no-cross-check" default
bash scripts/run_variants.sh $out/m8 "--mem-conv 8 --rate 3 --msg-len 164 --steps 3 --warmup 1 --pool 1024 --no-cross-check" default
bash scripts/run_variants.sh $out/m6 "--mem-conv 6 --rate 1 --list-size 1 --steps 3 --warmup 1 --pool 4096 --no-cross-check" default
bash scripts/run_variants.sh $out/L64 "--list-size 64 --slots 8 --steps 1 --warmup 1 --pool 16 --no-cross-check" default
bash scripts/run_variants.sh $out/m8L64 "--mem-conv 8 --rate 3 --msg-len 164 --list-size 64 --slots 32 --steps 1 --warmup 1 --pool 64 --no-cross-check" default
bash scripts/run_variants.sh $out/m11L1 "--list-size 1 --steps 2 --warmup 1 --pool 512 --no-cross-check" default
