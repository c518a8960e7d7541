#!/bin/bash
# round 3, GPU batch J: proof planes (twin tags + identity bytes): duplicates proven without fetching messages.  default = proof on; noproof = off
out=gpurun_out/r3j; mkdir -p $out
python -m pytest tests/test_gpu_lazy.py tests/test_gpu_fuzz_m11.py tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q > $out/tests_default.log 2>&1
echo "proof: $(tail -1 $out/tests_default.log)"
bash scripts/run_variants.sh $out "--steps 6 --warmup 2" noproof default
bash scripts/run_variants.sh $out/m14 "--mem-conv 14 --rate 7 --slots 8 --steps 2 --warmup 1 --pool 32 --no-cross-check" noproof default
bash scripts/run_variants.sh $out/m8 "--mem-conv 8 --rate 3 --msg-len 164 --steps 3 --warmup 1 --pool 1024 --no-cross-check" noproof default
