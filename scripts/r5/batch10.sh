#!/bin/bash
out=gpurun_out/r5pair; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_instances.py tests/test_gpu_random.py tests/test_gpu_golden.py tests/test_gpu_chain.py -m gpu -x -q -k "big_list or long_and_odd or instances or random or hundreds or L16 or L100 or L64 or list64 or wide" > $out/pytest.log 2>&1
tail -3 $out/pytest.log
bash scripts/run_variants.sh $out/L64 "--list-size 64 --slots 8 --steps 1 --warmup 0 --pool 8 --cross-check-reads 1" head default
bash scripts/run_variants.sh $out/L16 "--list-size 16 --slots 16 --steps 1 --warmup 0 --pool 16 --cross-check-reads 2" head default
bash scripts/run_variants.sh $out/L12r1 "--list-size 12 --rate 1 --slots 16 --steps 1 --warmup 0 --pool 16 --cross-check-reads 2" head default
bash scripts/run_variants.sh $out/m8L64 "--mem-conv 8 --rate 3 --msg-len 164 --list-size 64 --slots 32 --steps 1 --warmup 0 --pool 64 --cross-check-reads 2" head default
