#!/bin/bash
out=gpurun_out/r5cmp; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_instances.py tests/test_gpu_random.py tests/test_gpu_golden.py -m gpu -x -q -k "big_list or long_and_odd or instances or random or hundreds or L16 or L100 or wide" > $out/pytest.log 2>&1
tail -3 $out/pytest.log
bash scripts/run_variants.sh $out/L16 "--list-size 16 --slots 16 --steps 1 --warmup 0 --pool 16 --cross-check-reads 2" head default
bash scripts/run_variants.sh $out/L12r1 "--list-size 12 --rate 1 --slots 16 --steps 1 --warmup 0 --pool 16 --cross-check-reads 2" head default
bash scripts/run_variants.sh $out/m8L16 "--mem-conv 8 --rate 3 --msg-len 164 --list-size 16 --slots 64 --steps 1 --warmup 0 --pool 128 --cross-check-reads 4" head default
