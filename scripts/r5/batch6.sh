#!/bin/bash
bash scripts/run_variants.sh gpurun_out/r5v6 "--steps 4 --warmup 1 --cross-check-reads 8" nopf default g2w8 g4w7 g4w6 g8w5
