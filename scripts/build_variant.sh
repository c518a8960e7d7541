#!/bin/bash
# Build a kernel-experiment variant of the product library: scripts/build_variant.sh NAME "-DLVA_X=1 ..."
# -> variants/NAME.so (git-ignored; travels to the GPU box).  Use with LVA_LIB_PATH=variants/NAME.so.
set -e
cd "$(dirname "$0")/../nanopore_dna_storage_amd/csrc"
mkdir -p ../../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
  -Wno-unused-variable $2 -shared -o ../../variants/$1.so lva_api.cpp lva_code.cpp lva_kernels.hip bc_kernels.hip rs_kernels.hip
echo built variants/$1.so
