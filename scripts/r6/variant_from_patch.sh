#!/bin/bash
# usage: scripts/r6/variant_from_patch.sh NAME 'python expression list' -- builds variants/NAME.so from a patched COPY of csrc
# (experiments stay out of the product source; variants/ is git-ignored scratch).  The patch is a python snippet that edits the string `s` = lva_kernels.hip.
set -e
name=$1; patch=$2
root=$(cd "$(dirname "$0")/../.." && pwd)
tmp=/tmp/variant_$name; rm -rf $tmp; mkdir -p $tmp/csrc $tmp/include; cp $root/nanopore_dna_storage_amd/csrc/* $tmp/csrc/; cp $root/include/*.h $tmp/include/
python3 - "$tmp/csrc/lva_kernels.hip" <<PY
import sys
p=sys.argv[1]; s=open(p).read()
def rep(a,b,cnt=1):
    global s
    assert s.count(a)==cnt, (s.count(a), a[:80])
    s=s.replace(a,b)
$patch
open(p,'w').write(s)
PY
mkdir -p $root/variants
cd $tmp/csrc
sed -i 's#../../include/lva_decoder.h#../include/lva_decoder.h#' *.cpp *.h *.hip 2>/dev/null || true
# the build id every bench line and counter file names: hash of the PATCHED sources + the variant's name
id=$(cat lva_api.cpp lva_code.cpp lva_kernels.hip bc_kernels.hip rs_kernels.hip lva_code.h lva_device.h lva_kernels.h bc_kernels.h ../include/lva_decoder.h | sha256sum | cut -c1-12)-$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable -DLVA_BUILD_ID="\"$id\"" -shared -o $root/variants/$name.so lva_api.cpp lva_code.cpp lva_kernels.hip bc_kernels.hip rs_kernels.hip
echo built variants/$name.so build $id
