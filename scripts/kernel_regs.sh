#!/bin/bash
# Register / LDS / scratch usage of every kernel in lva_kernels.hip (from the gfx950 assembly metadata).
cd "$(dirname "$0")/../nanopore_dna_storage_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only $1 -S -o /tmp/lva_k.s lva_kernels.hip || exit 1
python3 - <<'PY'
import re
t = open('/tmp/lva_k.s').read()
for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)', t):
    pass
blocks = t.split('- .agpr_count:')[1:]
for b in blocks:
    g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, b).group(1)
    print('%-60s vgpr %3s sgpr %3s lds %6s scratch %4s' % (g('name')[:60], g('vgpr_count'), g('sgpr_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size')))
PY
