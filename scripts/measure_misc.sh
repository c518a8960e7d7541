#!/bin/bash
# Small measurements quoted in DESIGN.md / profiles/README.md (GPU box):
#  (1) latency of the compatibility mode: one process per read through viterbi/viterbi_nanopore.out (helper.py:305's call)
#  (2) the Reed-Solomon outer decode at the size of the paper's experiment 7
cd "$(dirname "$0")/.."
P=tests/golden/m11_r5_L8_clean.post
for i in 1 2 3; do
  t0=$(date +%s.%N)
  viterbi/viterbi_nanopore.out -m decode -i $P -o /tmp/cli_out.txt --mem-conv 11 --msg-len 180 -l 8 -t 8 -r 5 '' --max-deviation 20
  t1=$(date +%s.%N); echo "cli call $i: $(python3 -c "print('%.2f' % ($t1 - $t0))") s wall"
done
cmp <(cat /tmp/cli_out.txt) tests/golden/m11_r5_L8_clean.list && echo "cli output identical to the reference list"
python3 - <<'PY'
import time
import numpy as np
from nanopore_dna_storage_amd import rs_code
rng = np.random.default_rng(7)
nd, red, spr = 564, 169, 9
reads = [bytes(rng.integers(0, 256, size=2 * spr, dtype=np.uint8)) for _ in range(nd)]
t = time.time(); enc = rs_code.MainEncoder(reads, red); t_enc1 = time.time() - t
total = nd + red
erased = set(rng.choice(total, size=100, replace=False).tolist())
rx = [[i, enc[i]] for i in range(total) if i not in erased]
for j in rng.choice(len(rx), size=30, replace=False):
    rx[j][1] = bytes(rng.integers(0, 256, size=2 * spr, dtype=np.uint8))
ts = []
for _ in range(5):
    t = time.time(); dec = rs_code.MainDecoder(rx, red, total); ts.append(time.time() - t)
assert dec == reads
print("RS exp-7 size (9 columns, 564+169 oligos, 100 erasures + 30 errors): encode %.1f ms (first call %.1f), decode %s ms"
      % (1e3 * min([t_enc1]), 1e3 * t_enc1, ["%.1f" % (1e3 * x) for x in ts]))
PY
