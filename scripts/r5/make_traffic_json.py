#!/usr/bin/env python3
"""profiles/r5_traffic.json from the counter CSVs of scripts/r5/prof_final.sh (gpurun_out/r5prof/{fetch,write,sq1,sq2,ta}) and
the bench line of the same run: HBM bytes per launch of the dominant kernel at the DEFAULT 128 slots, and what the SQ / TA
counters say holds it back (bench.py copies `limiter` into its roofline object).

Since round 4 the slots are phase-aligned -- a launch runs ONE instance of lva_step_lazy (anchor on even launches, odd-step on odd
ones) over all slots, so "per launch" is the mean over the two instances weighted by their launch counts.

    python scripts/r5/make_traffic_json.py gpurun_out/r5prof profiles/r5_traffic.json
"""
import glob
import json
import os
import sys

import pandas as pd

src, dst = sys.argv[1], sys.argv[2]


def counters(name):
    f = max(glob.glob("%s/%s/*/*counter_collection.csv" % (src, name)), key=os.path.getmtime)
    df = pd.read_csv(f)
    df["k"] = df["Kernel_Name"].str.extract(r"(lva_step_lazy<[^>]*>)")[0]
    g = df.groupby(["k", "Counter_Name"])["Counter_Value"]
    return g.mean().unstack(), g.count().unstack()


(fetch, nf), (write, _), (sq1, _), (sq2, _), (ta, _) = (counters(n) for n in ("fetch", "write", "sq1", "sq2", "ta"))
bench = json.loads([ln for ln in open(src + "/r5_lazy128_bench_under_pmc.json") if ln.startswith("{")][-1])
kern, calls = {}, {}
for k in fetch.index:
    cyc = ta.loc[k, "GRBM_GUI_ACTIVE"] / 8.0                     # summed over the 8 XCDs
    calls[k] = int(nf.loc[k, "FETCH_SIZE"])
    kern[k] = dict(
        launches=calls[k],
        fetch_size_kb=float(fetch.loc[k, "FETCH_SIZE"]), write_size_kb=float(write.loc[k, "WRITE_SIZE"]),
        gpu_cycles=float(cyc), valu_insts=float(sq1.loc[k, "SQ_INSTS_VALU"]), salu_insts=float(sq1.loc[k, "SQ_INSTS_SALU"]),
        lds_insts=float(sq1.loc[k, "SQ_INSTS_LDS"]), vmem_rd_insts=float(sq1.loc[k, "SQ_INSTS_VMEM_RD"]),
        vmem_wr_insts=float(sq1.loc[k, "SQ_INSTS_VMEM_WR"]), waves=float(sq1.loc[k, "SQ_WAVES"]),
        wave_cycles=float(sq1.loc[k, "SQ_WAVE_CYCLES"]), wait_any=float(sq2.loc[k, "SQ_WAIT_ANY"]),
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs; TA_TA_BUSY_sum cycles summed over the 256 TAs
        valu_busy_frac=float(sq2.loc[k, "SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc),
        ta_busy_frac=float(ta.loc[k, "TA_TA_BUSY_sum"] / 256.0 / cyc),
        l2_hit_rate=float(ta.loc[k, "TCC_HIT_sum"] / (ta.loc[k, "TCC_HIT_sum"] + ta.loc[k, "TCC_MISS_sum"])))
tot = float(sum(calls.values()))
fk = sum(v["fetch_size_kb"] * calls[k] for k, v in kern.items()) / tot
wk = sum(v["write_size_kb"] * calls[k] for k, v in kern.items()) / tot
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
lim = "; ".join("%s: vector ALUs busy %.0f %% of the kernel's cycles, texture addresser %.0f %%" % (k, 100 * v["valu_busy_frac"], 100 * v["ta_busy_frac"])
                for k, v in sorted(kern.items()))
out = {
    # the library these counters were taken on (lva_version()'s source hash, written into the bench line): bench.py fills its
    # `traffic` field from this file only when it runs the same build
    "build_id": bench["library"]["build_id"], "library": bench["library"]["version"],
    "_comment": "HBM traffic and limiter of the dominant kernel on the benchmark shape at the DEFAULT 128 read slots: rocprofv3 --pmc passes "
                "(FETCH_SIZE and WRITE_SIZE separately; SQ and TA sets) over `python3 bench.py --steps 1 --warmup 0 --reads-per-step 128 --pool 128 "
                "--no-cpu-baseline --no-launch-events --no-cross-check`, restricted to the lva_step_lazy kernels (--kernel-include-regex) so that the "
                "passes return in seconds.  A launch runs one instance (anchor / odd step) over all slots: per-launch figures are the mean over "
                "the instances' launches.  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (scattered 8/16-byte gathers are "
                "uncalibrated: the doubled figure is an upper estimate).  raw / corrected traffic = %.2fx / %.2fx the algorithmic bytes of the same "
                "launches (%.2f GB)." % ((fk + wk) * 1024 / alg, (2 * fk + wk) * 1024 / alg, alg / 1e9),
    "kernel": " | ".join(sorted(kern)), "kernel_mode": 4, "config": "mem_conv=11 rate=5 list_size=8 msg_len=180 max_deviation=20",
    "slots": bench["config"]["mean_active_slots"], "nominal_slots": bench["config"]["slots"],
    "fetch_size_kb_per_launch": fk, "write_size_kb_per_launch": wk, "fetch_correction": 2.0,
    "algorithmic_bytes_per_launch": alg, "per_kernel": kern,
    "limiter": "instruction issue and lane-level memory operations, not HBM bytes (%s; HBM traffic %.2fx raw / %.2fx corrected of the algorithmic bytes)"
               % (lim, (fk + wk) * 1024 / alg, (2 * fk + wk) * 1024 / alg),
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("slots", "fetch_size_kb_per_launch", "write_size_kb_per_launch", "limiter")}, indent=1))
