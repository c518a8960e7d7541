#!/usr/bin/env python3
"""profiles/r3_traffic.json from the counter CSVs of scripts/r3/prof_final.sh (gpurun_out/r3prof/{fetch,write,sq1,sq2,ta}) and
the bench line of the same run: HBM bytes per launch of the dominant kernel(s) at the DEFAULT 64 slots, and what the SQ / TA
counters say holds them back (bench.py copies `limiter` into its roofline object).

    python scripts/r3/make_traffic_json.py gpurun_out/r3prof profiles/r3_traffic.json
"""
import glob
import json
import sys

import pandas as pd

src, dst = sys.argv[1], sys.argv[2]


def counters(name):
    import os
    f = max(glob.glob("%s/%s/*/*counter_collection.csv" % (src, name)), key=os.path.getmtime)   # (the directory collects every run: newest)
    df = pd.read_csv(f)
    df["k"] = df["Kernel_Name"].str.extract(r"(lva_step_lazy(?:_fused)?<[^>]*>)")[0]
    return df.groupby(["k", "Counter_Name"])["Counter_Value"].mean().unstack()


fetch, write, sq1, sq2, ta = (counters(n) for n in ("fetch", "write", "sq1", "sq2", "ta"))
bench = json.loads([ln for ln in open(src + "/r3_lazy64_bench_under_pmc.json") if ln.startswith("{")][-1])
kern = {}
for k in fetch.index:
    cyc = ta.loc[k, "GRBM_GUI_ACTIVE"] / 8.0                     # summed over the 8 XCDs
    kern[k] = dict(
        fetch_size_kb=float(fetch.loc[k, "FETCH_SIZE"]), write_size_kb=float(write.loc[k, "WRITE_SIZE"]),
        gpu_cycles=float(cyc), valu_insts=float(sq1.loc[k, "SQ_INSTS_VALU"]), salu_insts=float(sq1.loc[k, "SQ_INSTS_SALU"]),
        lds_insts=float(sq1.loc[k, "SQ_INSTS_LDS"]), vmem_rd_insts=float(sq1.loc[k, "SQ_INSTS_VMEM_RD"]),
        vmem_wr_insts=float(sq1.loc[k, "SQ_INSTS_VMEM_WR"]),
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs; TA_TA_BUSY_sum cycles summed over the 256 TAs
        valu_busy_frac=float(sq2.loc[k, "SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc),
        ta_busy_frac=float(ta.loc[k, "TA_TA_BUSY_sum"] / 256.0 / cyc),
        l2_hit_rate=float(ta.loc[k, "TCC_HIT_sum"] / (ta.loc[k, "TCC_HIT_sum"] + ta.loc[k, "TCC_MISS_sum"])))
fk = sum(v["fetch_size_kb"] for v in kern.values())
wk = sum(v["write_size_kb"] for v in kern.values())
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
lim = "; ".join("%s: vector ALUs busy %.0f %% of the kernel's cycles, texture addresser %.0f %%" % (k, 100 * v["valu_busy_frac"], 100 * v["ta_busy_frac"])
                for k, v in sorted(kern.items()))
out = {
    "_comment": "HBM traffic and limiter of the dominant kernel(s) on the benchmark shape at the DEFAULT 64 read slots: rocprofv3 --pmc passes "
                "(FETCH_SIZE and WRITE_SIZE separately; SQ and TA sets) over `python3 bench.py --steps 1 --warmup 0 --reads-per-step 64 --pool 64 "
                "--no-cpu-baseline --no-launch-events --no-cross-check`, restricted to the lva_step_lazy kernels (--kernel-include-regex) so that the "
                "passes return in seconds; means per launch.  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (scattered 8/16-byte "
                "gathers are uncalibrated: the doubled figure is an upper estimate).  raw / corrected traffic = %.2fx / %.2fx the algorithmic bytes "
                "of the same launches (%.2f GB)." % ((fk + wk) * 1024 / alg, (2 * fk + wk) * 1024 / alg, alg / 1e9),
    "kernel": " + ".join(sorted(kern)), "kernel_mode": 4, "config": "mem_conv=11 rate=5 list_size=8 msg_len=180 max_deviation=20",
    "slots": bench["config"]["mean_active_slots"], "nominal_slots": bench["config"]["slots"],
    "fetch_size_kb_per_launch": fk, "write_size_kb_per_launch": wk, "fetch_correction": 2.0,
    "algorithmic_bytes_per_launch": alg, "per_kernel": kern,
    "limiter": "instruction issue and lane-level memory operations, not HBM bytes (%s; HBM traffic %.2fx raw / %.2fx corrected of the algorithmic bytes)"
               % (lim, (fk + wk) * 1024 / alg, (2 * fk + wk) * 1024 / alg),
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("slots", "fetch_size_kb_per_launch", "write_size_kb_per_launch", "limiter")}, indent=1))
