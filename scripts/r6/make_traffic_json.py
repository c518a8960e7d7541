#!/usr/bin/env python3
"""profiles/r6_traffic.json: LINE traffic per launch of the dominant kernels -- the lazy pair of the benchmark AND the big-list
kernel of configs[4] -- from the counter summaries of the library that is shipped (taken once, on the final build, by
scripts/r6/prof_final2.sh: separate --pmc passes), in the two forms MI355X_MICROARCH.md's HBM section asks for:

    fetch_x2_plus_write   FETCH_SIZE x 2 (the gfx950 correction) + WRITE_SIZE, bytes
    tcc_miss_x128         TCC_MISS_sum x 128 B: every L2 miss is a line whatever part of it the lane wanted

and their ratio to the algorithmic bytes of the same launches.  bench.py fills roofline.traffic (headline) and
extra_configs[].traffic (configs[4]) from this file when the running library has the same build id.

    python scripts/r6/make_traffic_json.py <dir> profiles/r6_traffic.json [prefix of the files in <dir>, default r5]
"""
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
PRE = sys.argv[3] if len(sys.argv) > 3 else "r5"


def summary(path):
    """{kernel prefix: {counter: mean per dispatch}} from a scripts/pmc_summary.py text"""
    out, k = {}, None
    for ln in open(path):
        m = re.match(r"^(void lva::\S.*?)\s+([A-Z][A-Za-z_0-9]+)\s+([0-9.e+]+)\s+([0-9.e+]+)\s+(\d+)\s*$", ln)
        if m:
            k = re.search(r"lva_step_[a-z_]+<[^>(]*>?", m.group(1)).group(0)
            k = re.sub(r"\s", "", k)
            out.setdefault(k, {})[m.group(2)] = (float(m.group(3)), int(m.group(5)))
            continue
        m = re.match(r"^\s+([A-Z][A-Za-z_0-9]+)\s+([0-9.e+]+)\s+([0-9.e+]+)\s+(\d+)\s*$", ln)
        if m and k:
            out[k][m.group(1)] = (float(m.group(2)), int(m.group(4)))
    return out


def bench(path):
    return json.loads([ln for ln in open(path) if ln.startswith("{")][-1])


def entry(counters, alg_per_launch, launches_weight=None):
    """counters: {kernel: {counter: (mean, count)}} of the kernels that make up `a launch` (weighted by their dispatch counts)"""
    tot = sum(c["FETCH_SIZE"][1] for c in counters.values())
    w = {k: c["FETCH_SIZE"][1] / tot for k, c in counters.items()}
    fetch = sum(w[k] * c["FETCH_SIZE"][0] for k, c in counters.items()) * 1024.0      # FETCH_SIZE / WRITE_SIZE are in KB
    write = sum(w[k] * c["WRITE_SIZE"][0] for k, c in counters.items()) * 1024.0
    miss = sum(w[k] * c["TCC_MISS_sum"][0] for k, c in counters.items())
    hit = sum(w[k] * c["TCC_HIT_sum"][0] for k, c in counters.items())
    return {"fetch_size_bytes": fetch, "write_size_bytes": write, "fetch_x2_plus_write": 2 * fetch + write, "tcc_miss": miss, "tcc_hit": hit,
            "tcc_miss_x128": miss * 128.0, "algorithmic_bytes_per_launch": alg_per_launch,
            "ratio_fetch_x2_plus_write": (2 * fetch + write) / alg_per_launch, "ratio_tcc_miss_x128": miss * 128.0 / alg_per_launch,
            "ratio_raw_counters": (fetch + write) / alg_per_launch, "dispatch_weights": w}


lazy = {k: v for k, v in summary(src + "/%s_lazy128_pmc_summary.txt" % PRE).items() if k.startswith("lva_step_lazy")}
big = {k: v for k, v in summary(src + "/%s_big64_pmc_summary.txt" % PRE).items() if k.startswith("lva_step_big_rec")}
bl, bb = bench(src + "/%s_lazy128_bench_under_pmc.json" % PRE), bench(src + "/%s_big64_bench_under_pmc.json" % PRE)
assert bl["library"]["build_id"] == bb["library"]["build_id"]
r5 = json.load(open(src + "/%s_lazy_traffic.json" % PRE if PRE != "r5" else src + "/r5_traffic.json"))
out = {
    "build_id": bl["library"]["build_id"], "library": bl["library"]["version"],
    "_comment": "line traffic per launch of the dominant kernels on the shipped library (counters of separate rocprofv3 --pmc passes on the build "
                "named here): FETCH_SIZE x 2 + WRITE_SIZE and TCC_MISS x 128 B, both against the algorithmic bytes of the same launches.  "
                "The lazy pair moves FEWER lines than the algorithm counts (no wasted traffic: it is bound by instruction issue and lane-level memory "
                "operations); the big-list kernel moves 1.7-1.9x the algorithmic lines at ~5 TB/s: it IS line-bound, with about half of the lines wasted "
                "(every record line fetched twice -- walk, then output -- and partial-line writes).",
    "kernel_mode": 4, "config": "mem_conv=11 rate=5 list_size=8 msg_len=180 max_deviation=20",
    # the fields bench.py's headline roofline reads (as in r5_traffic.json)
    "kernel": r5["kernel"], "slots": r5["slots"], "nominal_slots": r5["nominal_slots"], "fetch_size_kb_per_launch": r5["fetch_size_kb_per_launch"],
    "write_size_kb_per_launch": r5["write_size_kb_per_launch"], "fetch_correction": 2.0, "limiter": r5["limiter"],
    "kernels": {
        "lazy_pair": dict(entry(lazy, bl["roofline"]["algorithmic_bytes_per_launch"]), kernel=" | ".join(sorted(lazy)),
                          slots=bl["config"]["mean_active_slots"], nominal_slots=bl["config"]["slots"], workload="configs[1]"),
        "big_rec_64": dict(entry(big, bb["roofline"]["algorithmic_bytes_per_launch"]), kernel=" | ".join(sorted(big)),
                           slots=bb["config"]["mean_active_slots"], nominal_slots=bb["config"]["slots"], workload="configs[4]",
                           list_size=64),
    },
}
json.dump(out, open(dst, "w"), indent=1)
for k, v in out["kernels"].items():
    print("%-11s FETCHx2+WRITE %.2f GB = %.2fx   TCC_MISS x128 %.2f GB = %.2fx   algorithmic %.2f GB   (raw counters %.2fx)"
          % (k, v["fetch_x2_plus_write"] / 1e9, v["ratio_fetch_x2_plus_write"], v["tcc_miss_x128"] / 1e9, v["ratio_tcc_miss_x128"],
             v["algorithmic_bytes_per_launch"] / 1e9, v["ratio_raw_counters"]))
