#!/usr/bin/env python3
"""Mean duration of the two lva_step_lazy instances in windows of 4000 launches over a whole 10 000-read job (rocprofv3 kernel trace),
with the clock / power samples of the same seconds (gpurun_out/r5job/smi.log)."""
import glob
import sys

import pandas as pd

f = max(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"), key=lambda p: 0)
df = pd.read_csv(f)
df = df[df["Kernel_Name"].str.contains("lva_step_lazy")].sort_values("Start_Timestamp").reset_index(drop=True)
df["dur_us"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
df["anchor"] = df["Kernel_Name"].str.contains("true")
t0 = df["Start_Timestamp"].iloc[0]
print("launches %d, job span %.1f s" % (len(df), (df["End_Timestamp"].iloc[-1] - t0) / 1e9))
print("%8s %10s %12s %12s %12s" % ("window", "t [s]", "anchor us", "odd us", "mean us"))
W = 4000
for k in range(0, len(df), W):
    w = df.iloc[k:k + W]
    if len(w) < W // 2:
        break
    print("%8d %10.1f %12.1f %12.1f %12.1f" % (k // W, (w["Start_Timestamp"].iloc[0] - t0) / 1e9, w[w.anchor]["dur_us"].mean(),
                                                w[~w.anchor]["dur_us"].mean(), w["dur_us"].mean()))
try:
    lines = [ln.strip() for ln in open(sys.argv[1] + "/../smi.log") if "sclk" in ln]
    print("clock / power samples (every 5 s): first, middle, last")
    for ln in (lines[1], lines[len(lines) // 2], lines[-2]):
        print("  " + ln[:220])
except Exception as e:
    print("no smi samples:", e)
