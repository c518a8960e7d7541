#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, per counter, mean per dispatch."""
import glob
import sys

import pandas as pd

for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        df = pd.read_csv(f)
        df["k"] = df["Kernel_Name"].str.slice(0, 40)
        g = df.groupby(["k", "Counter_Name"])["Counter_Value"].agg(["mean", "sum", "count"])
        print(f)
        print(g.to_string())
