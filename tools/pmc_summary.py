#!/usr/bin/env python3
"""Average PMC counters per kernel from rocprofv3 counter_collection.csv files:
   python tools/pmc_summary.py gpurun_out/pmc_slab k_spgemm_slab"""
import csv
import glob
import sys
from collections import defaultdict

root, pat = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(root + "/p*/run_counter_collection.csv")):
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[1:] if len(v) > 1 else v   # drop the first (cold) launch
        print("%-34s n=%2d mean=%.4g" % (k, len(v), sum(v) / len(v)))
