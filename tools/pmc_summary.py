#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel (optionally per launch geometry) mean counter values.

    python tools/pmc_summary.py gpurun_out/pmc_*/**/*counter_collection.csv
"""
import csv
import glob
import sys
from collections import defaultdict

rows = defaultdict(lambda: defaultdict(list))
for pat in sys.argv[1:]:
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("bn::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if name.startswith("void at::") or "at::native" in name:
                continue
            key = (name[:60], r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
            rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(rows):
    vals = {c: sum(v) / len(v) for c, v in rows[key].items()}
    n = max(len(v) for v in rows[key].values())
    print(f"{key[0]:60s} grid={key[1]:>10s} lds={key[2]:>6s} n={n}")
    print("    " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(vals.items())))
