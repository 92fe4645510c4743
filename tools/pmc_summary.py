#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel
(full-size dispatches only = those with the most common grid size for that kernel)."""
import csv, sys, collections, glob
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        by = collections.defaultdict(lambda: collections.defaultdict(list))
        grid = collections.defaultdict(collections.Counter)
        for r in rows:
            grid[r["Kernel_Name"]][r["Grid_Size"]] += 1
        for r in rows:
            k = r["Kernel_Name"]
            if r["Grid_Size"] != grid[k].most_common(1)[0][0]:
                continue
            by[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("==", f)
        for k, cs in by.items():
            if "rocclr" in k: continue
            short = k.split("(")[0].replace("void amph::", "").replace("amph::", "")
            print(f"  {short:28s} grid {grid[k].most_common(1)[0][0]:>9s} " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
