#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel name."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "wedm_step" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print(f"{d}\t{k}\t{c}\t{sum(v)/len(v):.4g}\t(n={len(v)})")
