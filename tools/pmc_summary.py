#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").split("::")[-1]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, ctrs in sorted(acc.items()):
    if not name.startswith("k_"):
        continue
    print(name)
    for c, v in sorted(ctrs.items()):
        print("   %-28s n=%-4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
