#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; counter_collection CSVs) into profiles/pmc_traffic.json:
HBM bytes per launch per stage (KB counters * 1024; FETCH_SIZE is NOT doubled: these kernels read with <= 8-byte
lanes, for which the gfx950 half-count correction is uncalibrated -- see MI355X_MICROARCH.md, HBM section).
Usage: pmc_traffic.py <fetch_dir> <write_dir> <batch> <out.json>"""
import csv
import glob
import json
import sys
from collections import defaultdict

STAGE = {"k_resize": "resize", "k_fast_cells_wave": "fast", "k_blur_cols": "blur", "k_octree_lds": "octree",
         "k_orient": "orient_desc", "k_orient_desc": "orient_desc", "k_best2": "match_best2"}
LAUNCHES_PER_STEP = {"k_resize": 7}


def per_kernel(root):
    acc = defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    return acc


def main():
    fetch, write, batch, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    res = defaultdict(float)
    detail = {}
    for label, root in (("fetch", fetch), ("write", write)):
        for k, v in per_kernel(root).items():
            if k not in STAGE:
                continue
            if k == "k_best2":  # bench launches a (B-1)-pair and a 1-pair problem: sum both
                per_step = sum(v) / (len(v) / 2)
            else:
                per_step = sum(v) / len(v) * LAUNCHES_PER_STEP.get(k, 1)
            res[STAGE[k]] += per_step * 1024
            detail["%s.%s_KB" % (k, label)] = round(per_step, 1)
    json.dump({"batch": batch, "unit": "bytes per stage per step (all launches of the stage)",
               "bytes_per_launch": {k: int(v) for k, v in res.items()}, "detail": detail}, open(out, "w"), indent=1)
    print(json.dumps({k: int(v) for k, v in res.items()}))


if __name__ == "__main__":
    main()
