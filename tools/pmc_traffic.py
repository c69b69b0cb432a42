#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; counter_collection CSVs) into profiles/pmc_traffic.json:
HBM bytes per launch per stage.  Counters are in KB.  FETCH_SIZE is doubled: on gfx950 it reports exactly 1/2 of the
bytes read (MI355X_MICROARCH.md, HBM section, for 16 B/lane streams), and tools/microbench/fetch_calib.hip confirms the
same factor 0.500 for the 8 B/lane and 4 B/lane reads these kernels use (profiles/r01_fetch_calibration.txt).
WRITE_SIZE is exact.
Usage: pmc_traffic.py <fetch_dir> <write_dir> <batch> <out.json>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from monoorbslam3_amd._lib import kernels_sha16  # noqa: E402  (hash of the kernel sources the counters belong to)

STAGE = {"k_resize": "resize", "k_fast_cells_wave": "fast", "k_fast_strip": "fast", "k_blur_cols": "blur", "k_blur_edges": "blur", "k_blur_mfma": "blur", "k_angle": "orient", "k_octree_lds": "octree",
         "k_orient": "orient", "k_orient_desc": "desc", "k_blur_desc": "desc", "k_desc_bins": "desc", "k_best2": "match_best2", "k_best2_mfma": "match_best2", "k_best2_fp4": "match_best2", "k_resize2": "resize", "k_resize_lds": "resize"}
# per-step totals = sum over all dispatches of a kernel / number of steps in the run; a step has exactly one quadtree launch
def kname(full):
    """'void k_blur_mfma<256>(FastSrc, ...)' -> 'k_blur_mfma'"""
    return full.split("(")[0].replace("void ", "").split("<")[0].strip().split("::")[-1]  # (oct_batch::k_octree_lds -> k_octree_lds)


STEP_MARKER = "k_octree_lds"


def per_kernel(root):
    acc = defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[kname(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return acc


def main():
    fetch, write, batch, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    res = defaultdict(float)
    detail = {}
    for label, root in (("fetch", fetch), ("write", write)):
        acc = per_kernel(root)
        n_steps = max(len(acc.get(STEP_MARKER, [])), 1)
        for k, v in acc.items():
            if k not in STAGE:
                continue
            per_step = sum(v) / n_steps  # resize: 7 launches, FAST: 2 (level 0 early + the rest), best-2: 2 per step
            res[STAGE[k]] += per_step * 1024 * (2 if label == "fetch" else 1)
            detail["%s.%s_KB" % (k, label)] = round(per_step, 1)
    json.dump({"batch": batch, "kernels_sha16": kernels_sha16(), "unit": "bytes per stage per step (all launches of the stage); fetch = 2 x FETCH_SIZE",
               "bytes_per_launch": {k: int(v) for k, v in res.items()}, "detail": detail}, open(out, "w"), indent=1)
    print(json.dumps({k: int(v) for k, v in res.items()}))


if __name__ == "__main__":
    main()
