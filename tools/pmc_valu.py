#!/usr/bin/env python3
"""Turn a rocprofv3 --pmc SQ_INSTS_VALU pass (counter_collection CSV) into profiles/pmc_valu.json: VALU wave-instructions
per stage per step (all launches of the stage).  Usage: pmc_valu.py <pmc_dir> <batch> <out.json>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from monoorbslam3_amd._lib import kernels_sha16  # noqa: E402  (hash of the kernel sources the counters belong to)

STAGE = {"k_resize": "resize", "k_fast_cells_wave": "fast", "k_fast_strip": "fast", "k_blur_cols": "blur", "k_blur_edges": "blur", "k_blur_mfma": "blur", "k_angle": "orient",
         "k_octree_lds": "octree", "k_octree": "octree", "k_orient": "orient", "k_orient_desc": "desc", "k_blur_desc": "desc", "k_desc_bins": "desc",
         "k_best2": "match_best2", "k_best2_mfma": "match_best2", "k_best2_fp4": "match_best2", "k_resize2": "resize", "k_resize_lds": "resize"}
def kname(full):
    """'void k_blur_mfma<256>(FastSrc, ...)' -> 'k_blur_mfma'"""
    return full.split("(")[0].replace("void ", "").split("<")[0].strip().split("::")[-1]  # (oct_batch::k_octree_lds -> k_octree_lds)


STEP_MARKER = "k_octree_lds"  # exactly one launch per step: per-step totals = sum over all dispatches / its count


def main():
    root, batch, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    acc = defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == "SQ_INSTS_VALU":
                acc[kname(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    res, detail = defaultdict(float), {}
    n_steps = max(len(acc.get(STEP_MARKER, [])), 1)
    for k, v in acc.items():
        if k not in STAGE:
            continue
        per_step = sum(v) / n_steps  # resize: 7 launches per step, FAST and best-2: 2
        res[STAGE[k]] += per_step
        detail[k] = round(per_step)
    json.dump({"batch": batch, "kernels_sha16": kernels_sha16(), "unit": "VALU wave-instructions per stage per step (SQ_INSTS_VALU, all launches)",
               "wave_instr_per_step": {k: int(v) for k, v in res.items()}, "detail": detail}, open(out, "w"), indent=1)
    print(json.dumps(dict(res)))


if __name__ == "__main__":
    main()
