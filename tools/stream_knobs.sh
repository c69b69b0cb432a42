#!/bin/bash
# Sweep of the stream-overlap knobs (--variant side_blur x early_fast) at the headline config; prints frames/s per setting.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/knobs
for sb in 0 1 2; do for ef in 0 1 2; do
  python bench.py --variant side_blur=$sb --variant early_fast=$ef --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs > gpurun_out/knobs/k.json 2> gpurun_out/knobs/k.err
  python -c "
import json
d=json.load(open('gpurun_out/knobs/k.json')); print('side_blur=$sb early_fast=$ef', round(d['value']), d['ms_per_step'])"
done; done | tee gpurun_out/knobs/sweep.txt
