#!/bin/bash
# usage: ab_env.sh "<bench arguments>" ...   one bench line per setting, e.g.
#   bash tools/ab_env.sh "" "--variant side_blur=3" "--variant desc=separate --variant side_blur=2"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
i=0
for e in "$@"; do
  i=$((i+1))
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs $e $AB_ARGS > gpurun_out/ab/env$i.json 2> gpurun_out/ab/env$i.err
  python -c "
import json
d=json.load(open('gpurun_out/ab/env$i.json')); print('[$e]', d['value'], d['ms_per_step'], d['stages_ms'], d.get('stages_ms_in_step'))"
done
