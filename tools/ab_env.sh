#!/bin/bash
# usage: ab_env.sh "VAR=val VAR2=val" ...   one bench line per environment
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
i=0
for e in "$@"; do
  i=$((i+1))
  env $e python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end $AB_ARGS > gpurun_out/ab/env$i.json 2> gpurun_out/ab/env$i.err
  python -c "
import json
d=json.load(open('gpurun_out/ab/env$i.json')); print('$e', d['value'], d['ms_per_step'], d['stages_ms'], d.get('stages_ms_in_step'))"
done
