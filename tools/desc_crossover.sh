#!/bin/bash
# k_blur_desc (blur + descriptors in one pass) against the blur pass + k_orient_desc over resident batch sizes: where does the
# one-pass kernel start to pay?  (ORBX_FUSED_DESC_MIN_PIXELS in csrc/orbx_api.hip is set from this.)
cd $GRAFT_REPO_ROOT
for b in ${@:-96 128 192 256 384}; do for v in fused separate; do
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs --batch $b --variant desc=$v 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('batch $b desc=$v', round(d['value']), d['ms_per_step'], d['stages_ms'])"
done; done
