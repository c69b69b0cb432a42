#!/bin/bash
# usage: ab_libs.sh name...   (each name: a library file under monoorbslam3_amd/lib/variants to be swapped in as
# liborbx.so on the GPU box's scratch copy; "liborbx.so" = the shipped one).  One line per variant.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
cp monoorbslam3_amd/lib/liborbx.so /tmp/orig.so
for n in "$@"; do
  if [ "$n" != "liborbx.so" ]; then cp monoorbslam3_amd/lib/variants/$n monoorbslam3_amd/lib/liborbx.so; else cp /tmp/orig.so monoorbslam3_amd/lib/liborbx.so; fi
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs $AB_ARGS > gpurun_out/ab/$n.json 2> gpurun_out/ab/$n.err
  python -c "
import json,sys
d=json.load(open('gpurun_out/ab/$n.json')); print('$n', d['value'], d['ms_per_step'], d['stages_ms'])"
done
cp /tmp/orig.so monoorbslam3_amd/lib/liborbx.so
