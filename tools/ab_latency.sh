#!/bin/bash
# usage: ab_latency.sh name...   single-frame C-caller latency with each named library (under monoorbslam3_amd/lib) swapped in
cd $GRAFT_REPO_ROOT
python3 -c "
import sys; sys.path.insert(0, '.')
from monoorbslam3_amd import synth
synth.make_frames(1, 1242, 375)[0].tofile('/tmp/frame_k.bin')"
cp monoorbslam3_amd/lib/liborbx.so /tmp/orig.so
for round in 1 2 3 4 5 6; do
for n in "$@"; do
  if [ "$n" != "liborbx.so" ]; then cp monoorbslam3_amd/lib/$n monoorbslam3_amd/lib/liborbx.so; else cp /tmp/orig.so monoorbslam3_amd/lib/liborbx.so; fi
  echo "$n: $(tools/bin/latency_c /tmp/frame_k.bin 1242 375 2000 500)"
done; done
cp /tmp/orig.so monoorbslam3_amd/lib/liborbx.so
