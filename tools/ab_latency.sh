#!/bin/bash
# usage: ab_latency.sh name...   single-frame latency (tools/latency_c.sh) with each library under monoorbslam3_amd/lib/variants
# swapped in as liborbx.so ("liborbx.so" = the shipped one)
cd $GRAFT_REPO_ROOT
cp monoorbslam3_amd/lib/liborbx.so /tmp/orig.so
for n in "$@"; do
  if [ "$n" != "liborbx.so" ]; then cp monoorbslam3_amd/lib/variants/$n monoorbslam3_amd/lib/liborbx.so; else cp /tmp/orig.so monoorbslam3_amd/lib/liborbx.so; fi
  echo "== $n"; bash tools/latency_c.sh
done
cp /tmp/orig.so monoorbslam3_amd/lib/liborbx.so
