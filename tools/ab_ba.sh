#!/bin/bash
# usage: ab_ba.sh name...   per-kernel BA times (tools/ba_kernel_stats.sh) with each library under monoorbslam3_amd/lib/variants
# swapped in as liborbx.so ("liborbx.so" = the shipped one); timing experiments only (variants may compute garbage)
cd $GRAFT_REPO_ROOT
cp monoorbslam3_amd/lib/liborbx.so /tmp/orig.so
for n in "$@"; do
  if [ "$n" != "liborbx.so" ]; then cp monoorbslam3_amd/lib/variants/$n monoorbslam3_amd/lib/liborbx.so; else cp /tmp/orig.so monoorbslam3_amd/lib/liborbx.so; fi
  echo "== $n"; bash tools/ba_kernel_stats.sh 2>&1 | grep "^k_lm_chol"
done
cp /tmp/orig.so monoorbslam3_amd/lib/liborbx.so
