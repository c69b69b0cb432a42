cd $GRAFT_REPO_ROOT
cp monoorbslam3_amd/lib/liborbx.so /tmp/orig.so
for n in "$@"; do
  if [ "$n" != "liborbx.so" ]; then cp monoorbslam3_amd/lib/variants/$n monoorbslam3_amd/lib/liborbx.so; else cp /tmp/orig.so monoorbslam3_amd/lib/liborbx.so; fi
  echo "== $n: $(python3 tools/ba_latency.py 2>&1 | grep pose_opt | tail -1)"
done
cp /tmp/orig.so monoorbslam3_amd/lib/liborbx.so
