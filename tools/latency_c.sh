#!/bin/bash
# C++ caller's latency of orbx_extract at the three BASELINE shapes (build: make -C tools latency_c, or the g++ line below).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
[ -x tools/bin/latency_c ] || g++ -O2 -std=c++17 tools/latency_c.cpp -o tools/bin/latency_c -Lmonoorbslam3_amd/lib -lorbx -Wl,-rpath,'$ORIGIN/../../monoorbslam3_amd/lib'
for shape in "1242 375 2000" "752 480 1000" "1920 1080 2000"; do
  set -- $shape
  python3 -c "
import sys; sys.path.insert(0, '.')
from monoorbslam3_amd import synth
synth.make_frames(1, $1, $2)[0].tofile('/tmp/frame_$1x$2.bin')"
  tools/bin/latency_c /tmp/frame_$1x$2.bin $1 $2 $3
done
# the same with the frame in a page-locked buffer (orbx_host_register): the copy no longer blocks the calling thread
for shape in "1242 375 2000" "752 480 1000" "1920 1080 2000"; do
  set -- $shape
  tools/bin/latency_c /tmp/frame_$1x$2.bin $1 $2 $3 200 pinned
done
