#!/bin/bash
# The two-thread latency table (profiles/r06_two_thread_latency.txt): tests/cpp/two_threads.cpp --latency against the shipped
# library and against every library named on the command line (files under monoorbslam3_amd/lib/variants, e.g. r05_head.so =
# round 5's head: blocking handle streams, the BA entry points on stream 0).  Run on the GPU box:
#   tools/two_thread_latency.sh [variant.so ...] > gpurun_out/two_thread_latency.txt
# A variant older than round 6 has no re-entrant orbv_transform: it is run with --own-voc (one vocabulary handle per thread).
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROCM=${ROCM_PATH:-/opt/rocm}
N=${N:-400}
mkdir -p /tmp/tt
g++ -std=c++17 -O1 -D__HIP_PLATFORM_AMD__ -I $ROCM/include -I include tests/cpp/two_threads.cpp -o /tmp/tt/two_threads \
    -L monoorbslam3_amd/lib -lorbx -L $ROCM/lib -lamdhip64 -pthread
shipped() {
  echo "== shipped liborbx.so (round 6: non-blocking handle streams, BA on leased streams), $N iterations"
  LD_LIBRARY_PATH=$PWD/monoorbslam3_amd/lib:$LD_LIBRARY_PATH /tmp/tt/two_threads --latency $N
}
[ -n "$SHIPPED_LAST" ] || shipped
for v in "$@"; do
  mkdir -p /tmp/tt/$v && cp monoorbslam3_amd/lib/variants/$v /tmp/tt/$v/liborbx.so
  echo
  echo "== variant $v, $N iterations (--own-voc)"
  LD_LIBRARY_PATH=/tmp/tt/$v:$LD_LIBRARY_PATH /tmp/tt/two_threads --latency $N --own-voc
done
[ -z "$SHIPPED_LAST" ] || { echo; shipped; }
