#!/bin/bash
# build a liborbx variant with extra -D flags into monoorbslam3_amd/lib/variants/<name>:
#   VARIANT_FILES="orbx_kernels" tools/build_variant.sh r16.so -DOCT_REG=16
# recompiles the sources named in VARIANT_FILES with the flags and links them with the objects of the regular build
# (build/obj, so run `make -C monoorbslam3_amd/csrc` first); tools/ab_libs.sh / ab_latency.sh then bench the variants
set -e
cd "$(dirname "$0")/../monoorbslam3_amd/csrc"
mkdir -p ../lib/variants
name=$1; shift
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-result --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1"
objs=""
for f in orbx_kernels orbx_api orbm_matcher orbba orbf_frame orbv_vocab orbd_dist; do
  if [[ " $VARIANT_FILES " == *" $f "* ]]; then
    /opt/rocm/bin/hipcc $FLAGS "$@" -c -o /tmp/var_${name}_$f.o $f.hip &
    objs="$objs /tmp/var_${name}_$f.o"
  else
    objs="$objs ../../build/obj/$f.o"
  fi
done
wait
/opt/rocm/bin/hipcc -fPIC --offload-arch=gfx950 -shared -o ../lib/variants/$name $objs
echo built $name
