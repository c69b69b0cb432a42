#!/bin/bash
# build liborbx variants with different -D flags into monoorbslam3_amd/lib/variants/<name>.so
set -e
cd /root/repo/monoorbslam3_amd/csrc
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
