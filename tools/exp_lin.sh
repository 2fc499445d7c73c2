#!/bin/bash
# Build an experimental variant of linear.hip into tools/_abl/libcountr_<tag>.so: bash tools/exp_lin.sh <tag> <extra hipcc flags...>
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p tools/_abl
objs=""
for f in api attention elementwise flash_attn flash_attn_fwd mae norm gemm; do objs="$objs countr_amd/build/$f.hip.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c countr_amd/csrc/linear.hip -o tools/_abl/linear_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_abl/libcountr_$tag.so $objs tools/_abl/linear_$tag.o
echo tools/_abl/libcountr_$tag.so
