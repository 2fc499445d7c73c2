#!/bin/bash
# Build an experimental variant of ONE csrc file into tools/_abl/libcountr_<tag>.so: bash tools/exp_file.sh <file (no .hip)> <tag> <extra hipcc flags...>
set -e
cd "$(dirname "$0")/.."
file=$1; tag=$2; shift; shift
mkdir -p tools/_abl
objs=""
for f in countr_amd/csrc/*.hip; do b=$(basename $f .hip); if [ "$b" != "$file" ]; then objs="$objs countr_amd/build/$b.hip.o"; fi; done
extra=""
if [ "$file" = "flash_attn_fwd" ]; then extra="-fno-slp-vectorize -fno-honor-nans"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 $extra "$@" -c countr_amd/csrc/$file.hip -o tools/_abl/${file}_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_abl/libcountr_$tag.so $objs tools/_abl/${file}_$tag.o
echo tools/_abl/libcountr_$tag.so
