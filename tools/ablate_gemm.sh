#!/bin/bash
# Builds ablation variants of gemm.hip (COUNTR_ABL = 1: no MFMA, 2: no fragment reads, 3: DMA only for the first tile) into
# tools/_abl/libcountr_abl<N>.so and times tools/bench_gemm.py with each (run on the GPU box: bash tools/ablate_gemm.sh [filter]).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_abl
for n in ${ABLS:-1 2 3}; do
  if [ ! -f tools/_abl/libcountr_abl$n.so ]; then
    objs=""
    for f in api attention elementwise flash_attn mae norm; do objs="$objs countr_amd/build/$f.hip.o"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -DCOUNTR_ABL=$n -c countr_amd/csrc/gemm.hip -o tools/_abl/gemm_abl$n.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_abl/libcountr_abl$n.so $objs tools/_abl/gemm_abl$n.o
  fi
done
echo "== baseline"; python tools/bench_gemm.py "$1" 30 2>&1 | grep -v amdgpu.ids
for n in ${ABLS:-1 2 3}; do echo "== COUNTR_ABL=$n"; COUNTR_LIB=$PWD/tools/_abl/libcountr_abl$n.so python tools/bench_gemm.py "$1" 30 2>&1 | grep -v amdgpu.ids; done
