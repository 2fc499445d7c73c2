#!/bin/bash
# Ablation variants of the attention forward (COUNTR_FA_ABL, see flash_attn.hip) built into tools/_abl/ and timed with
# tools/bench_attn.py (run on the GPU box: bash tools/ablate_attn.sh).  Results of the ablated builds are wrong by design.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_abl
for n in ${ABLS:-1 2 3 4 5 6 7}; do
  if [ ! -f tools/_abl/libcountr_fa$n.so ]; then
    objs=""
    for f in api attention elementwise gemm mae norm; do objs="$objs countr_amd/build/$f.hip.o"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -DCOUNTR_FA_ABL=$n -c countr_amd/csrc/flash_attn.hip -o tools/_abl/fa_abl$n.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_abl/libcountr_fa$n.so $objs tools/_abl/fa_abl$n.o
  fi
done
echo "== baseline"; python tools/bench_attn.py 2>&1 | grep "^B"
for n in ${ABLS:-1 2 3 4 5 6 7}; do echo "== COUNTR_FA_ABL=$n"; COUNTR_LIB=$PWD/tools/_abl/libcountr_fa$n.so python tools/bench_attn.py 2>&1 | grep "^B"; done
