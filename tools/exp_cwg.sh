#!/bin/bash
# Ablation builds of the lean conv-wgrad kernel (run HERE, then gpurun `python tools/bench_wgrad.py` with COUNTR_LIB=<variant>):
#   bash tools/exp_cwg.sh   -> countr_amd/build/libcountr_cwg_abl{1,2,3}.so
set -e
cd "$(dirname "$0")/.."
for n in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -DCWG_ABL=$n -c countr_amd/csrc/conv_wgrad.hip -o /tmp/cwg_abl$n.o &
done
wait
for n in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o countr_amd/build/libcountr_cwg_abl$n.so $(ls countr_amd/build/*.hip.o | grep -v conv_wgrad.hip.o) /tmp/cwg_abl$n.o
done
ls -la countr_amd/build/libcountr_cwg_abl*.so
