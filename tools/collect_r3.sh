#!/bin/bash
# Collects the round-3 evidence files into gpurun_out/r3/ (copy what is to be judged into profiles/): bash tools/collect_r3.sh
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; o=gpurun_out/r3; mkdir -p $o
(echo "# tools/_ubench_clock (hipcc --offload-arch=gfx950 -O3 tools/ubench_clock.hip): back-to-back independent v_mfma_f32_32x32x16_bf16, 4 or 8 waves per workgroup, 1 / 32 / 256 workgroups"; tools/_ubench_clock) > $o/r3_clock_microbench.txt 2>&1
(echo "# tools/_ubench_pingpong (hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/ubench_pingpong.hip): cycles per iteration of wave 0; modes in the source header (mode 4's figure is wave 0's idle loop, not the VALU waves')"; tools/_ubench_pingpong) > $o/r3_pingpong_microbench.txt 2>&1
(echo "# python tools/bench_lin.py 30: lean kernels (csrc/linear.hip) vs gemm_kernel, back to back, B = 8"; python tools/bench_lin.py 30 2>&1 | grep -v amdgpu.ids
 for t in abl1 abl2 abl3; do echo "# ablation build tools/_abl/libcountr_$t.so (LIN_ABL: 1 = no MFMA, 2 = no fragment reads, 3 = no DMA after the first two k-tiles; lean column only)"; for f in proj fc2 qkv fc1; do COUNTR_LIB=$PWD/tools/_abl/libcountr_$t.so python tools/bench_lin.py 30 "$f" 2>&1 | grep -v amdgpu.ids | grep -v "dec " | cut -c1-75; done; done
 echo "# python tools/bench_gemm.py 'conv fwd' 20: lean (COUNTR_LEAN_CONV=1) vs generic (=0)"; for l in 1 0; do echo "COUNTR_LEAN_CONV=$l"; COUNTR_LEAN_CONV=$l python tools/bench_gemm.py "conv fwd" 20 2>&1 | grep -v amdgpu; done) > $o/r3_linear_microbench.txt 2>&1
(echo "# COUNTR_LIB=tools/_abl/libcountr_stamp.so python tools/stamp_lin.py  (-DLIN_STAMP build): s_memtime cycles per k-tile"; COUNTR_LIB=$PWD/tools/_abl/libcountr_stamp.so python tools/stamp_lin.py 2>&1 | grep -v amdgpu) > $o/r3_linear_stamps.txt 2>&1
(echo "# python tools/stamp_attn.py (COUNTR_FA_ABL=7)"; python tools/stamp_attn.py 2>&1 | grep -v amdgpu; echo "# COUNTR_FA_ABL=a python tools/bench_attn.py --one, B = 32 (0 full, 1 K/V staged in the prologue only, 2 = 1 + no exp, 3 = 1 + no MFMA, 4 staging + barriers only, 6 one tile, 8 no fragment reads, 9 = 1 + no VALU, 10 = 1 + no row max / rescale)"; export BENCH_ATTN_SHAPES="32,576,12,64"; for a in 0 1 2 3 4 6 8 9 10; do echo -n "ABL=$a "; COUNTR_FA_ABL=$a python tools/bench_attn.py --one 2>&1 | grep -v amdgpu | cut -c1-60; done) > $o/r3_attention_anatomy.txt 2>&1
python bench.py > $o/bench.log 2>&1; tail -1 $o/bench.log > $o/r3_bench_line.json
python bench.py --workload pretrain 2>&1 | tail -1 > $o/r3_bench_pretrain_line.json
python bench.py --workload infer 2>&1 | tail -1 > $o/r3_bench_infer_line.json
bash tools/prof_step.sh r3 > /dev/null 2>&1; cp gpurun_out/r3_step_breakdown.txt gpurun_out/r3_kernel_stats.csv $o/
for cfg in "COUNTR_LEAN=0 COUNTR_LN_FOLD=0" "COUNTR_LEAN_CONV=0 COUNTR_LN_FOLD=0" "COUNTR_LN_FOLD=0" ""; do echo -n "[$cfg] ms/step: "; env $cfg python bench.py --steps 30 --warmup 5 --reps 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; done > $o/r3_step_ab.txt 2>&1
ls -la $o
