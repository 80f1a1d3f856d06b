#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r02_f
mkdir -p $out
for f in 0 2 7; do
  TB_PROF_STAMPS=1 TB_BENCH_ORDER=fused TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so TB_PATCH_TILE=5,5,6 TB_DEBUG_FLAGS=$f python3 scripts/bench_assembly.py --n 216 --reps 2 2>&1 | grep -v amdgpu | tail -2 | sed "s/^/flags=$f /" | tee -a $out/stamps_staged.txt
done
TB_PROF_STAMPS=1 TB_BENCH_ORDER=sep TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so TB_PATCH_TILE=5,5,6 TB_DEBUG_FLAGS=0 python3 scripts/bench_assembly.py --n 216 --reps 1 2>&1 | grep -v amdgpu | tail -7 | tee -a $out/stamps_staged.txt
