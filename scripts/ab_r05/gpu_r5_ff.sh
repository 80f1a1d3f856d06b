#!/bin/bash
mkdir -p gpurun_out/r5
export TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
{
for t in 8,8,8 16,8,4 16,4,8 16,8,8 32,4,4 16,6,6 12,8,6 8,8,8; do TB_VPATCH_TILE=$t python3 scripts/slab_kernels.py 216 | grep -v amdgpu; done
} > gpurun_out/r5/ff.log 2>&1
cat gpurun_out/r5/ff.log
