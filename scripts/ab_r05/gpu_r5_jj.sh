#!/bin/bash
mkdir -p gpurun_out/r5
export TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
{
for L in 54 108; do for b in 16 2 4 16 2; do TB_REACTION_BLOCKS_PER_CU=$b python3 scripts/slab_kernels.py $L | grep -v amdgpu; done; done
} > gpurun_out/r5/jj2.log 2>&1
grep reaction gpurun_out/r5/jj2.log | sed 's/.*cells \([0-9]*\) .*reaction \([0-9.]*\) ms (\([0-9.]*\) ps.*env \(.*\)/cells \1 reaction \2 ms \3 ps\/dof \4/'
