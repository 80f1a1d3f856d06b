#!/bin/bash
mkdir -p gpurun_out/r5
{
python3 scripts/slab_kernels.py 27
python3 scripts/slab_kernels.py 216
export TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
for b in 0 2 4 6 8 12 16 24 32; do TB_REACTION_BLOCKS_PER_CU=$b python3 scripts/slab_kernels.py 27; done
for b in 0 2 8 32; do TB_REACTION_BLOCKS_PER_CU=$b python3 scripts/slab_kernels.py 216; done
for t in 8,8,4 8,8,7 8,8,9 8,8,14 6,6,7 8,4,7 4,4,7; do TB_VPATCH_TILE=$t python3 scripts/slab_kernels.py 27; done
for t in 8,8,4 6,6,6 8,8,6; do TB_VPATCH_TILE=$t python3 scripts/slab_kernels.py 216; done
} > gpurun_out/r5/j.log 2>&1
cat gpurun_out/r5/j.log | grep -v amdgpu.ids
