#!/bin/bash
mkdir -p gpurun_out/r5
{
for L in 216 27 54 216 27; do python3 scripts/slab_kernels.py $L | grep -v amdgpu; done
python3 scripts/bench_reaction.py 2>/dev/null | tail -3 | cut -c1-600
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "reaction or rush or single_cell or graph_replay or monodomain" 2>&1 | tail -2
} > gpurun_out/r5/kk.log 2>&1
grep -v amdgpu gpurun_out/r5/kk.log | sed 's/.*cells \([0-9]*\) .*reaction \([0-9.]*\) ms (\([0-9.]*\) ps.*env \(.*\)/cells \1 reaction \2 ms \3 ps\/dof \4/'
