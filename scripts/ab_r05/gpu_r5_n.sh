#!/bin/bash
mkdir -p gpurun_out/r5
{
for i in 1 2; do
python3 scripts/slab_kernels.py 216
TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_vec4.so python3 scripts/slab_kernels.py 216
done
TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_vec4.so timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "source_vector_parity or config1" 2>&1 | tail -2
} > gpurun_out/r5/n.log 2>&1
cat gpurun_out/r5/n.log | grep -v amdgpu.ids
