#!/bin/bash
mkdir -p gpurun_out/r5
export TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
{
for g in 0 49152 1000000 0 49152 1000000; do
TB_SPMV_MIRROR_GRID=$g timeout 600 python3 bench.py --no-cpu-baseline --no-graph --steps 5 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['distributed_cg']; print('grid $g: cg iteration %.4f ms product+dot %.4f local spmv %.4f' % (c['iteration_ms'], c['product_halo_dot_ms'], c['local_spmv_ms']), {L:round(v['cg_iteration_ms'],4) for L,v in d['slab_sweep']['layers'].items()})"
done
} > gpurun_out/r5/ee.log 2>&1
grep -v amdgpu.ids gpurun_out/r5/ee.log
