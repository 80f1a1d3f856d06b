#!/bin/bash
mkdir -p gpurun_out/r5
{
for i in 1 2 3; do
timeout 600 python3 bench.py --no-cpu-baseline --no-graph --steps 5 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['distributed_cg']; print('cg iteration %.4f ms product+dot %.4f local spmv %.4f' % (c['iteration_ms'], c['product_halo_dot_ms'], c['local_spmv_ms']), {L:round(v['cg_iteration_ms'],4) for L,v in d['slab_sweep']['layers'].items()})"
done
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_distributed_gpu.py -q -x -k "cg or CG or heat or distributed or graph" 2>&1 | tail -2
} > gpurun_out/r5/ee.log 2>&1
grep -v amdgpu.ids gpurun_out/r5/ee.log
