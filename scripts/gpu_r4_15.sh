#!/bin/bash
mkdir -p gpurun_out
{
for k in sig wave sig wave; do
  echo "== TB_SPMV_KERNEL=$k (wave = two runs ahead, 3 waves/SIMD)"
  TB_SPMV_KERNEL=$k timeout 600 python3 bench.py --no-cpu-baseline --no-slab-sweep --steps 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  cg', d['distributed_cg']['iteration_ms'], d['distributed_cg']['local_spmv_ms'], d['distributed_cg']['product_halo_dot_ms'])"
done
} > gpurun_out/gpu_r4_15.log 2>&1
tail -c 3000 gpurun_out/gpu_r4_15.log
