#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "spmv or cg or heat" 2>&1 | tail -4
for k in sig wave sig wave; do
  echo "== TB_SPMV_KERNEL=$k"
  TB_SPMV_KERNEL=$k timeout 600 python3 bench.py --no-cpu-baseline --no-slab-sweep --steps 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  cg', d['distributed_cg']['iteration_ms'], d['distributed_cg']['local_spmv_ms'], d['distributed_cg']['product_halo_dot_ms'])"
done
} > gpurun_out/gpu_r4_13.log 2>&1
tail -c 3000 gpurun_out/gpu_r4_13.log
