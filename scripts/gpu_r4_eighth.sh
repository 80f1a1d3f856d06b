#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "config5" 2>&1 | grep -v "^$" | tail -40
for lib in libtbhip.so libtbhip_cap1024.so libtbhip_cap4096.so; do
  echo "== $lib"
  TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/$lib timeout 600 python3 bench.py --no-cpu-baseline --no-slab-sweep --steps 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  cg', d['distributed_cg']['iteration_ms'], d['distributed_cg']['local_spmv_ms'], d['distributed_cg']['product_halo_dot_ms'])"
done
for w in 2 4 5; do
  echo "== TB_SPMV_WG_PER_CU=$w"
  TB_SPMV_WG_PER_CU=$w timeout 600 python3 bench.py --no-cpu-baseline --no-slab-sweep --steps 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  cg', d['distributed_cg']['iteration_ms'], d['distributed_cg']['local_spmv_ms'], d['distributed_cg']['product_halo_dot_ms'])"
done
} > gpurun_out/gpu_r4_eighth.log 2>&1
tail -c 7000 gpurun_out/gpu_r4_eighth.log
