#!/bin/bash
mkdir -p gpurun_out
{
for lib in libtbhip.so libtbhip_lb4.so libtbhip.so libtbhip_lb4.so; do
  echo "== $lib wave"
  TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/$lib TB_SPMV_KERNEL=wave timeout 600 python3 bench.py --no-cpu-baseline --no-slab-sweep --steps 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  cg', d['distributed_cg']['iteration_ms'], d['distributed_cg']['local_spmv_ms'], d['distributed_cg']['product_halo_dot_ms'])"
done
echo "== 27 layers sig vs wave"
for k in sig wave; do
TB_SPMV_KERNEL=$k timeout 600 python3 bench.py --no-cpu-baseline --layers 27 --steps 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  $k cg', d['distributed_cg']['iteration_ms'], d['distributed_cg']['local_spmv_ms'], d['distributed_cg']['product_halo_dot_ms'], 'step', d['ms_per_step'])"
done
} > gpurun_out/gpu_r4_14.log 2>&1
tail -c 3000 gpurun_out/gpu_r4_14.log
