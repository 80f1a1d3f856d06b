#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "ord_split or matrix_parity or fused or 216" 2>&1 | tail -4
for lib in libtbhip_chain.so libtbhip.so libtbhip_chain.so libtbhip.so; do
  echo "== $lib"
  TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/$lib timeout 600 python3 bench.py --no-cpu-baseline --no-slab-sweep --no-dist-cg --steps 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], d['phase_ms'])"
done
} > gpurun_out/gpu_r4_twelfth.log 2>&1
tail -c 3000 gpurun_out/gpu_r4_twelfth.log
