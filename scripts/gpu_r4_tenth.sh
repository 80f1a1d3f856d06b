#!/bin/bash
mkdir -p gpurun_out
{
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_distributed_gpu.py tests/test_rccl_world1.py -q -x -k "matrix_parity or fused or pair or 216 or error_codes or variants or config or two_ranks or rccl or bench or element_strategy or deferred" 2>&1 | tail -6
for iso in 0 1; do
  echo "== TB_PATCH_ISO=$iso"
  TB_PATCH_ISO=$iso timeout 600 python3 bench.py --no-cpu-baseline --no-slab-sweep --steps 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], d['phase_ms'], 'cg', round(d['distributed_cg']['iteration_ms'],4))"
done
echo "== generic-energy Q2 tangent (guccione), 40^3 and HO for comparison"
timeout 900 python3 scripts/bench_mechanics.py --n 40 --energy guccione 2>&1 | tail -1 | cut -c1-400
timeout 900 python3 scripts/bench_mechanics.py --n 40 --cpu-n 2 2>&1 | tail -1 | cut -c1-300
} > gpurun_out/gpu_r4_tenth.log 2>&1
tail -c 5000 gpurun_out/gpu_r4_tenth.log
