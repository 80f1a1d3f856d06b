#!/bin/bash
mkdir -p gpurun_out/r5
{
for i in 1 2; do
echo "4 waves (44 B scratch):"; python3 scripts/bench_mechanics.py --n 80 --cpu-n 2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  linearize %.2f ms residual %.2f' % (d['linearize_ms'], d['residual_ms']))"
echo "3 waves:"; TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_g3.so python3 scripts/bench_mechanics.py --n 80 --cpu-n 2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  linearize %.2f ms residual %.2f' % (d['linearize_ms'], d['residual_ms']))"
done
} > gpurun_out/r5/aa.log 2>&1
grep -v amdgpu.ids gpurun_out/r5/aa.log
