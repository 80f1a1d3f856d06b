#!/bin/bash
mkdir -p gpurun_out/r5
{
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -q -x -k "hill or Hill or energies or mechanics or hyperelastic or prestress or newton" 2>&1 | tail -3
for h in extended generalized; do
echo "== $h, per-pairing instance"; python3 scripts/bench_mechanics.py --n 40 --hill $h --cpu-n 2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  linearize %.2f ms residual %.2f' % (d['linearize_ms'], d['residual_ms']))"
echo "== $h, run-time form (previous)"; TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_prev.so python3 scripts/bench_mechanics.py --n 40 --hill $h --cpu-n 2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  linearize %.2f ms residual %.2f' % (d['linearize_ms'], d['residual_ms']))"
done
} > gpurun_out/r5/cc.log 2>&1
grep -v amdgpu.ids gpurun_out/r5/cc.log
