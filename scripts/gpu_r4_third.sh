#!/bin/bash
# round 4: record (one-trip) patch kernel A/B + cached-signature SpMV
mkdir -p gpurun_out
{
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "spmv or matrix_parity or fused or pair or 216 or error_codes or deferred" 2>&1 | tail -6
for v in "staged:0" "record:0" "record:512" "record:1024" "record:2048"; do
  k=${v%%:*}; pf=${v##*:}
  echo "== TB_PATCH_KERNEL=$k prefetch=$pf"
  TB_PATCH_KERNEL=$k TB_PATCH_PREFETCH=$pf timeout 600 python3 bench.py --no-cpu-baseline --no-slab-sweep --steps 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], d['phase_ms'], 'cg', round(d['distributed_cg']['iteration_ms'],4), round(d['distributed_cg']['local_spmv_ms'],4))"
done
echo "== rows SpMV"
TB_SPMV_KERNEL=rows timeout 600 python3 bench.py --no-cpu-baseline --no-slab-sweep 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  rows kernel: cg', d['distributed_cg']['iteration_ms'], d['distributed_cg']['local_spmv_ms'])"
echo "== sweep"
timeout 900 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d['slab_sweep']['layers'].items(): print(k, v)"
} > gpurun_out/gpu_r4_third.log 2>&1
tail -c 6000 gpurun_out/gpu_r4_third.log
