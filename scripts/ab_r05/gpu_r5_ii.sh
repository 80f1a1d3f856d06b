#!/bin/bash
mkdir -p gpurun_out/r5
{
for n in 4 8; do
t0=$SECONDS
TB_BENCH_SHARE_DEVICE=1 timeout 300 python3 bench.py --gpus $n --n 32 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "^{" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('n_gpus', d['n_gpus'], 'ms/step %.3f' % d['ms_per_step'], d['phase_ms'], d['config']['layers_per_gpu'], d['config']['backend'], 'cg', d['distributed_cg']['iteration_ms'])"
echo "rc $? wall $((SECONDS-t0)) s"
done
} > gpurun_out/r5/ii.log 2>&1
grep -v amdgpu.ids gpurun_out/r5/ii.log
