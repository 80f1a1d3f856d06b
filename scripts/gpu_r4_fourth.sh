#!/bin/bash
# round 4: SpMV with 16-byte loads; where a 27-layer slab's time goes (kernel trace vs events)
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "spmv or cg or heat" 2>&1 | tail -6
echo "== sig SpMV (16-byte loads)"
timeout 600 python3 bench.py --no-cpu-baseline --no-slab-sweep 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], d['phase_ms'], 'cg', round(d['distributed_cg']['iteration_ms'],4), round(d['distributed_cg']['local_spmv_ms'],4))"
echo "== 27 layers, events vs kernel trace"
timeout 600 python3 bench.py --no-cpu-baseline --layers 27 --steps 50 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], d['phase_ms'], 'cg', round(d['distributed_cg']['iteration_ms'],4), round(d['distributed_cg']['local_spmv_ms'],4))"
rocprofv3 --kernel-trace --stats -d gpurun_out/kt27 -o kt -- python3 bench.py --no-cpu-baseline --layers 27 --steps 50 > /dev/null 2>&1
python3 scripts/rocpd_summary.py gpurun_out/kt27/kt_results.db | head -14
rm -rf gpurun_out/kt27
} > gpurun_out/gpu_r4_fourth.log 2>&1
tail -c 5000 gpurun_out/gpu_r4_fourth.log
