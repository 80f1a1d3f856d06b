#!/bin/bash
# round 6, experiment C: persistent streaming form (spill-free build) against the record kernel, with phase stamps
mkdir -p gpurun_out
L=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
run() { # label, env...
  echo "== $1"; shift
  env "$@" timeout 600 python3 bench.py --no-cpu-baseline --no-slab-sweep --no-dist-cg --no-graph --steps 20 2>gpurun_out/r6c_err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()}, d['roofline'].get('kernel'), 'setup', {k: round(v,2) for k,v in d['setup_s'].items()})"
  grep -E "phases" gpurun_out/r6c_err.log | tail -1
}
{
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mass or diffusion or fused or patch or config1 or unstructured or 216" 2>&1 | tail -5
for rep in 1 2; do
  run record TB_PATCH_KERNEL=record
  run stream TB_PATCH_KERNEL=stream
done
run stamps-record TB_LIBTBHIP=$L TB_PROF_STAMPS=1 TB_PATCH_KERNEL=record
run stamps-stream TB_LIBTBHIP=$L TB_PROF_STAMPS=1 TB_PATCH_KERNEL=stream
} > gpurun_out/gpu_r6_c.log 2>&1
cat gpurun_out/gpu_r6_c.log
