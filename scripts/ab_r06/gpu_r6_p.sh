#!/bin/bash
# round 6, experiment P: mass adds riding with the stiffness adds (one address per pair, -DTB_PATCH_FUSE_KM) against the two separate scatter loops
mkdir -p gpurun_out
run() { # label, env... -- bench args
  echo "== $1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python3 bench.py --no-slab-sweep --no-graph --no-cpu-baseline --no-dist-cg --steps 20 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()}, d['roofline'].get('kernel'))"
}
{
TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_fkm.so timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_mass_diffusion_pair" 2>&1 | tail -2
for rep in 1 2; do
  run base TB_X=1 --
  run fused-km TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_fkm.so --
done
} > gpurun_out/gpu_r6_p.log 2>&1
cat gpurun_out/gpu_r6_p.log
