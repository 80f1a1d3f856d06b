#!/bin/bash
# Mechanics tangent: contraction variants (sum-factorised default, matrix cores, vector sweep) — parity tests, then timing + kernel trace at 80³
mkdir -p gpurun_out/mechprof
{
timeout 150 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "hyperelastic_residual_and_tangent" 2>&1 | tail -3
if [ ${PIPESTATUS[0]} -ne 0 ]; then echo "first parity test failed or hung: stop"; exit 1; fi
if [ "$1" = "full" ]; then
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "nodal_fibre_field or active_stress_parity or newton_raphson_with_dirichlet or weak_boundary or hill_frameworks or mechanics_properties_80 or condensed_sarcomere_parity or other_energies" 2>&1 | tail -4
fi
for mode in sf mfma; do
  echo "== $mode"
  if [ $mode = mfma ]; then export TB_MECH_CONTRACT=mfma; else unset TB_MECH_CONTRACT; fi
  timeout 300 python3 scripts/bench_mechanics.py --n 80 --steps 3 --cpu-n 2 2>&1 | tail -1 | cut -c1-330
done
unset TB_MECH_CONTRACT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/mp_sf
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/mp_sf -o mech -- python3 $R/scripts/bench_mechanics.py --n 80 --steps 3 --cpu-n 2 > /tmp/mp_sf.log 2>&1
python3 $R/scripts/rocpd_summary.py $(find /tmp/mp_sf -name "*.db" | head -1) --json $R/gpurun_out/mechprof/mech_sf_kernel_stats.json | cut -c1-200
} > gpurun_out/mech_check.log 2>&1
tail -c 5000 gpurun_out/mech_check.log
