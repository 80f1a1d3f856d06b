#!/bin/bash
# Mechanics tangent at 80³: parity tests first, then timing and the kernel trace (integration + gather durations).  $1 = full: every mechanics parity test
mkdir -p gpurun_out/mechprof
{
timeout 200 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "hyperelastic_residual_and_tangent or mechanics_properties_80" 2>&1 | tail -3
if [ ${PIPESTATUS[0]} -ne 0 ]; then echo "parity failed or hung: stop"; exit 1; fi
if [ "$1" = "full" ]; then
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "nodal_fibre_field or active_stress_parity or newton_raphson_with_dirichlet or weak_boundary or hill_frameworks or condensed_sarcomere_parity or other_energies or passive_structure or sarcomere" 2>&1 | tail -4
fi
for g in lds direct; do
  echo "== gather $g"
  TB_MECH_GATHER=$g timeout 300 python3 scripts/bench_mechanics.py --n 80 --steps 3 --cpu-n 2 2>&1 | tail -1 | cut -c1-250
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/mp_sf
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/mp_sf -o mech -- python3 $R/scripts/bench_mechanics.py --n 80 --steps 3 --cpu-n 2 > /tmp/mp_sf.log 2>&1
python3 $R/scripts/rocpd_summary.py $(find /tmp/mp_sf -name "*.db" | head -1) --json $R/gpurun_out/mechprof/mech_gather_kernel_stats.json | cut -c1-200
} > gpurun_out/mech_gather.log 2>&1
tail -c 4000 gpurun_out/mech_gather.log
