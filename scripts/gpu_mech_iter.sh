#!/bin/bash
# Mechanics tangent iteration loop: quick parity, phase stamps (profiling build), timing of the product build at 80³
mkdir -p gpurun_out
{
timeout 300 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "hyperelastic_residual_and_tangent or mechanics_properties_80 or nodal_fibre_field or active_stress_parity or other_energies" 2>&1 | tail -3
if [ ${PIPESTATUS[0]} -ne 0 ]; then echo "parity failed or hung: stop"; exit 1; fi
TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so TB_PROF_STAMPS=1 timeout 300 python3 scripts/bench_mechanics.py --n 80 --steps 2 --cpu-n 2 2>&1 | grep "mechanics phases" | tail -1 | cut -c1-400
for rep in 1 2; do
timeout 300 python3 scripts/bench_mechanics.py --n 80 --steps 3 --cpu-n 2 2>&1 | tail -1 | cut -c1-250
done
} > gpurun_out/mech_iter.log 2>&1
cat gpurun_out/mech_iter.log
