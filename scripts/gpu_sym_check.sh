#!/bin/bash
# GPU contact of the symmetric-accumulator patch kernel: parity tests that touch it, A/B bench lines, phase stamps of the profiling build.
mkdir -p gpurun_out
{
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused_mass_diffusion or mass_matrix_parity or diffusion_matrix_parity or unstructured_hexahedral or properties_64 or single_cell or empty_and_ragged or error_codes" 2>&1 | tail -5
for mode in sym staged onepass; do
  echo "== $mode"
  if [ $mode = staged ]; then export TB_PATCH_KERNEL=staged; else unset TB_PATCH_KERNEL; fi
  if [ $mode = onepass ]; then export TB_SYM_MODE=onepass; else unset TB_SYM_MODE; fi
  timeout 600 python3 bench.py --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['phase_ms'], d['roofline']['frac'])"
done
unset TB_PATCH_KERNEL
export TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so TB_PROF_STAMPS=1 TB_PLAN_VERBOSE=1
for mode in sym onepass; do
  echo "== stamps $mode"
  if [ $mode = onepass ]; then export TB_SYM_MODE=onepass; else unset TB_SYM_MODE; fi
  timeout 600 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep -a "tbhip\]" | sort | uniq -c | sort -rn | head -8
done
} > gpurun_out/sym_check.log 2>&1
tail -c 6000 gpurun_out/sym_check.log
