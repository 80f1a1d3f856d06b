#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r02_h
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "diffusion_matrix_parity or mass_matrix_parity or single_cell or fused or properties_64" > $out/pytest.log 2>&1
tail -3 $out/pytest.log
python3 scripts/bench_assembly.py --n 216 --check 2>/dev/null | tee $out/asm_staged.json
TB_PATCH_TILE=5,5,6 python3 scripts/bench_assembly.py --n 216 2>/dev/null | tee -a $out/asm_tiles.txt
TB_PATCH_KERNEL=persistent TB_PATCH_TILE=5,5,6 python3 scripts/bench_assembly.py --n 216 2>/dev/null | tee $out/asm_persistent.json
TB_PROF_STAMPS=1 TB_BENCH_ORDER=fused TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so TB_PATCH_TILE=5,5,6 TB_DEBUG_FLAGS=0 python3 scripts/bench_assembly.py --n 216 --reps 2 2>&1 | grep -v amdgpu | tail -2 | tee -a $out/stamps_staged.txt
TB_PATCH_KERNEL=persistent TB_PROF_STAMPS=1 TB_BENCH_ORDER=fused TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so TB_PATCH_TILE=5,5,6 TB_DEBUG_FLAGS=0 python3 scripts/bench_assembly.py --n 216 --reps 2 2>&1 | grep -v amdgpu | tail -2 | tee -a $out/stamps_staged.txt
