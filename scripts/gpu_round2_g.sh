#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r02_g
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "diffusion_matrix_parity or mass_matrix_parity or single_cell or fused or properties_64" > $out/pytest.log 2>&1
tail -3 $out/pytest.log
python3 scripts/bench_assembly.py --n 216 --check 2>/dev/null | tee $out/asm_staged.json
for tile in 5,5,6 6,6,6 6,7,7; do TB_PATCH_TILE=$tile python3 scripts/bench_assembly.py --n 216 2>/dev/null | tee -a $out/asm_tiles.txt; done
TB_PATCH_KERNEL=simple TB_PATCH_TILE=5,5,6 python3 scripts/bench_assembly.py --n 216 2>/dev/null | tee $out/asm_simple.json
for f in 0 2 7; do
  TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so TB_PATCH_TILE=5,5,6 TB_DEBUG_FLAGS=$f python3 scripts/bench_assembly.py --n 216 2>/dev/null | sed "s/^/flags=$f /" | tee -a $out/ablation.txt
done
