#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r02_c
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "diffusion_matrix_parity or mass_matrix_parity or single_cell or fused or properties_64 or reference_gpu" > $out/pytest.log 2>&1
tail -5 $out/pytest.log
python3 scripts/bench_assembly.py --n 216 --check 2>/dev/null | tee $out/asm_persistent.json
TB_PATCH_KERNEL=simple python3 scripts/bench_assembly.py --n 216 2>/dev/null | tee $out/asm_simple.json
for w in 1 2 3 4; do TB_PATCH_WGS_PER_CU=$w python3 scripts/bench_assembly.py --n 216 2>/dev/null | sed "s/^/wgs=$w /" | tee -a $out/asm_wgs.txt; done
for tile in 5,5,5 4,6,6 4,4,8 3,6,8 4,5,5 4,4,4; do
  TB_PATCH_TILE=$tile python3 scripts/bench_assembly.py --n 216 2>/dev/null | tee -a $out/asm_tiles.txt
done
make -C thunderbolt.jl_amd/csrc ablation -j32 > /dev/null 2>&1
for f in 0 1 2 4 7; do
  TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so TB_DEBUG_FLAGS=$f python3 scripts/bench_assembly.py --n 216 2>/dev/null | sed "s/^/flags=$f /" | tee -a $out/ablation.txt
done
