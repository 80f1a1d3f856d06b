#!/bin/bash
# first GPU pass of round 2: parity of the sum-factorised patch kernels, then timing sweep
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r02_a
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "diffusion_matrix_parity or mass_matrix_parity or single_cell or fused or properties_64 or reference_gpu" > $out/pytest.log 2>&1
tail -5 $out/pytest.log
python3 scripts/bench_assembly.py --n 216 --check > $out/asm_default.json 2> $out/asm_default.err; cat $out/asm_default.json; tail -2 $out/asm_default.err
TB_PATCH_KERNEL=legacy python3 scripts/bench_assembly.py --n 216 > $out/asm_legacy.json 2>&1; cat $out/asm_legacy.json
for tile in 7,7,7 6,6,6 5,5,6 5,5,5 4,8,8 3,7,7 8,8,3 10,10,2 16,4,4; do
  TB_PATCH_TILE=$tile python3 scripts/bench_assembly.py --n 216 > $out/asm_$tile.json 2> $out/asm_$tile.err; cat $out/asm_$tile.json; tail -1 $out/asm_$tile.err
done
