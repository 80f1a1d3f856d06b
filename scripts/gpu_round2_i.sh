#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r02_i
mkdir -p $out
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix_parity or single_cell or fused or properties or reference_gpu or tet_mesh or error_codes" > $out/pytest.log 2>&1
tail -3 $out/pytest.log
for tile in 5,5,6 4,6,6 6,5,5 3,7,7 7,7,3 4,5,7 7,5,4 8,4,4; do TB_BENCH_ORDER=fused TB_PATCH_TILE=$tile python3 scripts/bench_assembly.py --n 216 2>/dev/null | tee -a $out/asm_tiles.txt; done
python3 bench.py > $out/bench_216.json 2> $out/bench_216.err; tail -c 1500 $out/bench_216.json; tail -3 $out/bench_216.err
