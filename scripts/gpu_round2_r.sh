#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r02_r
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix_parity or fused or unstructured" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
python3 scripts/bench_assembly.py --n 216 2>/dev/null | tee $out/asm_diag.json
python3 scripts/bench_assembly.py --n 160 --fibre 2>/dev/null | tee $out/asm_fibre_new.json
TB_PATCH_KERNEL=legacy python3 scripts/bench_assembly.py --n 160 --fibre 2>/dev/null | tee $out/asm_fibre_legacy.json
