#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r02_s
mkdir -p $out
TB_PATCH_KERNEL=stream TB_PATCH_TILE=5,5,5 timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix_parity or fused or properties_64 or single_cell" > $out/pytest_stream.log 2>&1; tail -3 $out/pytest_stream.log
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix_parity or fused or properties_64 or single_cell or unstructured" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
TB_PLAN_VERBOSE=1 TB_PATCH_KERNEL=stream TB_PATCH_TILE=5,5,5 python3 scripts/bench_assembly.py --n 216 --check 2>&1 | grep -v amdgpu | tee $out/asm_stream.json
TB_PATCH_TILE=5,5,5 python3 scripts/bench_assembly.py --n 216 2>/dev/null | tee $out/asm_staged_555.json
python3 scripts/bench_assembly.py --n 216 2>/dev/null | tee $out/asm_staged.json
