#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r02_k
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "source_vector or reference_gpu or needs_update or properties_64 or error_codes" > $out/pytest.log 2>&1
tail -8 $out/pytest.log
TB_PLAN_VERBOSE=1 TB_BENCH_ORDER=vec python3 scripts/bench_assembly.py --n 216 2>&1 | grep -v amdgpu | tee $out/vec_default.txt
TB_VECTOR_KERNEL=legacy TB_BENCH_ORDER=vec python3 scripts/bench_assembly.py --n 216 2>/dev/null | tee $out/vec_legacy.txt
for tile in 6,6,6 10,10,10 16,8,4 12,8,8; do TB_VPATCH_TILE=$tile TB_BENCH_ORDER=vec python3 scripts/bench_assembly.py --n 216 2>/dev/null | sed "s/^/vtile=$tile /" | tee -a $out/vec_tiles.txt; done
