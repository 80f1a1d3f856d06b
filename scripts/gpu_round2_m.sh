#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r02_m
mkdir -p $out
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hyperelastic or mechanics or subdomains or contracting or land2015 or condensed" > $out/pytest.log 2>&1
tail -4 $out/pytest.log
python3 scripts/bench_mechanics.py --n 80 --cpu-n 2 2>/dev/null | tee $out/mech_pipelined.json
TB_EA_PIPELINE=0 python3 scripts/bench_mechanics.py --n 80 --cpu-n 2 2>/dev/null | tee $out/mech_sequential.json
for c in 8 48 96; do TB_EA_CHUNKS=$c python3 scripts/bench_mechanics.py --n 80 --cpu-n 2 2>/dev/null | sed "s/^/chunks=$c /" | tee -a $out/mech_chunks.txt; done
