#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{
cd $R
python3 -m pytest tests/test_gpu_parity.py -q -x -k "chunked_linearization" 2>&1 | tail -3
python3 scripts/bench_mechanics.py --n 80 --steps 4 --cpu-n 4 --chunks 0,4,8,16 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('linearize %.2f ms residual %.2f ms' % (d['linearize_ms'], d['residual_ms']), d['linearize_ms_by_chunks'])"
export TB_MECH_CHUNKS=8
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/chunk_trace -- python3 scripts/bench_mechanics.py --n 80 --steps 2 --cpu-n 4 > /dev/null 2>&1
python3 scripts/trace_timeline.py gpurun_out/chunk_trace "" 40
} > $R/gpurun_out/r4_chunks.log 2>&1
rm -rf $R/gpurun_out/chunk_trace
tail -c 6000 $R/gpurun_out/r4_chunks.log
