#!/bin/bash
# Round 5: HIP-graph replay — parity test, then the bench line with the graph forms of the step / CG iteration and the slab sweep; unstructured meshes
mkdir -p gpurun_out/r5
{
python3 -m pytest tests/test_gpu_parity.py -q -x -k "graph_replay" 2>&1 | tail -15
show() { python3 - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print('ms/step %.3f value %.4g kernel %s frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac']), d['phase_ms'])
print('graph_step', d.get('graph_step'))
c=d.get('distributed_cg')
if c: print('cg it %.4f graph %s nodes %s err %s spmv %.4f' % (c['iteration_ms'], c['graph_iteration_ms'], c['graph_nodes'], c['graph_error'], c['local_spmv_ms']))
print('patch', d.get('patch_stats')); print('spmv_plan', d.get('spmv_plan'))
if 'cpu_baseline' in d: print('cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('parity'))
if 'slab_sweep' in d:
    print('xch', d['slab_sweep'].get('exchange_latency'))
    for k,v in d['slab_sweep']['layers'].items(): print('  ', k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a!='phase_ms'})
PY
}
timeout 900 python3 bench.py --no-cpu-baseline --steps 20 2> gpurun_out/r5/bench_graph.err | tail -1 > gpurun_out/r5/bench_graph.json; show gpurun_out/r5/bench_graph.json; tail -3 gpurun_out/r5/bench_graph.err
timeout 1200 python3 bench.py --mesh shuffled --steps 20 2> gpurun_out/r5/bench_shuffled.err | tail -1 > gpurun_out/r5/bench_shuffled.json; show gpurun_out/r5/bench_shuffled.json; tail -3 gpurun_out/r5/bench_shuffled.err
timeout 900 python3 bench.py --mesh lv --steps 20 2> gpurun_out/r5/bench_lv.err | tail -1 > gpurun_out/r5/bench_lv.json; show gpurun_out/r5/bench_lv.json; tail -3 gpurun_out/r5/bench_lv.err
} > gpurun_out/r5/graph.log 2>&1
tail -c 7000 gpurun_out/r5/graph.log
