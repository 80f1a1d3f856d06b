mkdir -p gpurun_out/n8
common="--n 24 --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-slab-sweep --checksums --preroll-ms 0"
python3 bench.py --gpus 1 $common 2>/dev/null | grep "^{" > gpurun_out/n8/one.json
( time TB_BENCH_SHARE_DEVICE=1 python3 bench.py --gpus 8 $common 2> gpurun_out/n8/err8.txt | grep "^{" > gpurun_out/n8/eight.json ) 2>&1 | grep real
python3 - <<'PY'
import json
a=json.load(open("gpurun_out/n8/one.json")); b=json.load(open("gpurun_out/n8/eight.json"))
print(b["n_gpus"], b["config"]["layers_per_gpu"], b["ms_per_step"], b["phase_ms"])
for k,v in a["checksums"].items(): print(k, v, b["checksums"][k], abs(b["checksums"][k]-v)/abs(v))
print(b.get("distributed_cg",{}).get("iteration_ms"))
PY
tail -3 gpurun_out/n8/err8.txt
