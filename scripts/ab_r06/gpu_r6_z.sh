# Z: node slots of the vector patches: ascending global id with twelve slots per nine-node line (default) against first touch (TB_VPATCH_SLOTS=touch, profiling build)
mkdir -p gpurun_out/r6z; rm -f gpurun_out/r6z/log.txt
python3 -m pytest tests -m gpu -q -x -k "source or vector or linear_form or bench_two_ranks or parity" 2>&1 | tail -3 >> gpurun_out/r6z/log.txt
export TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
run() { # label, args, env
  echo "== $1" >> gpurun_out/r6z/log.txt
  env $3 python3 bench.py $2 --no-slab-sweep --no-cpu-baseline --no-graph --no-dist-cg > gpurun_out/r6z/line.json 2>> gpurun_out/r6z/err.txt
  python3 - >> gpurun_out/r6z/log.txt <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r6z/line.json").read().splitlines() if l.startswith("{")][-1])
    print("  ms/step %.4f " % d["ms_per_step"], {k: round(v, 4) for k, v in d["phase_ms"].items()})
except Exception as e:
    print("  failed", e)
PY
}
for rep in 1 2; do
  run "216 layers, slots of twelve per line" "--steps 40 --warmup 5" "A=1"
  run "216 layers, first-touch slots" "--steps 40 --warmup 5" "TB_VPATCH_SLOTS=touch"
  run "27 layers, slots of twelve per line" "--layers 27 --steps 200 --warmup 20" "A=1"
  run "27 layers, first-touch slots" "--layers 27 --steps 200 --warmup 20" "TB_VPATCH_SLOTS=touch"
done
run "shuffled 96, slots by id" "--mesh shuffled --n 96 --steps 40 --warmup 5" "A=1"
run "shuffled 96, first touch" "--mesh shuffled --n 96 --steps 40 --warmup 5" "TB_VPATCH_SLOTS=touch"
run "lv, slots by id" "--mesh lv --steps 40 --warmup 5" "A=1"
run "lv, first touch" "--mesh lv --steps 40 --warmup 5" "TB_VPATCH_SLOTS=touch"
cat gpurun_out/r6z/log.txt; grep -v "Ferrite conv\|amdgpu.ids" gpurun_out/r6z/err.txt | tail -5
