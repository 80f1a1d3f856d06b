# AA: reaction kernel, second workgroup of every CU started late (TB_REACTION_STAGGER = n x 8128 cycles, profiling build): do staggered load / arithmetic phases overlap better?
mkdir -p gpurun_out/r6aa; rm -f gpurun_out/r6aa/log.txt
export TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
run() { # label, args, env
  echo "== $1" >> gpurun_out/r6aa/log.txt
  env $3 python3 bench.py $2 --no-slab-sweep --no-cpu-baseline --no-graph --no-dist-cg > gpurun_out/r6aa/line.json 2>> gpurun_out/r6aa/err.txt
  python3 - >> gpurun_out/r6aa/log.txt <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r6aa/line.json").read().splitlines() if l.startswith("{")][-1])
    print("  ms/step %.4f  reaction %.4f" % (d["ms_per_step"], d["phase_ms"]["reaction"]))
except Exception as e:
    print("  failed", e)
PY
}
for rep in 1 2; do
for st in 0 1 2 3 5; do
  run "27 layers, stagger $st" "--layers 27 --steps 200 --warmup 20" "TB_REACTION_STAGGER=$st"
done
done
for st in 0 1 2 4 8; do
  run "216 layers, stagger $st" "--steps 40 --warmup 5" "TB_REACTION_STAGGER=$st"
done
cat gpurun_out/r6aa/log.txt; grep -v "Ferrite conv\|amdgpu.ids" gpurun_out/r6aa/err.txt | tail -3
