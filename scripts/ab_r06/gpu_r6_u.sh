# U: vector-patch tile sizes on the 27-layer slab (profiling build honours TB_VPATCH_TILE): generation quantisation of 2 460 patches over 768 resident workgroups
mkdir -p gpurun_out/r6u; rm -f gpurun_out/r6u/log.txt
export TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
run() { # label, layers, env
  echo "== $1" >> gpurun_out/r6u/log.txt
  env $3 python3 bench.py --layers $2 --steps 200 --warmup 20 --no-slab-sweep --no-cpu-baseline --no-graph --no-dist-cg > gpurun_out/r6u/line.json 2>> gpurun_out/r6u/err.txt
  python3 - >> gpurun_out/r6u/log.txt <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r6u/line.json").read().splitlines() if l.startswith("{")][-1])
    print("  ms/step %.4f " % d["ms_per_step"], {k: round(v, 4) for k, v in d["phase_ms"].items()})
except Exception as e:
    print("  failed", e)
PY
}
for L in 27 216; do
for t in 8,8,8 8,8,4 6,6,7 8,8,5 7,7,7 6,8,8 4,8,8; do
  run "$L layers, tile $t" $L "TB_VPATCH_TILE=$t"
done
done
cat gpurun_out/r6u/log.txt; grep -v "Ferrite conv\|amdgpu.ids" gpurun_out/r6u/err.txt | tail
