# X: what the phase events cost a step (27 layers and the whole box): --phase-events all | matrix | none
mkdir -p gpurun_out/r6x; rm -f gpurun_out/r6x/log.txt
run() { # label, args
  echo "== $1" >> gpurun_out/r6x/log.txt
  python3 bench.py $2 --no-slab-sweep --no-cpu-baseline --no-graph --no-dist-cg > gpurun_out/r6x/line.json 2>> gpurun_out/r6x/err.txt
  python3 - >> gpurun_out/r6x/log.txt <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r6x/line.json").read().splitlines() if l.startswith("{")][-1])
    print("  ms/step %.4f  host enqueue %.4f " % (d["ms_per_step"], d["host_enqueue_ms_per_step"]), {k: (round(v, 4) if v == v else None) for k, v in d["phase_ms"].items()})
except Exception as e:
    print("  failed", e)
PY
}
for rep in 1 2; do
  for pe in all matrix none; do
    run "27 layers, --phase-events $pe" "--layers 27 --steps 200 --warmup 20 --phase-events $pe"
  done
done
for pe in all matrix none; do
  run "216 layers, --phase-events $pe" "--steps 40 --warmup 5 --phase-events $pe"
done
cat gpurun_out/r6x/log.txt; grep -v "Ferrite conv\|amdgpu.ids" gpurun_out/r6x/err.txt | tail -5
