# R: where to post the halo exchange of a step.  One GPU, the slab one of 8 (4, 2) ranks owns, a real RCCL exchange in every step (rank 0 as its own two
# neighbours, communicator's own queue): posted in front of the reaction kernel (late: the order through round 6) or behind b, in front of M + K (first).
mkdir -p gpurun_out/r6r; rm -f gpurun_out/r6r/log.txt
run() { # label, layers, env
  echo "== $1" >> gpurun_out/r6r/log.txt
  env $3 python3 bench.py --layers $2 --steps 200 --warmup 20 --no-slab-sweep --no-cpu-baseline --no-graph --no-dist-cg > gpurun_out/r6r/line.json 2>> gpurun_out/r6r/err.txt
  python3 - >> gpurun_out/r6r/log.txt <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r6r/line.json").read().splitlines() if l.startswith("{")][-1])   # (RCCL prints its banner behind the line at exit)
    print("  ms/step %.4f" % d["ms_per_step"], {k: round(v, 4) for k, v in d["phase_ms"].items()})
except Exception as e:
    print("  failed", e)
PY
}
for rep in 1 2; do
  run "27 layers, no exchange" 27 "A=1"
  run "27 layers, exchange posted in front of the reaction kernel" 27 "TB_BENCH_SELF_EXCHANGE=late"
  run "27 layers, exchange posted behind b, in front of M + K" 27 "TB_BENCH_SELF_EXCHANGE=first"
done
run "54 layers, no exchange" 54 "A=1"
run "54 layers, late" 54 "TB_BENCH_SELF_EXCHANGE=late"
run "54 layers, first" 54 "TB_BENCH_SELF_EXCHANGE=first"
cat gpurun_out/r6r/log.txt; tail -5 gpurun_out/r6r/err.txt
