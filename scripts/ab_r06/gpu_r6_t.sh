# T: the N-rank step on one GPU through HaloExchange (pack, exchange, unpack-add; rank 0 its own two neighbours): torch.distributed (RCCL, world size 1), the C ABI's
# communicator in the device's queue, and on its own queue.  27 layers (GPU-bound) and a 24-cube (host-bound: what the host needs per step).
mkdir -p gpurun_out/r6t; rm -f gpurun_out/r6t/log.txt
run() { # label, args, env
  echo "== $1" >> gpurun_out/r6t/log.txt
  env $3 python3 bench.py $2 --steps 200 --warmup 20 --no-slab-sweep --no-cpu-baseline --no-graph --no-dist-cg > gpurun_out/r6t/line.json 2>> gpurun_out/r6t/err.txt
  python3 - >> gpurun_out/r6t/log.txt <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r6t/line.json").read().splitlines() if l.startswith("{")][-1])
    print("  ms/step %.4f  host enqueue %.4f " % (d["ms_per_step"], d["host_enqueue_ms_per_step"]), {k: round(v, 4) for k, v in d["phase_ms"].items()})
except Exception as e:
    print("  failed", e)
PY
}
for rep in 1 2; do
  for sz in "--layers 27" "--n 24"; do
    run "$sz, no exchange" "$sz" "A=1"
    run "$sz, torch.distributed" "$sz" "TB_BENCH_SELF_EXCHANGE=torch"
    run "$sz, C ABI in the device's queue" "$sz" "TB_BENCH_SELF_EXCHANGE=abi"
    run "$sz, C ABI on its own queue" "$sz" "TB_BENCH_SELF_EXCHANGE=abi-overlap"
  done
done
cat gpurun_out/r6t/log.txt; grep -v "Ferrite conv\|amdgpu.ids" gpurun_out/r6t/err.txt | tail -20
