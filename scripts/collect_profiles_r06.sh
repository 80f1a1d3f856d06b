#!/bin/bash
# Run on the GPU box (gpurun -- 'bash scripts/collect_profiles_r06.sh <tag>'): rocprofv3 evidence for profiles/<tag>/ and profiles/traffic.json, round 6.
# What changed against scripts/collect_profiles.sh: every traced run brackets its timed region with marker kernels (bench.py --trace-markers) and the
# summaries keep the dispatches between them only (rocpd_summary.py --between scan): the kernel-trace average is that of the TIMED launches of the SAME
# process whose line is stored beside it (bench_under_rocprof.json), and the PMC passes count the same launches.  Kernel trace and every PMC group are
# separate passes (no PMC together with tracing domains).
tag=${1:-r06_v1}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p "$out"
bash scripts/host_probe.sh > "$out/host_probe.txt" 2>&1
python3 bench.py > "$out/bench_216.json" 2> "$out/bench_216.err"
rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt -- python3 bench.py --no-cpu-baseline --no-slab-sweep --no-graph --no-dist-cg --trace-markers > "$out/bench_under_rocprof.json" 2>/dev/null
python3 scripts/rocpd_summary.py "$out/kt/kt_results.db" --between scan --json "$out/kernel_stats.json" > "$out/kernel_stats.txt"
python3 - "$out" <<'PY'
import json, sys
d = sys.argv[1]
line = json.loads(open(d + "/bench_under_rocprof.json").read().strip().splitlines()[-1])
ks = json.load(open(d + "/kernel_stats.json"))
name = next(k for k in ks if "k_patch_hex8" in k)
ev, tr = line["roofline"]["launch_ms"], ks[name]["avg_us"] * 1e-3
open(d + "/line_vs_trace.txt", "w").write(
    "same process (bench.py --trace-markers under rocprofv3 --kernel-trace): %s\n  HIP events of the line (roofline.launch_ms) %.4f ms\n  kernel-trace average over the %d timed launches %.4f ms\n  difference %.2f %%\n"
    % (name[:90], ev, ks[name]["calls"], tr, 100.0 * (tr - ev) / ev))
print(open(d + "/line_vs_trace.txt").read())
PY
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "WRITE_SIZE" "FETCH_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d "$out/pmc$i" -o pmc -- python3 bench.py --no-cpu-baseline --no-slab-sweep --no-graph --no-dist-cg --steps 5 --trace-markers > /dev/null 2>&1
  python3 scripts/rocpd_summary.py "$out/pmc$i/pmc_results.db" --between scan --json "$out/pmc$i.json" > /dev/null
done
python3 scripts/make_traffic_json.py "$out" "$tag" > "$out/traffic.json"
if [ "$2" != "bench-only" ]; then
  python3 scripts/bench_mechanics.py --n 80 > "$out/mechanics_80_q2.json" 2>/dev/null
  for st in atomic color; do python3 scripts/bench_mechanics.py --n 80 --strategy $st --cpu-n 2 2>/dev/null | tail -1; done > "$out/mechanics_80_q2_scattering_strategies.json"
  python3 scripts/bench_tets.py --n 100 > "$out/tets_100.json" 2>/dev/null
  python3 scripts/bench_q2_scalar.py --n 64 > "$out/q2_scalar_64.json" 2>/dev/null
  python3 bench.py --mesh shuffled --no-slab-sweep > "$out/bench_shuffled.json" 2> "$out/bench_shuffled.err"
  python3 bench.py --mesh shuffled --renumber dofs --no-slab-sweep > "$out/bench_shuffled_renumbered_dofs.json" 2> "$out/bench_shuffled_renumbered_dofs.err"
  python3 bench.py --mesh lv --no-slab-sweep > "$out/bench_lv.json" 2> "$out/bench_lv.err"
  rocprofv3 --kernel-trace --stats -d "$out/kt27" -o kt27 -- python3 bench.py --no-cpu-baseline --layers 27 --steps 50 --no-graph --no-dist-cg --trace-markers > "$out/bench_slab27_under_rocprof.json" 2>/dev/null
  python3 scripts/rocpd_summary.py "$out/kt27/kt27_results.db" --between scan --json "$out/slab27_kernel_stats.json" > "$out/slab27_kernel_stats.txt"
  python3 scripts/rccl_latency.py 47089 2>/dev/null | grep "^{" > "$out/rccl_latency_world1.json"
  TB_BENCH_SHARE_DEVICE=1 python3 bench.py --gpus 2 --mesh lv --no-cpu-baseline --no-slab-sweep --no-graph --checksums > "$out/bench_lv_two_ranks_one_device.json" 2> "$out/bench_lv_two_ranks_one_device.err"
fi
rm -rf "$out"/kt27 "$out"/kt "$out"/pmc1 "$out"/pmc2 "$out"/pmc3 "$out"/pmc4
tail -c 900 "$out/bench_216.json"; head -6 "$out/kernel_stats.txt"; head -30 "$out/traffic.json"
