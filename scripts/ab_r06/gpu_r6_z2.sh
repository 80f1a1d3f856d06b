# Z2: LDS bank-conflict counters of the source kernel under the two slot layouts (profiling build)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/r6z
export TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d gpurun_out/r6z/pmcA -o pmc -- python3 bench.py --no-cpu-baseline --no-slab-sweep --no-graph --no-dist-cg --steps 5 --trace-markers > /dev/null 2>&1
python3 scripts/rocpd_summary.py $(find gpurun_out/r6z/pmcA -name '*.db' | head -1) --between scan --json gpurun_out/r6z/pmc_slots12.json > /dev/null
export TB_VPATCH_SLOTS=touch
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d gpurun_out/r6z/pmcB -o pmc -- python3 bench.py --no-cpu-baseline --no-slab-sweep --no-graph --no-dist-cg --steps 5 --trace-markers > /dev/null 2>&1
python3 scripts/rocpd_summary.py $(find gpurun_out/r6z/pmcB -name '*.db' | head -1) --between scan --json gpurun_out/r6z/pmc_touch.json > /dev/null
rm -rf gpurun_out/r6z/pmcA gpurun_out/r6z/pmcB
python3 - <<'PY'
import json
for f in ("pmc_slots12", "pmc_touch"):
    d = json.load(open("gpurun_out/r6z/%s.json" % f))
    for k, v in d.items():
        if "k_vector_hex8_patch" in k:
            p = v.get("pmc_per_launch", {})
            print(f, v["avg_us"], p, "conflict cycles per LDS instruction: %.3f" % (p.get("SQ_LDS_BANK_CONFLICT", 0) / max(p.get("SQ_INSTS_LDS", 1), 1)))
PY
