#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of scripts/collect_profiles.sh: HBM-side bytes per launch of the bench kernels.
reads  = 128·(RDREQ − BUBBLE − RDREQ_32B) + 64·BUBBLE + 32·RDREQ_32B   (TCC_EA0 request counters; rocprofv3's FETCH_SIZE expression tallies the
          first group at 64 B, which reads ½ of a wide coalesced stream on gfx950 — MI355X_MICROARCH.md §HBM; calibrated on the reaction kernel,
          whose algorithmic traffic is known exactly)
writes = WRITE_SIZE · 1024 (exact).
usage: make_traffic_json.py <dir with pmc1.json (RDREQ), pmc2.json (WRITE_SIZE), pmc3.json (FETCH_SIZE)> <tag>"""
import json
import os
import sys

d, tag = sys.argv[1], sys.argv[2]
rd = json.load(open(os.path.join(d, "pmc1.json")))
wr = json.load(open(os.path.join(d, "pmc2.json")))
ft = json.load(open(os.path.join(d, "pmc3.json")))
bench = {}
try:
    bench = json.loads(open(os.path.join(d, "bench_216.json")).read().strip().splitlines()[-1])
except Exception:
    pass
out = {"source": "profiles/%s/pmc1-3.json (rocprofv3 --pmc, separate passes of python3 bench.py at 216^3)" % tag,
       "cells": bench.get("config", {}).get("cells_per_gpu", 10077696),
       "note": "read_bytes = 128*(RDREQ - BUBBLE - RDREQ_32B) + 64*BUBBLE + 32*RDREQ_32B; write_bytes = WRITE_SIZE KiB * 1024; fetch_size_bytes = FETCH_SIZE as reported",
       "kernels": {}}


def role(name):
    if "k_patch_hex8" in name and "Lb1ELb1E" in name[name.find("k_patch_hex8"):][:40]:
        return "fused"
    if "k_patch_hex8" in name and "Lb1ELb0E" in name[name.find("k_patch_hex8"):][:40]:
        return "diffusion"
    if "k_patch_hex8" in name and "Lb0ELb1E" in name[name.find("k_patch_hex8"):][:40]:
        return "mass"
    if "k_reaction" in name:
        return "reaction" if "reaction" not in out["kernels"] else None
    if "k_vector" in name or "k_source" in name:
        return "source" if "source" not in out["kernels"] else None
    if "k_spmv_mirrorILb1E" in name:
        return "spmv_mirror_dot"     # the product of a CG iteration (fused with pᵀAp) from the sliced mirror
    if "k_spmv_mirrorILb0E" in name:
        return "spmv_mirror"
    if "k_spmv_sig_rowsILi2048ELb0E" in name:
        return "spmv_csr"
    if "k_mirror_fill" in name:
        return "mirror_fill"
    return None


for name, v in sorted(rd.items(), key=lambda kv: -kv[1].get("total_ms", 0)):
    r = role(name)
    p = v.get("pmc_per_launch", {})
    if r is None or "TCC_EA0_RDREQ_sum" not in p:
        continue
    rq, r32, bub = p["TCC_EA0_RDREQ_sum"], p.get("TCC_EA0_RDREQ_32B_sum", 0.0), p.get("TCC_BUBBLE_sum", 0.0)
    w = wr.get(name, {}).get("pmc_per_launch", {}).get("WRITE_SIZE")
    f = ft.get(name, {}).get("pmc_per_launch", {}).get("FETCH_SIZE")
    out["kernels"][r] = {"kernel": name, "avg_us": v["avg_us"], "read_bytes": 128.0 * (rq - bub - r32) + 64.0 * bub + 32.0 * r32,
                         "write_bytes": None if w is None else w * 1024.0, "fetch_size_bytes": None if f is None else f * 1024.0,
                         "RDREQ": rq, "RDREQ_32B": r32, "BUBBLE": bub}
print(json.dumps(out, indent=1))
