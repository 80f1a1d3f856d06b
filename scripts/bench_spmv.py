#!/usr/bin/env python3
"""y = A·x on the 216³ heat-matrix pattern: the CSR kernel of tb_spmv_csr against the sliced mirror (tb_spmv_mirror), grid sweep of the latter.
Prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=216)
ap.add_argument("--grids", default="0,512,768,1024,1536,2048,4096,40000")
args = ap.parse_args()
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
g = tb.generate_mesh(tb.Hexahedron, (args.n,) * 3, perturb=0.1)
dh = tb.DofHandler(g)
sp = tb.allocate_matrix(dh)
pat = tb.DevicePattern(tb.DeviceMesh(dev, dh), sp)
rng = np.random.default_rng(0)
A = dev.to_device(rng.normal(size=sp.nnz))
x, y = dev.to_device(rng.normal(size=dh.ndofs)), dev.zeros(dh.ndofs)
L = tb.lib()

def timed(reps=20):
    from _preroll import preroll
    preroll(dev, lambda: tb.check(L.tb_spmv_csr(pat.h, A.ptr, x.ptr, 1.0, 0.0, y.ptr)), 80.0)   # steady clocks
    for _ in range(3):
        tb.check(L.tb_spmv_csr(pat.h, A.ptr, x.ptr, 1.0, 0.0, y.ptr))
    e0, e1 = dev.event(), dev.event()
    e0.record()
    for _ in range(reps):
        tb.check(L.tb_spmv_csr(pat.h, A.ptr, x.ptr, 1.0, 0.0, y.ptr))
    e1.record(); dev.synchronize()
    return e0.elapsed_ms(e1) / reps

d = dev.zeros(1)
def timed_dot(reps=20):
    from _preroll import preroll
    preroll(dev, lambda: tb.check(L.tb_spmv_csr(pat.h, A.ptr, x.ptr, 1.0, 0.0, y.ptr)), 80.0)   # steady clocks
    for _ in range(3):
        tb.check(L.tb_spmv_csr_dot(pat.h, A.ptr, x.ptr, y.ptr, d.ptr))
    e0, e1 = dev.event(), dev.event()
    e0.record()
    for _ in range(reps):
        tb.check(L.tb_spmv_csr_dot(pat.h, A.ptr, x.ptr, y.ptr, d.ptr))
    e1.record(); dev.synchronize()
    return e0.elapsed_ms(e1) / reps

out = {"rows": dh.ndofs, "nnz": sp.nnz, "csr_dot_ms": timed_dot(), "csr_ms": timed()}
y_csr = y.to_host()
assert pat.mirror(A)
e0, e1 = dev.event(), dev.event()
e0.record(); pat.mirror(A); e1.record(); dev.synchronize()
out["mirror_bind_ms"] = e0.elapsed_ms(e1)
out["mirror_ms_by_grid"] = {}
for gsz in args.grids.split(","):
    if gsz == "0":
        os.environ.pop("TB_SPMV_MIRROR_GRID", None)
    else:
        os.environ["TB_SPMV_MIRROR_GRID"] = gsz
    out.setdefault("mirror_dot_ms_by_grid", {})[gsz] = timed_dot()
    out["mirror_ms_by_grid"][gsz] = timed()
out["bit_identical"] = bool(np.array_equal(y.to_host(), y_csr))
bytes_ = 8.0 * sp.nnz + 16.0 * dh.ndofs
out["mirror_TBps"] = bytes_ / (min(out["mirror_ms_by_grid"].values()) * 1e-3) / 1e12
print(json.dumps(out))
