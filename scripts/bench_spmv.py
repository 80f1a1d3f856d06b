"""CSR SpMV at n³ hexes (Q1 scalar pattern, 27 nz per interior row): time per product and algorithmic bandwidth (12 B/nnz + 16 B/row);
TB_SPMV_LANES=16 selects the lanes-per-row kernel, unset = the stream kernel.  Also checks the product against scipy on the host."""
import argparse
import os
import sys
import time

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=216)
ap.add_argument("--reps", type=int, default=50)
a = ap.parse_args()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import thunderbolt_jl_amd as tb  # noqa: E402
dev = tb.MI355XDevice(0)
g = tb.generate_mesh(tb.Hexahedron, (a.n,) * 3)
dh = tb.DofHandler(g)
sp = tb.allocate_matrix(dh)
M = tb.update_operator(tb.setup_operator(tb.PatchAssemblyStrategy(dev), tb.BilinearMassIntegrator(tb.ConstantCoefficient(1.0)), dh, sp), 0.0)
rng = np.random.default_rng(0)
xh = rng.normal(size=dh.ndofs)
x = dev.to_device(xh)
y = dev.zeros(dh.ndofs)
M.mul(y, x)
import scipy.sparse as ssp
ref = ssp.csr_matrix((M.A.to_host(), sp.colidx, sp.rowptr), shape=(dh.ndofs,) * 2) @ xh
print("rel err vs scipy", np.abs(y.to_host() - ref).max() / np.abs(ref).max())
dev.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    M.mul(y, x)
dev.synchronize()
dt = (time.perf_counter() - t0) / a.reps
print(f"n={a.n} nnz={sp.nnz} spmv {dt * 1e3:.3f} ms  {(12 * sp.nnz + 16 * dh.ndofs) / dt / 1e12:.2f} TB/s")
b = dev.to_device(ref)
u = dev.zeros(dh.ndofs)
dev.synchronize()
t0 = time.perf_counter()
its, res = tb.cg_solve(M.pattern, M.A, b, u, rtol=1e-10, atol=0.0, maxiter=500)
dev.synchronize()
dt = time.perf_counter() - t0
print(f"CG on M: {its} iterations, {dt / max(its, 1) * 1e3:.3f} ms/iteration, err {np.abs(u.to_host() - xh).max():.2e}")
