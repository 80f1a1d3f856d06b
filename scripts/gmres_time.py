"""Time of one restarted-GMRES solve and one Jacobi-CG solve on the heat matrix of an n^3 box (profiling aid: where the Krylov loops spend their time)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import thunderbolt_jl_amd as tb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = tb.MI355XDevice(0)
g = tb.generate_mesh(tb.Hexahedron, (n, n, n), (0, 0, 0), (1, 1, 1), perturb=0.2)
dh = tb.DofHandler(g); sp = tb.allocate_matrix(dh)
st = tb.PatchAssemblyStrategy(dev)
kap = np.diag([4.5e-5, 2.0e-5, 1.0e-5])
M = tb.setup_operator(st, tb.BilinearMassIntegrator(tb.ConstantCoefficient(1.0)), dh, sp)
K = tb.setup_operator(st, tb.BilinearDiffusionIntegrator(tb.ConstantCoefficient(kap)), dh, sp)
tb.update_operators(M, K, 0.0)
import ctypes as C
A = dev.zeros(sp.nnz)
tb._lib.check(tb.lib().tb_heat_matrix(dev.h, M.A.n, M.A.ptr, K.A.ptr, 50.0, A.ptr))   # A = M − 50 K
pat = K.pattern
rng = np.random.default_rng(0)
b = dev.to_device(rng.standard_normal(dh.ndofs))
def run(name, fn, reps=3):
    fn(); dev.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    dev.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    print("%-6s n %d dofs %d: %.2f ms per solve, iters %d (%.3f ms per iteration) res %.2e" % (name, n, dh.ndofs, dt, out[0], dt / max(out[0], 1), out[1]))
run("cg", lambda: tb.cg_solve(pat, A, b, dev.zeros(dh.ndofs), rtol=1e-8, atol=0.0, maxiter=500))
run("gmres", lambda: tb.gmres_solve(pat, A, b, dev.zeros(dh.ndofs), rtol=1e-8, atol=0.0, maxiter=500, restart=30))
run("pcg", lambda: tb.pcg_solve(pat, A, b, dev.zeros(dh.ndofs), rtol=1e-8, atol=0.0, maxiter=500, precond=tb.ChebyshevPrecBuilder(4)))
