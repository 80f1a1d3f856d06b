"""Reaction step and source vector timed on a 216x216xL slab (L from argv, default 27) and printed as one line: the kernels whose slab
efficiency decides the strong-scaling prediction.  Profiling aid (used with TB_LIBTBHIP=…/libtbhip_ablation.so and its grid switches)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import thunderbolt_jl_amd as tb

L = int(sys.argv[1]) if len(sys.argv) > 1 else 27
n = int(sys.argv[2]) if len(sys.argv) > 2 else 216
dev = tb.MI355XDevice(0)
torch.cuda.set_stream(torch.cuda.Stream()); dev.set_stream(torch.cuda.current_stream().cuda_stream)
g = tb.generate_mesh(tb.Hexahedron, (n, n, L), (0, 0, 0), (1, 1, L / n), perturb=0.2)
dh = tb.DofHandler(g)
src = tb.setup_operator(tb.AtomicAssemblyStrategy(dev), tb.LinearIntegrator(tb.AnalyticalCoefficient("cos_exp")), dh)
model = tb.TT06()
npts = dh.ndofs
u0 = np.tile(model.default_initial_state(), (npts, 1)); u0[:, model.phi_index] += np.linspace(0, 60, npts)
u = torch.from_numpy(np.ascontiguousarray(u0.T).ravel()).cuda()
f = tb.PointwiseODEFunction(npts, model)
cache = tb.setup_solver_cache(f, tb.ForwardEulerCellSolver(dev), u=u, keep_du=False)
e0, e1 = dev.event(), dev.event()
def timed(fn, reps=200):
    for _ in range(300): fn()          # (clocks: see scripts/_preroll.py)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); dev.synchronize()
    return e0.elapsed_ms(e1) / reps
tr = timed(lambda: tb.perform_step(f, cache, 0.0, 0.001))
ts = timed(lambda: tb.update_operator(src, 0.1))
print("L %d cells %d dofs %d reaction %.4f ms (%.1f ps/dof) source %.4f ms (%.1f ps/cell) env %s" % (L, g.n_cells, npts, tr, tr * 1e9 / npts, ts, ts * 1e9 / g.n_cells,
      {k: v for k, v in os.environ.items() if k.startswith("TB_") and k != "TB_LIBTBHIP"}))
