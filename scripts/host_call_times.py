"""Host-side duration of every ABI call of a bench step (no synchronize in between): a call that takes as long as a kernel is waiting for the GPU."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import thunderbolt_jl_amd as tb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
L = int(sys.argv[2]) if len(sys.argv) > 2 else n
dev = tb.MI355XDevice(0)
torch.cuda.set_stream(torch.cuda.Stream()); dev.set_stream(torch.cuda.current_stream().cuda_stream)
g = tb.generate_mesh(tb.Hexahedron, (n, n, L), (0, 0, 0), (1, 1, L / n), perturb=0.2)
dh = tb.DofHandler(g); sp = tb.allocate_matrix(dh)
st = tb.PatchAssemblyStrategy(dev)
kap = np.array([[4.5e-5, 1e-5, 0], [1e-5, 2.0e-5, 0], [0, 0, 2.0e-5]])
M = tb.setup_operator(st, tb.BilinearMassIntegrator(tb.ConstantCoefficient(1.0)), dh, sp)
K = tb.setup_operator(st, tb.BilinearDiffusionIntegrator(tb.ConstantCoefficient(kap)), dh, sp)
src = tb.setup_operator(tb.AtomicAssemblyStrategy(dev), tb.LinearIntegrator(tb.AnalyticalCoefficient("cos_exp")), dh)
model = tb.TT06(); npts = dh.ndofs
u0 = np.tile(model.default_initial_state(), (npts, 1)); u0[:, model.phi_index] += np.linspace(0, 60, npts)
u = torch.from_numpy(np.ascontiguousarray(u0.T).ravel()).cuda()
f = tb.PointwiseODEFunction(npts, model)
cache = tb.setup_solver_cache(f, tb.ForwardEulerCellSolver(dev), u=u, keep_du=False)
def step(t, rec=None):
    a = time.perf_counter(); tb.update_operators(M, K, t)
    b = time.perf_counter(); tb.update_operator(src, t)
    c = time.perf_counter(); tb.perform_step(f, cache, t, 0.001)
    d = time.perf_counter()
    if rec is not None: rec.append((b - a, c - b, d - c))
for i in range(3): step(0.01 * i)
dev.synchronize()
for defer in (False, True):
    dev.defer_status(defer)
    rec = []
    t0 = time.perf_counter()
    for i in range(20): step(0.01 * i, rec)
    t1 = time.perf_counter(); dev.synchronize(); t2 = time.perf_counter()
    r = np.array(rec) * 1e6
    print("   first steps (us):", [tuple(round(float(x), 1) for x in row) for row in r[:6]], " last:", tuple(round(float(x), 1) for x in r[-1]))
    print("defer_status %s: host enqueue per step %.1f us (M+K %.1f, source %.1f, reaction %.1f; max single call %.1f), wall per step %.1f us" %
          (defer, (t1 - t0) / 20 * 1e6, r[:, 0].mean(), r[:, 1].mean(), r[:, 2].mean(), r.max(), (t2 - t0) / 20 * 1e6))
dev.defer_status(False)

# the three calls of a step touch disjoint data (matrices | b | states): one stream each, no joins — what do the queues overlap?
streams = [torch.cuda.Stream() for _ in range(3)]
def step3(t):
    dev.set_stream(streams[0].cuda_stream); tb.update_operators(M, K, t)
    dev.set_stream(streams[1].cuda_stream); tb.update_operator(src, t)
    dev.set_stream(streams[2].cuda_stream); tb.perform_step(f, cache, t, 0.001)
dev.defer_status(True)
for name, fn in (("one stream", lambda t: step(t)), ("three streams", step3), ("one stream", lambda t: step(t)), ("three streams", step3)):
    for i in range(3): fn(0.01 * i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40): fn(0.01 * i)
    torch.cuda.synchronize()
    print("%-13s wall per step %.1f us" % (name, (time.perf_counter() - t0) / 40 * 1e6))
    dev.set_stream(torch.cuda.current_stream().cuda_stream)
dev.defer_status(False)
