"""Thread scaling of the CPU oracle (baseline leg) on the box's host cores."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as o
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
xyz, conn = o.generate_grid_hex(n, n, n, (0, 0, 0), (1, 1, 1))
cd, nd = o.close_dofs(o.HEX8, 1, conn, len(xyz))
rp, ci = o.build_pattern(cd, nd)
col, nc = o.color_cells(cd, nd)
m = o.Mesh(o.HEX8, 2, xyz, conn, cd)
cK = o.Coef(o.COEF_CONST_TENSOR, np.diag([4.5e-5, 2e-5, 2e-5]).ravel())
p = o.cell_default_params(o.CELL_PCG2019)
u = np.ascontiguousarray(np.tile(o.cell_default_state(o.CELL_PCG2019, p), (nd, 1)).T).ravel().copy()
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for th in (1, 4, 16, 32, 64, 128, 256):
    best = [1e9, 1e9, 1e9]
    for rep in range(3):
        t0 = time.perf_counter(); o.assemble_matrix(m, 1, cK, rp, ci, nthreads=th, color=col, ncolors=nc)
        t1 = time.perf_counter(); o.assemble_source(m, o.SRC_COS_EXP, nthreads=th)
        t2 = time.perf_counter(); o.reaction_step(o.CELL_PCG2019, p, u, nd, dt=0.01, nthreads=th, want_du=False)
        t3 = time.perf_counter()
        best = [min(best[0], t1 - t0), min(best[1], t2 - t1), min(best[2], t3 - t2)]
    print("threads %3d  K %.3e cells/s  b %.3e cells/s  reaction %.3e dof-upd/s" % (th, n**3 / best[0], n**3 / best[1], 7 * nd / best[2]), flush=True)
