import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
rng = np.random.default_rng(3)
for n in (8, 9, 1000, 1001, 4096):
    a, b = rng.normal(size=n), rng.normal(size=n)
    out, da, db = dev.zeros(n), dev.to_device(a), dev.to_device(b)
    tb._lib.check(tb.lib().tb_heat_matrix(dev.h, n, da.ptr, db.ptr, 0.3, out.ptr))
    o = out.to_host()
    print(n, np.abs(o - (a - 0.3 * b)).max(), np.abs(o - a).max(), np.abs(o - b).max(), np.abs(da.to_host() - a).max(), o[:3], (a - 0.3 * b)[:3])
