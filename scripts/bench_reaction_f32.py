"""Reaction step with Float32 storage (tb_reaction_step_f32: one pass, Float64 arithmetic) against Float64 storage, 10.2 M points."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
lib, check = tb.lib(), tb._lib.check
n = 217 ** 3
for cls, dt in (("FHNModel", 0.1), ("PCG2019", 0.01), ("TT06", 0.001)):
    m = getattr(tb, cls)()
    ns = m.nstates
    host = np.ascontiguousarray(np.tile(m.default_initial_state(), (n, 1)).T).ravel()
    par = m.params.ctypes.data_as(tb._lib.c_dp)
    u64, u32 = dev.to_device(host), dev.to_device(host.astype(np.float32))
    def f64(): check(lib.tb_reaction_step(dev.h, m.model_id, par, len(m.params), u64.ptr, None, n, ns, 0, 0.0, dt, 1, 0.0))
    def f32(): check(lib.tb_reaction_step_f32(dev.h, m.model_id, par, len(m.params), u32.ptr, None, n, ns, 0, None, 0, 0.0, dt, 1, 0.0))
    out = {}
    for name, fn in (("f64", f64), ("f32", f32)):
        fn(); fn()
        a, b = dev.event(), dev.event()
        a.record()
        for _ in range(10):
            fn()
        b.record(); dev.synchronize()
        out[name] = a.elapsed_ms(b) / 10
    print("%-9s %2d states: Float64 storage %.3f ms, Float32 storage %.3f ms (%.2fx)" % (cls, ns, out["f64"], out["f32"], out["f64"] / out["f32"]))
