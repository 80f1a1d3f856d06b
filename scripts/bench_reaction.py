#!/usr/bin/env python3
"""One reaction step at 10.2 M points (BASELINE config 3's size): forward Euler (tb_reaction_step) and Rush–Larsen (tb_reaction_step_rl) for the
models given; algorithmic bytes = 16 B per state and point.  TB_REACTION_PIPE=0 in the environment switches the next-point requests of the
one-wave-per-SIMD kernels (O'Hara–Rudy) off.  Prints one JSON line."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
lib, check = tb.lib(), tb._lib.check
n = 217 ** 3
out = {"points": n, "pipe": os.environ.get("TB_REACTION_PIPE", "1")}
for cls, dt in [(c, d) for c, d in (("PCG2019", 0.01), ("TT06", 0.001), ("ORd2011", 0.001)) if len(sys.argv) < 2 or c in sys.argv[1:]]:
    m = getattr(tb, cls)()
    ns = m.nstates
    host = np.ascontiguousarray(np.tile(m.default_initial_state(), (n, 1)).T).ravel()
    par = m.params.ctypes.data_as(tb._lib.c_dp)
    u = dev.to_device(host)
    def fe(): check(lib.tb_reaction_step(dev.h, m.model_id, par, len(m.params), u.ptr, None, n, ns, 0, 0.0, dt, 1, 0.0))
    def rl(): check(lib.tb_reaction_step_rl(dev.h, m.model_id, par, len(m.params), u.ptr, n, ns, 0, 0.0, dt))
    for name, fn in (("fe", fe), ("rl", rl)):
        fn(); fn()
        from _preroll import preroll
        preroll(dev, fn, 80.0)   # steady clocks
        a, b = dev.event(), dev.event()
        a.record()
        for _ in range(10):
            fn()
        b.record(); dev.synchronize()
        ms = a.elapsed_ms(b) / 10
        out["%s_%s" % (cls, name)] = {"ms": ms, "hbm_frac": 16.0 * ns * n / (ms * 1e-3) / 8e12}
    del u
print(json.dumps(out))
