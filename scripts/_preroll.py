"""Untimed GPU work in front of a timed loop: after an idle period the GPU needs ~30 ms of kernels to return to full clocks
(profiles/r05_v1/clock_ramp_trace.txt), so every benchmark script runs its own workload for `ms` before it reads a clock."""
import time


def preroll(dev, fn, ms=150.0):
    import gc
    gc.collect(); gc.disable()   # and no collector pause (40–80 ms over a set-up's arrays) inside the loop that follows
    if ms <= 0:
        return
    dev.synchronize()
    t0 = time.perf_counter()
    fn(); fn()
    dev.synchronize()
    per = max((time.perf_counter() - t0) / 2, 1e-5)
    for _ in range(int(min(max(ms * 1e-3 / per, 1), 2000))):
        fn()
    dev.synchronize()
