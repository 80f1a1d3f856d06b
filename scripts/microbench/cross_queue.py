#!/usr/bin/env python3
"""What a fork / join between two HIP queues costs the queue that forks (one GPU).  A loop of a filler kernel (≈ 100 µs) with, per iteration, one of:
  A nothing   B fork only (event on main, side waits)   C fork + small side kernel + join at once   D fork, filler, join (the overlapped form)
  E RCCL exchange in the main queue (tb_comm_exchange, rank 0 as its own neighbours)   F tb_comm_exchange_begin, filler, tb_comm_exchange_end
  G as D with events created with release-to-device scope (hipEventDisableSystemFence) through the raw runtime
Prints µs per iteration above A (per pair of fillers for D / F / G, against two fillers)."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import torch
    import thunderbolt_jl_amd as tb
    torch.cuda.set_device(0)
    dev = tb.MI355XDevice(0)
    main_s = torch.cuda.Stream()
    torch.cuda.set_stream(main_s)
    dev.set_stream(main_s.cuda_stream)
    side = torch.cuda.Stream()
    y = torch.zeros(int(sys.argv[1]) if len(sys.argv) > 1 else 24_000_000, dtype=torch.float64, device="cuda")
    z = torch.zeros(47089, dtype=torch.float64, device="cuda")
    npl = 47089
    cm = tb.distributed.RcclComm(dev, 0, 1)
    sb = [torch.zeros(npl, dtype=torch.float64, device="cuda") for _ in range(2)]
    rb = [torch.empty(npl, dtype=torch.float64, device="cuda") for _ in range(2)]
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
    raw = []
    for _ in range(2):
        e = ctypes.c_void_p()
        assert hip.hipEventCreateWithFlags(ctypes.byref(e), 0x2 | 0x20000000) == 0      # hipEventDisableTiming | hipEventDisableSystemFence
        raw.append(e)
    e1, e2 = torch.cuda.Event(), torch.cuda.Event()

    def filler():
        y.add_(1.0)

    def A():
        filler()

    def B():
        filler(); e1.record(main_s); side.wait_event(e1)

    def C():
        filler(); e1.record(main_s); side.wait_event(e1)
        with torch.cuda.stream(side):
            z.add_(1.0)
        e2.record(side); main_s.wait_event(e2)

    def D():
        filler(); e1.record(main_s); side.wait_event(e1)
        with torch.cuda.stream(side):
            z.add_(1.0)
        e2.record(side)
        filler(); main_s.wait_event(e2)

    def E():
        filler(); cm.exchange([0, 0], sb, rb)

    def F():
        filler(); cm.exchange([0, 0], sb, rb, overlapped=True); filler(); cm.exchange_end()

    def G():
        filler(); hip.hipEventRecord(raw[0], main_s.cuda_stream); hip.hipStreamWaitEvent(side.cuda_stream, raw[0], 0)
        with torch.cuda.stream(side):
            z.add_(1.0)
        hip.hipEventRecord(raw[1], side.cuda_stream)
        filler(); hip.hipStreamWaitEvent(main_s.cuda_stream, raw[1], 0)

    def A2():
        filler(); filler()

    def timeit(fn, reps=300):
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    out = {}
    for rnd in range(2):
        a, a2 = timeit(A), timeit(A2)
        out["round%d" % rnd] = {"filler_us": a, "B_fork_only": timeit(B) - a, "C_fork_kernel_join_at_once": timeit(C) - a, "D_fork_filler_join": timeit(D) - a2,
                                "E_rccl_in_queue": timeit(E) - a, "F_rccl_begin_filler_end": timeit(F) - a2, "G_as_D_device_scope_events": timeit(G) - a2}
    cm.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
