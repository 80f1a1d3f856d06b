#!/usr/bin/env python3
"""RCCL at world size 1 behind the C ABI: stream time of the step's halo exchange (rank 0 as its own lower and upper neighbour, one interface plane of
PLANE doubles each way) and of the two-double all-reduce of a CG iteration.  Prints one JSON line.  bench.py runs this in a child process (a crash or
a hang of the communication library must not cost the bench line) and adds the figures to slab_sweep (`exchange_latency`).  No xGMI hop is in them —
one GPU — but the launch, the protocol and the copy kernels are."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    npl = int(sys.argv[1]) if len(sys.argv) > 1 else 217 * 217
    import torch
    import thunderbolt_jl_amd as tb
    torch.cuda.set_device(0)
    dev = tb.MI355XDevice(0)
    torch.cuda.set_stream(torch.cuda.Stream())
    dev.set_stream(torch.cuda.current_stream().cuda_stream)
    cm = tb.distributed.RcclComm(dev, 0, 1)
    sb = [torch.zeros(npl, dtype=torch.float64, device="cuda") for _ in range(2)]
    rb = [torch.empty(npl, dtype=torch.float64, device="cuda") for _ in range(2)]
    sc = torch.zeros(2, dtype=torch.float64, device="cuda")
    for _ in range(5):
        cm.exchange([0, 0], sb, rb)
        cm.allreduce(sc)
    torch.cuda.synchronize()
    e = [dev.event() for _ in range(4)]
    reps = 20
    e[0].record()
    for _ in range(reps):
        cm.exchange([0, 0], sb, rb)
    e[1].record()
    for _ in range(reps):
        cm.allreduce(sc)
    e[2].record()
    for _ in range(reps):                                           # the overlapped form: begin … end with nothing in between = its pure cost on the device's stream
        cm.exchange([0, 0], sb, rb, overlapped=True)
        cm.exchange_end()
    e[3].record()
    torch.cuda.synchronize()
    # the whole halo sum of a step as HaloExchange runs it: one gather (both planes), the grouped send / receive in the device's queue, one scatter-add
    vec = torch.zeros(4 * npl, dtype=torch.float64, device="cuda")
    hx = tb.distributed.HaloExchange([(0, torch.arange(npl, device="cuda")), (0, torch.arange(3 * npl, 4 * npl, device="cuda"))], cm, vec, dev)
    for _ in range(5):
        hx.exchange_sum(vec)
    e4, e5 = dev.event(), dev.event()
    e4.record()
    for _ in range(reps):
        hx.exchange_sum(vec)
    e5.record()
    torch.cuda.synchronize()
    out = {"halo_sum_ms": e4.elapsed_ms(e5) / reps, "halo_exchange_ms": e[0].elapsed_ms(e[1]) / reps, "allreduce_ms": e[1].elapsed_ms(e[2]) / reps, "halo_exchange_begin_end_ms": e[2].elapsed_ms(e[3]) / reps,
           "plane_doubles": npl,
           "note": "RCCL behind the C ABI at world size 1 (tb_comm_exchange with rank 0 as its own two neighbours, tb_comm_allreduce of 2 doubles; halo_sum = gather + exchange + scatter-add through HaloExchange): stream time per call, no xGMI hop"}
    cm.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
