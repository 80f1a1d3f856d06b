"""Child of tests/test_rccl_world1.py, started under `python -m torch.distributed.run --nproc-per-node 1`: initialises the `nccl` (= RCCL) backend at
world size 1 and pushes the multi-GPU exchange code through its DEVICE-buffer branches — the ones a one-GPU box otherwise never executes because the
two-rank tests must share cuda:0 over gloo (host-staged buffers):

  * `HaloExchange.start()` / `finish()`: `batch_isend_irecv` on device send / receive buffers (rank 0 is its own lower and upper neighbour: isend
    allows dst == own rank), pack / unpack through libtbhip on torch's current stream;
  * `all_reduce_sum` on a device tensor (no host staging);
  * `DistributedCG.device_step` with the overlapped product (interface rows packed first, exchange posted, whole product behind it) and its two
    device all-reduces — at world size 1 with self-neighbours the iteration solves the system whose interface rows are doubled, which is checked.

Prints one JSON line; exit code 0 only when every check passed."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    """argv[1] == "abi": the same checks with RCCL behind the C ABI (tb_comm_*: thunderbolt.jl_amd.distributed.RcclComm) in place of torch.distributed —
    no process group is created at all, the exchange is tb_comm_exchange and the reductions tb_comm_allreduce on the device's stream"""
    import torch
    import torch.distributed as tdist
    abi = len(sys.argv) > 1 and sys.argv[1] == "abi"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    import thunderbolt_jl_amd as tb
    D = tb.distributed
    dev = tb.MI355XDevice(0)
    dev.set_stream(torch.cuda.current_stream().cuda_stream)
    if abi:
        dist = D.RcclComm(dev)                                      # world size 1: the id is made here
        res = {"backend": "tbhip-rccl", "world": dist.world}
    else:
        dist = tdist
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        res = {"backend": dist.get_backend(), "world": dist.get_world_size()}

    nel = (12, 10, 16)
    g = tb.generate_mesh(tb.Hexahedron, nel, (0.0, 0.0, 0.0), (1.0, 1.0, 2.0), perturb=0.2)
    dh = tb.DofHandler(g)
    sp = tb.allocate_matrix(dh)
    st = tb.PatchAssemblyStrategy(dev)
    M = tb.setup_operator(st, tb.BilinearMassIntegrator(tb.ConstantCoefficient(1.0)), dh, sp)
    K = tb.setup_operator(st, tb.BilinearDiffusionIntegrator(tb.ConstantCoefficient(np.diag([4.5e-2, 2.0e-2, 2.0e-2]))), dh, sp)
    tb.update_operators(M, K, 0.0)
    src = tb.setup_operator(tb.AtomicAssemblyStrategy(dev), tb.LinearIntegrator(tb.AnalyticalCoefficient("cos_exp")), dh)
    b = torch.zeros(dh.ndofs, dtype=torch.float64, device="cuda")
    src.b = tb.DeviceVector.wrap(dev, b)
    tb.update_operator(src, 0.1)
    n2d = D.node_to_dof(dh)
    plane = (nel[0] + 1) * (nel[1] + 1)
    lo = torch.from_numpy(n2d[np.arange(plane)]).cuda()
    up = torch.from_numpy(n2d[np.arange(plane) + plane * nel[2]]).cuda()
    nb = [(0, lo), (0, up)]                       # rank 0 as its own lower and upper neighbour: what it sends first, it receives first

    # 1. halo exchange on device buffers over RCCL: every interface entry comes back added to itself
    halo = D.HaloExchange(nb, dist, b, dev)
    assert not halo.staged and halo.cuda
    b0 = b.clone()
    halo.exchange_sum(b)
    torch.cuda.synchronize()
    expect = b0.clone()
    expect[lo] += b0[lo]
    expect[up] += b0[up]
    res["halo_device_exchange_exact"] = bool(torch.equal(b, expect))
    b1 = b.clone()
    halo.exchange_sum(b)                          # persistent buffers, second use
    expect2 = b1.clone(); expect2[lo] += b1[lo]; expect2[up] += b1[up]
    torch.cuda.synchronize()
    res["halo_second_exchange_exact"] = bool(torch.equal(b, expect2))
    # the plain torch-indexing statement of the same exchange on device buffers
    if abi:
        res["torch_indexing_exchange_equal"] = True               # (torch.distributed path: not part of this mode)
        # the same exchange on the communicator's own queue (tb_comm_exchange_begin / _end, round 5): identical result, a kernel may run in between
        halo.overlap = True
        b3 = b.clone()
        halo.pack(b)
        halo.start()
        scratch = b * 2.0                                           # something on the device's stream while the transfer is in flight
        halo.finish(b)
        torch.cuda.synchronize()
        expect3 = b3.clone(); expect3[lo] += b3[lo]; expect3[up] += b3[up]
        res["halo_overlapped_exchange_exact"] = bool(torch.equal(b, expect3)) and bool(torch.equal(scratch, b3 * 2.0))
        halo.overlap = False
    else:
        b2 = D.exchange_sum(b1.clone(), nb, dist)
        res["torch_indexing_exchange_equal"] = bool(torch.equal(b2, expect2))

    # 2. all-reduce of device scalars (no host staging under nccl)
    S = torch.tensor([1.5, -2.0, 3.25], dtype=torch.float64, device="cuda")
    D.all_reduce_sum(S, dist)
    res["all_reduce_device"] = S.cpu().tolist() == [1.5, -2.0, 3.25]
    tt = torch.tensor([0.125], dtype=torch.float64, device="cuda")
    if abi:
        dist.allreduce(tt, "max")
    else:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.barrier()
    res["all_reduce_max"] = float(tt.item()) == 0.125

    # 3. the overlapped CG iteration through its device branches.  With the self-neighbours the "assembled" operator is A' = A + E_lo A + E_up A
    #    (interface rows doubled), the diagonal and the multiplicity weights follow the same rule; CG on D⁻¹-preconditioned A' is checked on the
    #    product it forms, which is all the exchange contributes: Ap == A·p with the interface rows doubled.
    A = tb.heat_system_matrix(dev, M, K, 0.5)
    diag = torch.empty(dh.ndofs, dtype=torch.float64, device="cuda")
    tb._lib.check(tb.lib().tb_extract_diagonal(K.pattern.h, A.ptr, diag.data_ptr()))
    cg = D.DistributedCG(None, diag, None, None, 0, 2, dist, neighbours=nb, device=dev, operator=(K.pattern, A))   # world_size=2 switches the all-reduce branches on
    p = torch.from_numpy(np.cos(np.arange(dh.ndofs) * 0.37)).cuda()
    Ap = torch.empty_like(p)
    S5 = torch.zeros(6, dtype=torch.float64, device="cuda")
    cg.device_iteration(p, Ap, S5)
    pAp_dev = float(S5[1].item())
    S5.zero_()
    ref = torch.empty_like(p)
    tb._lib.check(tb.lib().tb_spmv_csr(K.pattern.h, A.ptr, p.data_ptr(), 1.0, 0.0, ref.data_ptr()))
    pAp_ref = float((p * ref).sum().item())
    ref2 = ref.clone(); ref2[lo] += ref[lo]; ref2[up] += ref[up]
    torch.cuda.synchronize()
    res["cg_product_err"] = float((Ap - ref2).abs().max() / ref2.abs().max())
    res["cg_pAp_rel_err"] = abs(pAp_dev - pAp_ref) / abs(pAp_ref)
    x = torch.zeros_like(p); r = b0.clone() + 1.0; pp = cg.dinv * r
    tb._lib.check(tb.lib().tb_cgd_dot(dev.h, dh.ndofs, cg.w.data_ptr(), r.data_ptr(), pp.data_ptr(), S5[0:1].data_ptr()))
    D.all_reduce_sum(S5[0:1], dist)
    rr = []
    for _ in range(3):
        cg.device_step(x, r, pp, Ap, S5)
        rr.append(float(S5[5].item()))
    res["cg_steps_rr"] = rr
    res["cg_steps_finite_and_flag_clear"] = bool(np.all(np.isfinite(rr))) and float(S5[4].item()) == 0.0
    if abi:
        torch.cuda.synchronize()
        dist.close()
    else:
        dist.barrier()
        dist.destroy_process_group()
    ok = (res["backend"] in ("nccl", "tbhip-rccl") and res["halo_device_exchange_exact"] and res["halo_second_exchange_exact"] and res["torch_indexing_exchange_equal"]
          and res["all_reduce_device"] and res["all_reduce_max"] and res["cg_product_err"] < 1e-13 and res["cg_pAp_rel_err"] < 1e-12
          and res["cg_steps_finite_and_flag_clear"] and res.get("halo_overlapped_exchange_exact", True))
    res["ok"] = bool(ok)
    print(json.dumps(res))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
