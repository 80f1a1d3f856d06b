"""Multi-GPU sharding of the hot path (new work: the reference is shared-memory only, README.md:7).

Cells are partitioned into P contiguous z-slabs (structured boxes); every rank assembles its own
sub-domain operators with NO communication; the only exchange is the sum of vector entries on the
interface planes shared by neighbouring slabs (chain topology → neighbour send/recv, or one
all-reduce of the packed interface vector).  The reaction step owns its points and never communicates.
"""
import numpy as np


def slab_range(nz, world_size, rank):
    """[z0, z1) layers of rank `rank` when nz layers are split as evenly as possible."""
    base, rem = divmod(nz, world_size)
    z0 = rank * base + min(rank, rem)
    return z0, z0 + base + (1 if rank < rem else 0)


class SlabPartition:
    """Local sub-grid of a global (nx, ny, nz) hex box for one rank, with interface bookkeeping.

    local node (i,j,k) ↔ global node (i,j,k+z0); interface planes are the local k=0 plane (shared with
    rank-1) and the local k=nzl plane (shared with rank+1)."""

    def __init__(self, nel, left, right, world_size, rank):
        self.nel, self.world_size, self.rank = tuple(nel), world_size, rank
        nx, ny, nz = nel
        self.z0, self.z1 = slab_range(nz, world_size, rank)
        self.nzl = self.z1 - self.z0
        lz, rz = left[2], right[2]
        self.left = (left[0], left[1], lz + (rz - lz) * (self.z0 / nz))
        self.right = (right[0], right[1], rz if self.z1 == nz else lz + (rz - lz) * (self.z1 / nz))
        self.plane = (nx + 1) * (ny + 1)

    def local_nel(self):
        return (self.nel[0], self.nel[1], self.nzl)

    def interface_nodes(self):
        """(lower, upper): local node ids on the planes shared with rank-1 / rank+1 (None at the ends)."""
        lower = np.arange(self.plane, dtype=np.int64) if self.rank > 0 else None
        upper = (np.arange(self.plane, dtype=np.int64) + self.plane * self.nzl) if self.rank < self.world_size - 1 else None
        return lower, upper


def node_to_dof(dh):
    """dof id of every mesh node for a first-order scalar field (dof id ≠ node id in Ferrite numbering)."""
    n2d = np.full(dh.grid.n_nodes, -1, dtype=np.int64)
    n2d[dh.grid.conn.ravel()] = dh.cell_dofs.ravel()
    return n2d


def halo_sum(vec, lower_idx, upper_idx, rank, world_size, dist):
    """Sum interface entries of `vec` (torch tensor, any device) with both neighbours, in place.

    After the call every rank holds the globally assembled value on its interface dofs.  Uses
    neighbour isend/irecv (RCCL over xGMI on GPU, gloo on CPU)."""
    import torch

    ops, bufs = [], []
    for idx, peer in ((lower_idx, rank - 1), (upper_idx, rank + 1)):
        if idx is None or peer < 0 or peer >= world_size:
            continue
        send = vec[idx].contiguous()
        recv = torch.empty_like(send)
        ops.append(dist.P2POp(dist.isend, send, peer))
        ops.append(dist.P2POp(dist.irecv, recv, peer))
        bufs.append((idx, recv))
    if ops:
        for r in dist.batch_isend_irecv(ops):
            r.wait()
    for idx, recv in bufs:
        vec[idx] += recv
    return vec


class DistributedCG:
    """Jacobi-preconditioned CG on sub-domain (interface-unassembled) matrices A = Σ_p R_pᵀ A_p R_p.

    Vectors are torch tensors holding every dof of the slab (owned + interface); solution-type vectors are kept
    *consistent* (both neighbours hold the same interface value), operator results are summed over the interface
    with `halo_sum`.  Dot products weight interface dofs by 1/multiplicity and are all-reduced (one scalar
    all-reduce — RCCL on GPU).  `local_spmv(x) -> y` applies the rank's own A_p (tb_spmv_csr on device; any callable
    in tests).  New work: the reference has no distributed solver (README.md:7)."""

    def __init__(self, local_spmv, local_diag, lower_idx, upper_idx, rank, world_size, dist):
        import torch
        self.torch, self.dist = torch, dist
        self.spmv, self.rank, self.world = local_spmv, rank, world_size
        self.lo, self.up = lower_idx, upper_idx
        d = local_diag.clone()
        halo_sum(d, self.lo, self.up, rank, world_size, dist)      # assembled diagonal
        self.dinv = 1.0 / d
        w = torch.ones_like(d)
        for idx in (self.lo, self.up):
            if idx is not None:
                w[idx] = 0.5                                      # slab interfaces are shared by exactly two ranks
        self.w = w

    def dot(self, a, b):
        s = (self.w * a * b).sum().reshape(1)
        if self.world > 1:
            self.dist.all_reduce(s)
        return float(s.item())

    def apply(self, x):
        y = self.spmv(x)
        halo_sum(y, self.lo, self.up, self.rank, self.world, self.dist)
        return y

    def solve(self, b, x, rtol=1e-5, atol=1e-6, maxiter=1000):
        """b: ASSEMBLED right-hand side (consistent), x: initial guess (consistent); returns (x, iterations, ‖r‖)."""
        r = b - self.apply(x)
        z = self.dinv * r
        p = z.clone()
        rz = self.dot(r, z)
        rn = self.dot(r, r) ** 0.5
        tol = atol + rtol * rn
        it = 0
        while rn > tol and it < maxiter:
            Ap = self.apply(p)
            alpha = rz / self.dot(p, Ap)
            x += alpha * p
            r -= alpha * Ap
            z = self.dinv * r
            rz_new = self.dot(r, z)
            rn = self.dot(r, r) ** 0.5
            p = z + (rz_new / rz) * p
            rz = rz_new
            it += 1
        return x, it, rn
