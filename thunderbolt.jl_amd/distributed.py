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
