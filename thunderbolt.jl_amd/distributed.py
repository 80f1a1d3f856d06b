"""Multi-GPU sharding of the hot path (new work: the reference is shared-memory only, README.md:7).

Cells are partitioned into P contiguous z-slabs (structured boxes) or, for general meshes, by recursive
coordinate bisection of the cell centroids; every rank assembles its own sub-domain operators with NO
communication; the only exchange is the sum of vector entries on the dofs shared with neighbouring parts
(neighbour send/recv — chain topology for slabs, an arbitrary neighbour set for general parts).  The reaction
step owns its points and never communicates.
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


def partition_cells_rcb(centroids, nparts):
    """Recursive coordinate bisection: part id per cell.  Splits along the longest extent, proportionally when the
    number of parts is odd, so every part receives ⌊n/P⌋ or ⌈n/P⌉ cells.  Deterministic (stable sorts)."""
    centroids = np.asarray(centroids, dtype=np.float64)
    part = np.zeros(len(centroids), dtype=np.int32)

    def split(idx, p0, np_):
        if np_ == 1 or len(idx) == 0:
            part[idx] = p0
            return
        c = centroids[idx]
        axis = int(np.argmax(c.max(axis=0) - c.min(axis=0)))
        order = idx[np.argsort(c[:, axis], kind="stable")]
        nl = np_ // 2
        cut = (len(idx) * nl) // np_
        split(order[:cut], p0, nl)
        split(order[cut:], p0 + nl, np_ - nl)

    split(np.arange(len(centroids)), 0, int(nparts))
    return part


class GeneralPartition:
    """One rank's share of an arbitrary mesh: the cells with part id == rank, their nodes renumbered locally (first
    visit), and for every other rank the nodes shared with it — both sides list them in ascending *global* node id,
    so the packed exchange buffers line up without any negotiation."""

    def __init__(self, conn, part, rank):
        conn, part = np.asarray(conn), np.asarray(part)
        self.rank = int(rank)
        self.cells = np.flatnonzero(part == rank)
        gl = conn[self.cells]
        self.global_nodes, first = np.unique(gl.ravel(), return_index=True)
        order = np.argsort(first, kind="stable")                    # local ids by first visit
        self.global_nodes = self.global_nodes[order]
        lookup = {int(g): i for i, g in enumerate(self.global_nodes)}
        self.conn = np.vectorize(lookup.__getitem__, otypes=[np.int32])(gl) if gl.size else gl.astype(np.int32)
        mine = set(lookup)
        self.neighbours = []                                        # [(peer rank, local node ids)], peers ascending
        for q in sorted(set(part.tolist()) - {self.rank}):
            shared = sorted(mine.intersection(np.unique(conn[part == q]).tolist()))
            if shared:
                self.neighbours.append((int(q), np.array([lookup[g] for g in shared], dtype=np.int64)))

    def multiplicity(self):
        m = np.ones(len(self.global_nodes))
        for _, idx in self.neighbours:
            m[idx] += 1.0
        return m


class RcclComm:
    """RCCL communicator behind the C ABI (tb_comm_*): the exchange and the all-reduces of the multi-GPU path as the Julia host would drive them — no
    torch.distributed on the data path.  One per process (one process per GPU).  `id128`: the 128 bytes of tb_comm_unique_id from rank 0 (None at world
    size 1: made here); `from_torch` carries them with an existing torch.distributed group (any backend), which is then needed for nothing else.
    Pass an instance wherever HaloExchange / DistributedCG / all_reduce_sum take `dist`."""

    def __init__(self, device, rank=0, world_size=1, id128=None):
        import ctypes as C
        from ._lib import check, lib
        self.dev, self.rank, self.world = device, int(rank), int(world_size)
        if id128 is None:
            if self.world != 1:
                raise ValueError("RcclComm: ranks > 0 need the communicator id made by rank 0 (RcclComm.unique_id)")
            id128 = RcclComm.unique_id()
        self._id = bytes(id128)
        buf = C.create_string_buffer(self._id, 128)
        self.h = C.c_void_p()
        check(lib().tb_comm_create(device.h, buf, self.rank, self.world, C.byref(self.h)))

    @staticmethod
    def unique_id():
        import ctypes as C
        from ._lib import check, lib
        buf = C.create_string_buffer(128)
        check(lib().tb_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_torch(cls, device, dist):
        """communicator over the ranks of the default torch.distributed group: rank 0 makes the id, the group broadcasts it"""
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [None]
        if rank == 0:                                                # a failure on rank 0 travels with the broadcast: every rank raises, none waits
            try:
                box[0] = RcclComm.unique_id()
            except Exception as ex:
                box[0] = "RcclComm.unique_id on rank 0: %s" % ex
        dist.broadcast_object_list(box, src=0)
        if not isinstance(box[0], (bytes, bytearray)):
            raise RuntimeError(str(box[0]))
        return cls(device, rank, world, box[0])

    def exchange(self, peers, send, recv, overlapped=False):
        """one grouped send / receive with every listed peer (device tensors of equal length per peer), enqueued on the device's stream — or, with
        overlapped=True, on the communicator's own queue (tb_comm_exchange_begin): what the device's stream does until `exchange_end()` runs beside it"""
        import ctypes as C
        from ._lib import check, lib
        n = len(peers)
        if n == 0:
            return
        P = (C.c_int32 * n)(*[int(p) for p in peers])
        cnt = (C.c_int64 * n)(*[int(t.numel()) for t in send])
        S = (C.c_void_p * n)(*[t.data_ptr() for t in send])
        R = (C.c_void_p * n)(*[t.data_ptr() for t in recv])
        check((lib().tb_comm_exchange_begin if overlapped else lib().tb_comm_exchange)(self.h, n, P, cnt, S, R))

    def exchange_end(self):
        from ._lib import check, lib
        check(lib().tb_comm_exchange_end(self.h))

    def allreduce(self, t, op="sum"):
        import ctypes as C
        from ._lib import check, lib
        assert t.is_cuda and t.is_contiguous() and t.dtype.itemsize == 8
        check(lib().tb_comm_allreduce(self.h, C.c_void_p(t.data_ptr()), t.numel(), 0 if op == "sum" else 1))
        return t

    def close(self):
        from ._lib import lib
        if self.h:
            lib().tb_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _host_staged(vec, dist):
    """gloo moves host buffers only: device tensors are staged through the host (the CPU-backend test configuration; RCCL sends device
    buffers directly)."""
    return vec.is_cuda and not isinstance(dist, RcclComm) and dist.get_backend() == "gloo"


def exchange_sum(vec, neighbours, dist):
    """Sum the entries of `vec` (torch tensor, any device) shared with each neighbour, in place: every rank sends its
    own *partial* values of the shared dofs to each sharing peer and adds what it receives, so a dof held by k ranks
    ends up with the sum of all k partials on each of them.  neighbours = [(peer, index tensor)].

    Plain torch indexing with fresh buffers per call: the statement of the exchange that `HaloExchange` (persistent buffers, pack / unpack
    through the C ABI) is checked against bit for bit, and the path of host tensors."""
    import torch

    staged = _host_staged(vec, dist)
    ops, bufs = [], []
    for peer, idx in neighbours:
        send = vec[idx].contiguous()                                # partials, taken before anything is added
        if staged:
            send = send.cpu()
        recv = torch.empty_like(send)
        ops.append(dist.P2POp(dist.isend, send, peer))
        ops.append(dist.P2POp(dist.irecv, recv, peer))
        bufs.append((idx, recv, send))
    if ops:
        for r in dist.batch_isend_irecv(ops):
            r.wait()
    for idx, recv, _ in bufs:
        vec[idx] += recv.to(vec.device) if staged else recv
    return vec


class HaloExchange:
    """The neighbour exchange of one sub-domain with persistent buffers, behind the C ABI: per neighbour an Int32 index list on the device, a
    send and a receive buffer allocated once; pack = tb_gather_indexed (or tb_spmv_csr_rows: the interface rows of a product, computed straight
    into the send buffer), unpack = tb_scatter_add_indexed.  `start()` posts the sends / receives and returns; `finish(vec)` waits and adds — what
    is launched in between (the interior SpMV) overlaps the transfer.  Device tensors + an MI355XDevice use libtbhip; host tensors (the gloo
    tests) use the equivalent torch calls on the same persistent buffers.  Under gloo with device tensors the buffers are staged through pinned
    host memory (one-GPU test configuration)."""

    def __init__(self, neighbours, dist, like, device=None):
        import torch
        self.torch, self.dist, self.dev = torch, dist, device
        # device tensors + an MI355XDevice: pack / unpack through libtbhip on the device's stream; device tensors without one (plain torch callers):
        # the equivalent torch calls on the same persistent buffers, like host tensors
        self.cuda = bool(like.is_cuda) and device is not None
        self.abi = isinstance(dist, RcclComm)                       # exchange through tb_comm_exchange (RCCL behind the C ABI) instead of torch.distributed
        self.overlap = False                                         # ABI path: exchange on the communicator's own queue (see start())
        if self.abi and not self.cuda:
            raise ValueError("HaloExchange: an RcclComm moves device buffers of an MI355XDevice")
        self.staged = bool(like.is_cuda) and dist is not None and not self.abi and dist.is_initialized() and dist.get_backend() == "gloo"
        self.peers, self.idx, self.idx32, self.send, self.recv, self.send_h, self.recv_h = [], [], [], [], [], [], []
        for peer, idx in neighbours:
            idx = torch.as_tensor(idx, dtype=torch.int64, device=like.device)
            self.peers.append(int(peer))
            self.idx.append(idx)
            self.idx32.append(idx.to(torch.int32).contiguous())
            self.send.append(torch.empty(idx.numel(), dtype=like.dtype, device=like.device))
            self.recv.append(torch.empty(idx.numel(), dtype=like.dtype, device=like.device))
            if self.staged:
                self.send_h.append(torch.empty(idx.numel(), dtype=like.dtype).pin_memory())
                self.recv_h.append(torch.empty(idx.numel(), dtype=like.dtype).pin_memory())
        # several neighbours whose index lists share no dof (the two planes of a slab): ONE gather fills every send buffer and ONE scatter-add takes every
        # receive buffer — the buffers become slices of one allocation each.  A 47 000-entry gather is all launch: 4–6 µs of the queue per call, and a step
        # of a thin slab is 0.4 ms (profiles/r06_v1/slab27_self_exchange_timeline_abi.txt).  Lists that share dofs (a corner of a general partition) keep
        # one call per neighbour: tb_scatter_add_indexed wants distinct indices within a call.
        self.cat_idx32 = None
        import os
        if self.cuda and len(self.peers) > 1 and not os.environ.get("TB_HALO_SEPARATE_CALLS"):     # (the switch: tests compare the two forms)
            cat = torch.cat(self.idx)
            if torch.unique(cat).numel() == cat.numel():
                self.cat_idx32 = cat.to(torch.int32).contiguous()
                self.cat_send = torch.empty(cat.numel(), dtype=like.dtype, device=like.device)
                self.cat_recv = torch.empty(cat.numel(), dtype=like.dtype, device=like.device)
                off = 0
                for k, idx in enumerate(self.idx):
                    self.send[k] = self.cat_send[off:off + idx.numel()]
                    self.recv[k] = self.cat_recv[off:off + idx.numel()]
                    off += idx.numel()
        self.reqs = []

    @property
    def nbytes(self):
        return sum(b.numel() * b.element_size() for b in self.send)

    def _ptr(self, t):
        import ctypes as C
        return C.c_void_p(t.data_ptr())

    def pack(self, vec):
        """send buffers ← the partial values of the shared dofs"""
        if self.cuda:
            from ._lib import check, lib
            if self.cat_idx32 is not None:
                check(lib().tb_gather_indexed(self.dev.h, self.cat_idx32.numel(), self._ptr(vec), self._ptr(self.cat_idx32), self._ptr(self.cat_send)))
                return
            for idx32, send in zip(self.idx32, self.send):
                check(lib().tb_gather_indexed(self.dev.h, idx32.numel(), self._ptr(vec), self._ptr(idx32), self._ptr(send)))
        else:
            for idx, send in zip(self.idx, self.send):
                self.torch.index_select(vec, 0, idx, out=send)

    def pack_product_rows(self, pattern, nz, x):
        """send buffers ← the interface rows of A·x (device path only): pattern / nz are the sub-domain CSR operator"""
        from ._lib import check, lib
        for idx32, send in zip(self.idx32, self.send):
            check(lib().tb_spmv_csr_rows(pattern.h, nz.ptr, self._ptr(x), idx32.numel(), self._ptr(idx32), self._ptr(send)))

    def start(self):
        if not self.peers:
            return
        if self.abi:
            # stream-ordered by default.  overlap=True puts the exchange on the communicator's own queue (tb_comm_exchange_begin / _end) so that the kernels
            # launched until finish() run beside it — measured at world size 1 (scripts/rccl_latency.py): 0.061 ms for begin + end against 0.012 ms for the
            # in-stream exchange of one 377 KB plane each way: the two cross-queue event waits cost more than a transfer of this size takes, so the
            # overlapped form only pays for much larger interfaces
            self.dist.exchange(self.peers, self.send, self.recv, overlapped=self.overlap)
            return
        dist, ops = self.dist, []
        if self.staged:
            for s, sh in zip(self.send, self.send_h):
                sh.copy_(s, non_blocking=True)
            self.torch.cuda.current_stream().synchronize()
        for k, peer in enumerate(self.peers):
            ops.append(dist.P2POp(dist.isend, self.send_h[k] if self.staged else self.send[k], peer))
            ops.append(dist.P2POp(dist.irecv, self.recv_h[k] if self.staged else self.recv[k], peer))
        self.reqs = dist.batch_isend_irecv(ops)

    def finish(self, vec, own_from_send=False):
        """wait for the transfers, then vec[shared dofs] += received partials.  own_from_send: first vec[shared dofs] = the send buffers (the
        overlapped product: the rows this rank sent were formed by tb_spmv_csr_rows, the peer's too — replacing the stream kernel's interface rows
        by them makes both sides add the same two numbers, so a dof shared by two ranks holds the same bits on both)"""
        for r in self.reqs:
            r.wait()
        self.reqs = []
        if self.abi and self.peers and self.overlap:
            self.dist.exchange_end()                                 # the device's stream waits for the transfer (no host wait)
        if self.staged:
            for r_, rh in zip(self.recv, self.recv_h):
                r_.copy_(rh, non_blocking=True)
        if own_from_send:
            if self.cuda:
                from ._lib import check, lib
                for idx32, send in zip(self.idx32, self.send):
                    check(lib().tb_scatter_indexed(self.dev.h, idx32.numel(), self._ptr(send), self._ptr(idx32), self._ptr(vec)))
            else:
                for idx, send in zip(self.idx, self.send):
                    vec[idx] = send
        if self.cuda:
            from ._lib import check, lib
            if self.cat_idx32 is not None:
                check(lib().tb_scatter_add_indexed(self.dev.h, self.cat_idx32.numel(), self._ptr(self.cat_recv), self._ptr(self.cat_idx32), self._ptr(vec)))
                return vec
            for idx32, recv in zip(self.idx32, self.recv):
                check(lib().tb_scatter_add_indexed(self.dev.h, idx32.numel(), self._ptr(recv), self._ptr(idx32), self._ptr(vec)))
        else:
            for idx, recv in zip(self.idx, self.recv):
                vec.index_add_(0, idx, recv)
        return vec

    def exchange_sum(self, vec):
        self.pack(vec)
        self.start()
        return self.finish(vec)


def all_reduce_sum(t, dist):
    """Sum a (small) tensor over the ranks in place; device tensors go through the host under gloo."""
    if isinstance(dist, RcclComm):
        return dist.allreduce(t.contiguous() if not t.is_contiguous() else t, "sum")
    if _host_staged(t, dist):
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
    else:
        dist.all_reduce(t)
    return t


def node_to_dof(dh):
    """dof id of every mesh node for a first-order scalar field (dof id ≠ node id in Ferrite numbering)."""
    n2d = np.full(dh.grid.n_nodes, -1, dtype=np.int64)
    n2d[dh.grid.conn.ravel()] = dh.cell_dofs.ravel()
    return n2d


def slab_neighbours(lower_idx, upper_idx, rank, world_size):
    """[(peer, index tensor)] of a z-slab: the plane shared with rank − 1 and the one shared with rank + 1"""
    return [(peer, idx) for idx, peer in ((lower_idx, rank - 1), (upper_idx, rank + 1)) if idx is not None and 0 <= peer < world_size]


def halo_sum(vec, lower_idx, upper_idx, rank, world_size, dist):
    """Sum interface entries of `vec` (torch tensor, any device) with both neighbours, in place.

    After the call every rank holds the globally assembled value on its interface dofs.  Uses
    neighbour isend/irecv (RCCL over xGMI on GPU, gloo on CPU)."""
    return exchange_sum(vec, slab_neighbours(lower_idx, upper_idx, rank, world_size), dist)


class DistributedCG:
    """Jacobi-preconditioned CG on sub-domain (interface-unassembled) matrices A = Σ_p R_pᵀ A_p R_p.

    Vectors are torch tensors holding every dof of the part (owned + interface); solution-type vectors are kept
    *consistent* (all sharing ranks hold the same interface value), operator results are summed over the interface
    by a `HaloExchange`.  Consistency is bitwise for dofs shared by two ranks (every slab interface; faces of general partitions): both sides
    add the same two partials, formed by the same kernel (own + received = received + own); a dof shared by three or more ranks of a general
    partition receives its partials in peer order, which differs between the ranks — equal to rounding there, not bitwise.  Dot products weight interface dofs by 1/multiplicity and are all-reduced; pᵀAp is the sum over the ranks of the
    local quadratic forms pᵀA_p p (p is consistent), so it needs no halo.
    `local_spmv(x) -> y` applies the rank's own A_p.  New work: the reference has no distributed solver (README.md:7).

    Execution paths with the same arithmetic:
      * device=<MI355XDevice>, device tensors: the vector work runs in libtbhip (tb_cgd_dot / tb_cgd_update / tb_cgd_direction), α and β are
        formed on the device from all-reduced device scalars, and the host reads (‖r‖², breakdown flag) once per look — every `look`
        iterations — like tb_cg_solve on one device.  With operator=(pattern, nz) the iteration is ordered for overlap: the interface rows of
        A_p·p go straight into the send buffers (tb_spmv_csr_rows), the exchange is posted, the whole local product and its quadratic form
        follow (tb_spmv_csr_dot) while the transfer is in flight, then the received partials are added (tb_scatter_add_indexed);
        without it `local_spmv` is called and its result packed afterwards (no overlap).
      * device=None: plain torch ops on whatever device the tensors live on (CPU in the gloo tests, where the oracle assembles).

    The device's kernels and torch's work (all-reduces, the staging copies under gloo) must share one stream: the constructor puts the device
    on torch's current stream and `solve` refuses to run if that has changed."""

    def __init__(self, local_spmv, local_diag, lower_idx, upper_idx, rank, world_size, dist, neighbours=None, device=None, look=1, operator=None):
        import torch
        self.torch, self.dist = torch, dist
        self.spmv, self.rank, self.world = local_spmv, rank, world_size
        self.dev, self.look, self.operator = device, max(1, int(look)), operator
        self.one_call = True                                          # one rank without shared dofs: a whole iteration through tb_cgd_iteration (False: the four calls)
        if device is not None and local_diag.is_cuda:
            cur = int(torch.cuda.current_stream().cuda_stream)
            if device.stream_handle != cur:
                device.set_stream(cur)
        # slab partitions pass (lower, upper); general partitions pass neighbours = [(peer, index tensor)]
        self.nb = neighbours if neighbours is not None else slab_neighbours(lower_idx, upper_idx, rank, world_size)
        self.halo = HaloExchange(self.nb, dist, local_diag, device if local_diag.is_cuda else None)
        d = local_diag.clone()
        self.halo.exchange_sum(d)                                   # assembled diagonal
        self.dinv = 1.0 / d
        mult = torch.ones_like(d)
        for _, idx in self.nb:
            mult[idx] += 1.0
        self.w = 1.0 / mult                                         # a dof held by k ranks counts 1/k in every dot product
        self.breakdown = None

    def dot(self, a, b):
        s = (self.w * a * b).sum().reshape(1)
        if self.world > 1:
            all_reduce_sum(s, self.dist)
        return float(s.item())

    def apply(self, x):
        if self.spmv is None:                                       # operator=(pattern, nz) given instead of a callable
            from ._lib import check, lib
            pattern, nz = self.operator
            y = self.torch.empty_like(x)
            check(lib().tb_spmv_csr(pattern.h, nz.ptr, x.data_ptr(), 1.0, 0.0, y.data_ptr()))
        else:
            y = self.spmv(x)
        self.halo.exchange_sum(y)
        return y

    def solve(self, b, x, rtol=1e-5, atol=1e-6, maxiter=1000):
        """b: ASSEMBLED right-hand side (consistent), x: initial guess (consistent); returns (x, iterations, ‖r‖)."""
        if self.dev is not None and x.is_cuda:
            return self._solve_device(b, x, rtol, atol, maxiter)
        r = b - self.apply(x)
        z = self.dinv * r
        p = z.clone()
        rz = self.dot(r, z)
        rn = self.dot(r, r) ** 0.5
        tol = atol + rtol * rn
        it = 0
        while rn > tol and it < maxiter:
            Ap = self.apply(p)
            pAp = self.dot(p, Ap)
            if not pAp > 0.0:
                self.breakdown = pAp
                raise ArithmeticError("DistributedCG: pᵀAp = %g ≤ 0 at iteration %d — the operator is not positive definite" % (pAp, it))
            alpha = rz / pAp
            x += alpha * p
            r -= alpha * Ap
            z = self.dinv * r
            rz_new = self.dot(r, z)
            rn = self.dot(r, r) ** 0.5
            p = z + (rz_new / rz) * p
            rz = rz_new
            it += 1
        return x, it, rn

    def device_iteration(self, p, Ap, S):
        """One product of the device path: Ap ← assembled A·p, S[1] ← pᵀAp (summed over the ranks).  S: device scalars rz | pAp | rz_new | rr | flag."""
        import ctypes as C
        from ._lib import check, lib
        L, ptr = lib(), (lambda t: C.c_void_p(t.data_ptr()))
        n = p.numel()
        if self.operator is not None:
            pattern, nz = self.operator
            self.halo.pack_product_rows(pattern, nz, p)              # interface rows first …
            self.halo.start()                                        # … their exchange in flight …
            check(L.tb_spmv_csr_dot(pattern.h, nz.ptr, ptr(p), ptr(Ap), ptr(S[1:2])))   # … behind the whole local product + local pᵀA_p p
            self.halo.finish(Ap, own_from_send=True)                 # interface rows: own (rows kernel) + received (the peer's rows kernel), bitwise symmetric
        else:
            y = self.spmv(p)
            check(L.tb_cgd_dot(self.dev.h, n, None, ptr(p), ptr(y), ptr(S[1:2])))       # local quadratic form, before the halo sum
            self.halo.exchange_sum(y)
            Ap.copy_(y)
        if self.world > 1:
            all_reduce_sum(S[1:2], self.dist)

    def device_step(self, x, r, p, Ap, S):
        """One whole CG iteration of the device path, no host read: product + halo + pᵀAp, update of x and r with the two weighted sums, direction.
        S: SIX device scalars rz | pAp | rz_new | rr | breakdown flag | ‖r‖² of the last finished step; S[1:4] must be zero on entry (they are after
        every step: tb_cgd_rotate).  bench.py times exactly this."""
        import ctypes as C
        from ._lib import check, lib
        L, dev, n = lib(), self.dev, x.numel()
        ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        if self.world == 1 and not self.nb and self.operator is not None and self.one_call:
            pattern, nz = self.operator                              # no shared dofs: the four launches from one library call (tb_cgd_iteration)
            check(L.tb_cgd_iteration(pattern.h, nz.ptr, ptr(self.dinv), ptr(x), ptr(r), ptr(p), ptr(Ap), ptr(S)))
            return
        self.device_iteration(p, Ap, S)                              # S[1:4] are zero: set by the caller before the first step, by tb_cgd_rotate after every step
        w = ptr(self.w) if self.nb else None                         # no shared dofs (one rank, or an isolated part): every weight is 1, the kernel skips the read
        check(L.tb_cgd_update(dev.h, n, w, ptr(self.dinv), ptr(p), ptr(Ap), ptr(x), ptr(r), ptr(S[0:1]), ptr(S[1:2]), ptr(S[2:5])))
        if self.world > 1:
            all_reduce_sum(S[2:4], self.dist)
        check(L.tb_cgd_direction(dev.h, n, ptr(self.dinv), ptr(r), ptr(p), ptr(S[0:1]), ptr(S[2:3])))
        check(L.tb_cgd_rotate(dev.h, ptr(S)))                        # rz ← rz_new, ‖r‖² → S[5], accumulators back to zero: one launch

    def _solve_device(self, b, x, rtol, atol, maxiter):
        import ctypes as C
        from ._lib import check, lib
        torch, dist, dev, n = self.torch, self.dist, self.dev, x.numel()
        if dev.stream_handle != int(torch.cuda.current_stream().cuda_stream):
            raise RuntimeError("DistributedCG: the MI355XDevice is not on torch's current stream (device.set_stream(torch.cuda.current_stream().cuda_stream)); "
                               "its kernels would race with the all-reduces and the exchange")
        L = lib()
        ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        S = torch.zeros(6, dtype=torch.float64, device=x.device)    # rz | pAp | rz_new | rr | breakdown flag | ‖r‖² of the last step  (device-resident scalars)
        Ap = torch.empty_like(x)
        r = b - self.apply(x)
        p = self.dinv * r
        check(L.tb_cgd_dot(dev.h, n, ptr(self.w), ptr(r), ptr(p), ptr(S[0:1])))
        check(L.tb_cgd_dot(dev.h, n, ptr(self.w), ptr(r), ptr(r), ptr(S[3:4])))
        if self.world > 1:
            all_reduce_sum(S[0:4], dist)
        rn = float(S[3].item()) ** 0.5
        S[1:4].zero_()
        tol = atol + rtol * rn
        it = 0
        while rn > tol and it < maxiter:
            for _ in range(min(self.look, maxiter - it)):
                self.device_step(x, r, p, Ap, S)
                it += 1
            h = S[4:6].cpu()                                         # the one host read of the look: the breakdown flag and ‖r‖²
            if float(h[0]) != 0.0:
                self.breakdown = float(h[0])
                raise ArithmeticError("DistributedCG: pᵀAp = %g ≤ 0 — the operator is not positive definite (or the iteration broke down)" % self.breakdown)
            rn = float(h[1]) ** 0.5
        return x, it, rn
