"""Host-side mirror of Thunderbolt.jl's device / operator API for the hot path, bound to libtbhip.so.

The reference host language is Julia; no Julia toolchain exists in the build image, so this mirror is
Python over ctypes (the Julia `ccall` binding a maintainer would add is in INTEGRATION.md and
julia/ThunderboltHIPBackend.jl).  Names, argument meaning and error behaviour follow the reference:

  reference (file:line)                                              here
  ------------------------------------------------------------------------------------------------
  AbstractGPUDevice / CudaDevice   ext/CuThunderboltExt.jl:48-49      MI355XDevice
  PerColor/ElementAssemblyStrategy src/Thunderbolt.jl:22-32           *AssemblyStrategy(device)
  generate_mesh                    src/mesh/generators.jl:942         generate_mesh
  DofHandler/add!/close!           src/discretization/fem.jl:170-196  DofHandler(mesh, ip)
  allocate_matrix                  src/solver/interface.jl:162-168    allocate_matrix(dh)
  Bilinear{Mass,Diffusion}Integrator  core/mass.jl:6-11, diffusion.jl:6-11
  LinearIntegrator                 core/linear.jl:6-14                LinearIntegrator
  setup_operator / update_operator!   src/solver/interface.jl:17-94, euler.jl:172-176
  PointwiseODEFunction             src/modeling/functions.jl:46-65
  ForwardEulerCellSolver / AdaptiveForwardEulerSubstepper / setup_solver_cache / perform_step!
                                   src/solver/time/partitioned_solver.jl:14-52,54-60,101-124,162-175
Julia's `f!` becomes `f_` is avoided: mutating functions simply drop the bang.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import TBError, check, lib


def _ptr(x):
    """device pointer of a DeviceVector, a torch tensor, or a raw int"""
    if x is None:
        return None
    if isinstance(x, DeviceVector):
        return x.ptr
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


# --------------------------------------------------------------------------------------- device
class MI355XDevice:
    """`MI355XDevice{Tv,Ti} <: AbstractGPUDevice` — value type Float64, index type Int32."""
    value_type = np.float64
    index_type = np.int32

    def __init__(self, device_id=0):
        h = C.c_void_p()
        check(lib().tb_device_create(device_id, C.byref(h)))
        self.h = h
        self.device_id = device_id
        self.stream_handle = None     # None: the private non-blocking stream tb_device_create made; else the handle given to set_stream

    def set_stream(self, hip_stream):
        """hip_stream: a hipStream_t handle (int); 0 = the legacy default stream (what torch.cuda.current_stream().cuda_stream is in a
        fresh process: the device's kernels are then ordered with torch's own work and its collectives); None = a private non-blocking stream."""
        self.stream_handle = None if hip_stream is None else int(hip_stream)
        if hip_stream is None:
            check(lib().tb_device_set_stream(self.h, None))
        elif int(hip_stream) == 0:
            check(lib().tb_device_use_null_stream(self.h))
        else:
            check(lib().tb_device_set_stream(self.h, C.c_void_p(int(hip_stream))))

    def synchronize(self):
        check(lib().tb_device_synchronize(self.h))

    def defer_status(self, on=True):
        """tb_device_defer_status: assembly calls stop reading the status block (no stream synchronisation per call); `poll_status` reads it."""
        check(lib().tb_device_defer_status(self.h, 1 if on else 0))

    def poll_status(self):
        """tb_device_poll_status: synchronise, raise the first assembly error (detJ ≤ 0, coupling missing from the pattern) since the last poll, clear."""
        check(lib().tb_device_poll_status(self.h))

    def capture(self, body):
        """HIP graph of the enqueue-only calls `body()` makes on this device (tb_graph_begin / tb_graph_end): returns a DeviceGraph whose `launch(t)` replays
        them with one launch.  Run the sequence once uncaptured first (plans must exist); nothing inside may read back to the host."""
        check(lib().tb_graph_begin(self.h))
        h = C.c_void_p()
        try:
            body()
        except BaseException:
            # the capture is closed either way; a graph that did end is released (nobody will hold its handle) before the exception travels on
            if lib().tb_graph_end(self.h, C.byref(h)) == 0 and h:
                lib().tb_graph_destroy(h)
            raise
        check(lib().tb_graph_end(self.h, C.byref(h)))
        return DeviceGraph(self, h)

    def info(self):
        name = C.create_string_buffer(64)
        ncu, mem = C.c_int(), C.c_size_t()
        check(lib().tb_device_info(self.h, name, 64, C.byref(ncu), C.byref(mem)))
        return {"name": name.value.decode(), "n_cu": ncu.value, "hbm_bytes": mem.value}

    def zeros(self, n, dtype=np.float64):
        v = DeviceVector(self, n, dtype)
        v.fill_zero()
        return v

    def to_device(self, a):
        a = np.ascontiguousarray(a)
        v = DeviceVector(self, a.size, a.dtype)
        v.copy_from_host(a)
        return v

    def event(self):
        return Event(self)

    def close(self):
        if self.h:
            lib().tb_device_destroy(self.h)
            self.h = None


class Event:
    def __init__(self, dev):
        self.dev = dev
        self.h = C.c_void_p()
        check(lib().tb_event_create(dev.h, C.byref(self.h)))

    def record(self):
        check(lib().tb_event_record(self.dev.h, self.h))
        return self

    def elapsed_ms(self, stop):
        ms = C.c_float()
        check(lib().tb_event_elapsed_ms(self.h, stop.h, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if self.h:
                lib().tb_event_destroy(self.h)
        except Exception:
            pass


class DeviceGraph:
    """a captured sequence of device work (MI355XDevice.capture); `launch(t)` sets the time slot the captured forms / ionic models read and replays it"""

    def __init__(self, dev, h):
        self.dev, self.h = dev, h

    def launch(self, t=0.0):
        check(lib().tb_graph_launch(self.h, float(t)))

    @property
    def nodes(self):
        n = C.c_int()
        check(lib().tb_graph_node_count(self.h, C.byref(n)))
        return n.value

    def close(self):
        if self.h:
            lib().tb_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceVector:
    """create_system_vector(::Type{<:DeviceVector}, …) (ext/CuThunderboltExt.jl:126-127)."""

    def __init__(self, dev, n, dtype=np.float64):
        self.dev, self.n, self.dtype = dev, int(n), np.dtype(dtype)
        self.ptr = C.c_void_p()
        self._owner = None
        check(lib().tb_malloc(dev.h, self.n * self.dtype.itemsize, C.byref(self.ptr)))

    @classmethod
    def wrap(cls, dev, tensor):
        """Non-owning view of a host framework's device buffer (e.g. a torch tensor used for RCCL exchange)."""
        v = cls.__new__(cls)
        v.dev, v.n, v.dtype = dev, int(tensor.numel()), np.dtype(np.float64)
        v.ptr = C.c_void_p(tensor.data_ptr())
        v._owner = tensor
        return v

    def view(self, offset, n):
        """Non-owning sub-vector (e.g. the φₘ block `u[heat_dofrange]` of the split problem, fem.jl:399-408)."""
        v = DeviceVector.__new__(DeviceVector)
        v.dev, v.n, v.dtype = self.dev, int(n), self.dtype
        v.ptr = C.c_void_p(self.data_ptr() + int(offset) * self.dtype.itemsize)
        v._owner = self
        return v

    @property
    def nbytes(self):
        return self.n * self.dtype.itemsize

    def data_ptr(self):
        return self.ptr.value or 0

    def fill_zero(self):
        check(lib().tb_memset(self.dev.h, self.ptr, 0, self.nbytes))

    def copy_from_host(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.size == self.n
        check(lib().tb_memcpy_h2d(self.dev.h, self.ptr, a.ctypes.data_as(C.c_void_p), self.nbytes))

    def to_host(self):
        out = np.empty(self.n, dtype=self.dtype)
        check(lib().tb_memcpy_d2h(self.dev.h, out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr and self._owner is None:
            lib().tb_free(self.dev.h, self.ptr)
        self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# --------------------------------------------------------------------------------------- strategies
class _Strategy:
    code = None

    def __init__(self, device):
        self.device = device


class AtomicAssemblyStrategy(_Strategy):
    code = L.TB_STRATEGY_ATOMIC


class PerColorAssemblyStrategy(_Strategy):
    code = L.TB_STRATEGY_PER_COLOR


class ElementAssemblyStrategy(_Strategy):
    code = L.TB_STRATEGY_ELEMENT


class PatchAssemblyStrategy(_Strategy):
    code = L.TB_STRATEGY_PATCH


# --------------------------------------------------------------------------------------- mesh / dofs
Hexahedron, Tetrahedron, Quadrilateral = L.TB_HEX8, L.TB_TET4, L.TB_QUAD4


class LagrangeCollection:
    """LagrangeCollection{order}() (src/ferrite-addons/collections.jl:48-58); `** 3` vectorises."""

    def __init__(self, order=1, ncomp=1):
        self.order, self.ncomp = order, ncomp

    def __pow__(self, n):
        return LagrangeCollection(self.order, n)


class Grid:
    def __init__(self, cell_kind, xyz, conn, dims=None):
        self.cell_kind = cell_kind
        xyz = np.asarray(xyz, dtype=np.float64)
        if xyz.ndim == 2 and xyz.shape[1] == 2:   # 2-D meshes travel with z = 0 (tbhip.h, TB_QUAD4)
            xyz = np.hstack([xyz, np.zeros((len(xyz), 1))])
        self.xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        self.conn = np.ascontiguousarray(conn, dtype=np.int32)
        self.dims = dims

    @property
    def n_cells(self):
        return self.conn.shape[0]

    @property
    def n_nodes(self):
        return self.xyz.shape[0]

    def addcellset(self, name, predicate, all=True):
        """addcellset!(grid, name, x -> Bool; all = true) (Ferrite): cells whose nodes all (any) satisfy the predicate; or pass an
        explicit array of 0-based cell ids instead of a predicate."""
        if not hasattr(self, "cellsets"):
            self.cellsets = {}
        if callable(predicate):
            ok = np.array([bool(predicate(x)) for x in self.xyz])[self.conn]
            cells = np.flatnonzero(ok.all(axis=1) if all else ok.any(axis=1))
        else:
            cells = np.asarray(predicate, dtype=np.int64)
        self.cellsets[name] = cells.astype(np.int32)
        return self.cellsets[name]

    def getcellset(self, name):
        return self.cellsets[name]

    # Ferrite.reference_facets(RefHexahedron), 0-based vertex ids, in Ferrite's facet order
    HEX_FACETS = ((0, 3, 2, 1), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (0, 4, 7, 3), (4, 5, 6, 7))

    def addfacetset(self, name, predicate, all=True, boundary_only=True):
        """addfacetset!(grid, name, x -> Bool; all = true) (Ferrite): the (cell, local facet) pairs — 0-based — whose nodes all (any)
        satisfy the predicate; boundary facets only, like Ferrite's default use on generated grids (boundary_only=False: internal ones too)."""
        if self.cell_kind != Hexahedron:
            raise NotImplementedError("addfacetset: hexahedral grids")
        if getattr(self, "facetsets", None) is None:
            self.facetsets = {}
        ok = np.array([bool(predicate(x)) for x in self.xyz])
        out = []
        count = {}
        if boundary_only:
            for c in range(self.n_cells):
                for f in self.HEX_FACETS:
                    key = tuple(sorted(self.conn[c, list(f)].tolist()))
                    count[key] = count.get(key, 0) + 1
        for c in range(self.n_cells):
            for lf, f in enumerate(self.HEX_FACETS):
                nodes = self.conn[c, list(f)]
                hit = ok[nodes].all() if all else ok[nodes].any()
                if hit and (not boundary_only or count[tuple(sorted(nodes.tolist()))] == 1):
                    out.append((c, lf))
        self.facetsets[name] = np.array(out, dtype=np.int32).reshape(-1, 2)
        return self.facetsets[name]

    def getnodeset(self, name):
        return self.nodesets[name]

    def facetset(self, name):
        """getfacetset(grid, name) of a generated box: left/right (x), front/back (y), bottom/top (z), as (cell, local
        facet) pairs, 0-based (Ferrite generate_grid facetsets; local facets as Ferrite.reference_facets(RefHexahedron))."""
        if getattr(self, "facetsets", None) and name in self.facetsets:      # generated rings / ventricles, loaded meshes
            return self.facetsets[name]
        if self.dims is None or self.cell_kind != Hexahedron:
            raise KeyError("facetset %r: named sets exist on generated meshes only; pass explicit (cell, facet) pairs otherwise" % name)
        nx, ny, nz = self.dims
        c = np.arange(nx * ny * nz).reshape(nz, ny, nx)
        sel, lf = {"left": (c[:, :, 0], 4), "right": (c[:, :, -1], 2), "front": (c[:, 0, :], 1), "back": (c[:, -1, :], 3),
                   "bottom": (c[0, :, :], 0), "top": (c[-1, :, :], 5)}[name]
        cells = sel.ravel()
        return np.stack([cells, np.full_like(cells, lf)], axis=1).astype(np.int32)


def generate_mesh(cell_kind, nel, left=(-1.0, -1.0, -1.0), right=(1.0, 1.0, 1.0), perturb=0.0):
    """generate_mesh(Hexahedron, (nx,ny,nz), left, right) / generate_mesh(Quadrilateral, (nx,ny), left, right) — Ferrite
    generate_grid conventions."""
    if cell_kind == Quadrilateral:
        nx, ny = nel
        xyz = np.empty(((nx + 1) * (ny + 1), 3))
        conn = np.empty((nx * ny, 4), dtype=np.int32)
        le, ri = np.asarray(left, dtype=np.float64)[:2].copy(), np.asarray(right, dtype=np.float64)[:2].copy()
        check(lib().tb_host_generate_grid_quad(nx, ny, le.ctypes.data_as(L.c_dp), ri.ctypes.data_as(L.c_dp),
                                              xyz.ctypes.data_as(L.c_dp), conn.ctypes.data_as(L.c_i32p)))
        return Grid(cell_kind, xyz, conn, dims=(nx, ny))
    if cell_kind != Hexahedron:
        raise NotImplementedError("generate_mesh: Hexahedron boxes and Quadrilateral rectangles are generated; pass your own Grid for others")
    nx, ny, nz = nel
    nn = (nx + 1) * (ny + 1) * (nz + 1)
    xyz = np.empty((nn, 3))
    conn = np.empty((nx * ny * nz, 8), dtype=np.int32)
    le, ri = np.asarray(left, dtype=np.float64), np.asarray(right, dtype=np.float64)
    check(lib().tb_host_generate_grid_hex(nx, ny, nz, le.ctypes.data_as(L.c_dp), ri.ctypes.data_as(L.c_dp),
                                         xyz.ctypes.data_as(L.c_dp), conn.ctypes.data_as(L.c_i32p)))
    if perturb:
        check(lib().tb_host_perturb_nodes(nx, ny, nz, float(perturb), xyz.ctypes.data_as(L.c_dp)))
    return Grid(cell_kind, xyz, conn, dims=(nx, ny, nz))


class DofHandler:
    """DofHandler(mesh); add!(dh, :u, ip); close!(dh) for ONE field on ONE subdomain."""

    def __init__(self, grid, ip=None, cell_dofs=None, ndofs=None):
        ip = ip or LagrangeCollection(1)
        self.grid, self.ip = grid, ip
        if grid.cell_kind == Hexahedron:
            self.field_kind = L.TB_HEX8 if ip.order == 1 else L.TB_HEX27
        else:
            if ip.order != 1:
                raise NotImplementedError("only first-order tetrahedra / quadrilaterals")
            self.field_kind = grid.cell_kind
        nb = {L.TB_HEX8: 8, L.TB_HEX27: 27, L.TB_TET4: 4, L.TB_QUAD4: 4}[self.field_kind]
        self.ndofs_per_cell = nb * ip.ncomp
        if cell_dofs is None:  # close!(dh)
            cd = np.empty((grid.n_cells, self.ndofs_per_cell), dtype=np.int32)
            nd = lib().tb_host_close_dofs(self.field_kind, ip.ncomp, grid.n_cells, grid.n_nodes,
                                          grid.conn.ctypes.data_as(L.c_i32p), cd.ctypes.data_as(L.c_i32p))
            if nd < 0:
                check(int(nd))
            self.cell_dofs, self.ndofs = cd, int(nd)
        else:  # Ferrite's own table handed over by the host
            self.cell_dofs = np.ascontiguousarray(cell_dofs, dtype=np.int32)
            self.ndofs = int(ndofs if ndofs is not None else self.cell_dofs.max() + 1)
        self._mesh = {}

    def device_mesh(self, device):
        key = id(device)
        if key not in self._mesh:
            self._mesh[key] = DeviceMesh(device, self)
        return self._mesh[key]


def ndofs(dh):
    return dh.ndofs


def locality_permutation(grid, dh=None):
    """tb_host_locality_permutation: (cell_perm, node_perm, dof_perm) for an arbitrarily numbered grid — cell_perm[k] = cell to store k-th,
    node_perm[v] / dof_perm[d] = new numbers (dof_perm is None without a DofHandler).  The reference's cell loop does not care how a mesh is
    numbered (src/modeling/core/coordinate_systems.jl:145-171); the device plans do."""
    nc, nn = grid.n_cells, grid.n_nodes
    cp, npm = np.empty(nc, dtype=np.int32), np.empty(nn, dtype=np.int32)
    dp = np.empty(dh.ndofs, dtype=np.int32) if dh is not None else None
    check(lib().tb_host_locality_permutation(grid.cell_kind, nn, grid.xyz.ctypes.data_as(L.c_dp), nc, grid.conn.ctypes.data_as(L.c_i32p),
                                             dh.ndofs_per_cell if dh is not None else 0, dh.cell_dofs.ctypes.data_as(L.c_i32p) if dh is not None else None,
                                             dh.ndofs if dh is not None else 0, 0, cp.ctypes.data_as(L.c_i32p), npm.ctypes.data_as(L.c_i32p),
                                             dp.ctypes.data_as(L.c_i32p) if dp is not None else None))
    return cp, npm, dp


def renumber_grid(grid, cell_perm, node_perm):
    """The same mesh stored in the locality order: cells[cell_perm], nodes moved to their new numbers (what a Julia host does before DofHandler(grid))."""
    inv = np.empty_like(node_perm)
    inv[node_perm] = np.arange(len(node_perm), dtype=node_perm.dtype)
    return Grid(grid.cell_kind, grid.xyz[inv], node_perm[grid.conn[cell_perm]].astype(np.int32), dims=grid.dims)


def renumber_dofs(dh, dof_perm):
    """Ferrite.renumber!(dh, perm): the DofHandler with every dof d renamed dof_perm[d] (cells stay where they are)."""
    return DofHandler(dh.grid, dh.ip, cell_dofs=dof_perm[dh.cell_dofs].astype(np.int32), ndofs=dh.ndofs)


class SparsityPattern:
    def __init__(self, rowptr, colidx):
        self.rowptr, self.colidx = rowptr, colidx

    @property
    def nnz(self):
        return int(self.rowptr[-1])


def allocate_matrix(dh):
    """Sparsity pattern of allocate_matrix(dh), as CSR (create_system_matrix, src/solver/interface.jl:162-168)."""
    cd = dh.cell_dofs
    rowptr = np.empty(dh.ndofs + 1, dtype=np.int64)
    nnz = lib().tb_host_build_pattern(cd.shape[0], cd.shape[1], cd.ctypes.data_as(L.c_i32p), dh.ndofs,
                                      rowptr.ctypes.data_as(L.c_i64p), None)
    if nnz < 0:
        check(int(nnz))
    colidx = np.empty(nnz, dtype=np.int32)
    lib().tb_host_build_pattern(cd.shape[0], cd.shape[1], cd.ctypes.data_as(L.c_i32p), dh.ndofs,
                                rowptr.ctypes.data_as(L.c_i64p), colidx.ctypes.data_as(L.c_i32p))
    return SparsityPattern(rowptr, colidx)


class DeviceMesh:
    """Adapt.adapt_structure(device, dh/grid): device-resident coordinates, connectivity, dof table."""

    def __init__(self, device, dh):
        self.device, self.dh = device, dh
        g = dh.grid
        self.h = C.c_void_p()
        check(lib().tb_mesh_create(device.h, g.cell_kind, g.n_nodes, g.xyz.ctypes.data_as(L.c_dp), g.n_cells,
                                   g.conn.ctypes.data_as(L.c_i32p), dh.field_kind, dh.ip.ncomp,
                                   dh.cell_dofs.ctypes.data_as(L.c_i32p), dh.ndofs, 0, C.byref(self.h)))
        self._patterns = {}

    def pattern(self, sp):
        key = id(sp)
        if key not in self._patterns:
            self._patterns[key] = DevicePattern(self, sp)
        return self._patterns[key]

    def __del__(self):
        try:
            if self.h:
                lib().tb_mesh_destroy(self.h)
        except Exception:
            pass


class DevicePattern:
    def __init__(self, dmesh, sp):
        self.dmesh, self.sp = dmesh, sp
        self.h = C.c_void_p()
        check(lib().tb_pattern_create(dmesh.h, len(sp.rowptr) - 1, sp.rowptr.ctypes.data_as(L.c_i64p),
                                      sp.colidx.ctypes.data_as(L.c_i32p), 0, C.byref(self.h)))
        self.nnz = sp.nnz

    def patch_stats(self):
        """Work statistics of the PATCH plan (tb_pattern_patch_stats): patches, cell instances, instances per cell, largest patch."""
        out = np.zeros(6, dtype=np.int64)
        check(lib().tb_pattern_patch_stats(self.h, out.ctypes.data_as(L.c_i64p)))
        return {"patches": int(out[0]), "instances": int(out[1]), "cells": int(out[2]), "instances_per_cell": float(out[1]) / max(int(out[2]), 1),
                "max_instances": int(out[3]), "max_rows": int(out[4]), "lds_bytes_per_block": int(out[5])}

    def mirror(self, nz):
        """Bind a sliced mirror of the value array `nz` (tb_spmv_mirror): every product of this pattern with `nz` — tb_spmv_csr, the products of the
        Krylov solves — then streams the mirrored copy (coalesced, no LDS staging; the same bits).  Call again after changing the matrix; `nz=None`
        unbinds.  Returns False where the pattern has no mirror (3 × 3 block rows, numberings without shared row signatures)."""
        rc = lib().tb_spmv_mirror(self.h, None if nz is None else nz.ptr)
        if rc == L.TB_ERR_UNSUPPORTED:
            return False
        check(rc)
        # the binding is the address: keep the bound vectors alive so that the allocator cannot hand their addresses to another matrix (two slots per pattern)
        self._mirrored = [] if nz is None else ([v for v in getattr(self, "_mirrored", []) if v is not nz] + [nz])[-2:]
        return nz is not None

    def __del__(self):
        try:
            if self.h:
                lib().tb_pattern_destroy(self.h)
        except Exception:
            pass


# --------------------------------------------------------------------------------------- coefficients
class ConstantCoefficient:
    def __init__(self, val):
        self.val = val


class FieldCoefficient:
    """FieldCoefficient(data[basis, cell]) — scalar nodal data per cell (coefficients.jl:85-99)."""

    def __init__(self, data):
        self.data = np.ascontiguousarray(data, dtype=np.float64)  # stored [cell][basis]


class OrthotropicMicrostructure:
    def __init__(self, f, s, n):
        self.f, self.s, self.n = (np.asarray(v, dtype=np.float64) for v in (f, s, n))


class TransverselyIsotropicMicrostructure:
    def __init__(self, f):
        self.f = np.asarray(f, dtype=np.float64)


class OrthotropicMicrostructureModel:
    """Nodal f, s, n fields per cell, shape (n_cells, nbasis, 3) each (microstructure.jl:145-187)."""

    def __init__(self, f, s, n):
        self.fsn = np.ascontiguousarray(np.stack([f, s, n], axis=2), dtype=np.float64)  # [cell][basis][3][3]


class SpectralTensorCoefficient:
    def __init__(self, eigenvectors, eigenvalues):
        self.eigenvectors, self.eigenvalues = eigenvectors, eigenvalues


class ConductivityToDiffusivityCoefficient:
    """κ/(Cₘ·χ) (coefficients.jl:124-162; built by fem.jl:413-419)."""

    def __init__(self, conductivity, capacitance, chi):
        self.conductivity, self.capacitance, self.chi = conductivity, capacitance, chi


class AnalyticalCoefficient:
    """AnalyticalCoefficient(f, CartesianCoordinateSystem): `f` is a closed-form id
    ("const", "norm_plus_t", "cos_exp") or a Python callable f(x, t) that is tabulated on the host."""
    KINDS = {"const": L.TB_SRC_CONST, "norm_plus_t": L.TB_SRC_NORM_PLUS_T, "cos_exp": L.TB_SRC_COS_EXP}

    def __init__(self, f, value=0.0):
        self.f, self.value = f, value


def _lower_coef(coef, wrap=False, Cm=1.0, chi=1.0):
    c = L.tb_coef()
    c.wrap, c.Cm, c.chi = int(wrap), float(Cm), float(chi)
    keep = None
    if isinstance(coef, ConductivityToDiffusivityCoefficient):
        Cm_, chi_ = coef.capacitance, coef.chi
        Cm_ = Cm_.val if isinstance(Cm_, ConstantCoefficient) else Cm_
        chi_ = chi_.val if isinstance(chi_, ConstantCoefficient) else chi_
        return _lower_coef(coef.conductivity, True, float(Cm_), float(chi_))
    if isinstance(coef, ConstantCoefficient):
        v = np.asarray(coef.val, dtype=np.float64)
        if v.ndim == 0:
            c.kind = L.TB_COEF_CONST_SCALAR
            c.p[0] = float(v)
        else:
            if v.shape == (2, 2):                   # 2-D tensor: upper-left block of the 3×3 the ABI carries
                v = np.block([[v, np.zeros((2, 1))], [np.zeros((1, 2)), np.ones((1, 1))]])
            assert v.shape == (3, 3)
            c.kind = L.TB_COEF_CONST_TENSOR
            for i, x in enumerate(v.ravel()):
                c.p[i] = x
    elif isinstance(coef, FieldCoefficient):
        c.kind = L.TB_COEF_FIELD_SCALAR
        keep = coef.data
    elif isinstance(coef, SpectralTensorCoefficient):
        lam = coef.eigenvalues.val if isinstance(coef.eigenvalues, ConstantCoefficient) else coef.eigenvalues
        lam = np.asarray(lam, dtype=np.float64)
        ev = coef.eigenvectors.val if isinstance(coef.eigenvectors, ConstantCoefficient) else coef.eigenvectors
        if isinstance(ev, OrthotropicMicrostructure):
            c.kind = L.TB_COEF_SPECTRAL_CONST
            for i, x in enumerate(np.concatenate([ev.f, ev.s, ev.n, lam])):
                c.p[i] = x
        elif isinstance(ev, TransverselyIsotropicMicrostructure):
            c.kind = L.TB_COEF_TRANSVERSE_CONST
            for i, x in enumerate(np.concatenate([ev.f, lam])):
                c.p[i] = x
        elif isinstance(ev, OrthotropicMicrostructureModel):
            c.kind = L.TB_COEF_SPECTRAL_FIELD
            for i, x in enumerate(lam):
                c.p[i] = x
            keep = ev.fsn
        else:
            raise TypeError("SpectralTensorCoefficient: unsupported eigenvector coefficient %r" % (ev,))
    else:
        raise TypeError("unsupported coefficient %r" % (coef,))
    if keep is not None:
        c.field = keep.ctypes.data_as(L.c_dp)
        c.field_len = keep.size
    return c, keep


# --------------------------------------------------------------------------------------- integrators / operators
class BilinearMassIntegrator:
    form = L.TB_FORM_MASS

    def __init__(self, rho, qorder=0):
        self.coef, self.qorder = rho, qorder


class BilinearDiffusionIntegrator:
    form = L.TB_FORM_DIFFUSION

    def __init__(self, D, qorder=0):
        self.coef, self.qorder = D, qorder


class LinearIntegrator:
    """LinearIntegrator(source_term, qrc); nonzero_intervals as in AnalyticalTransmembraneStimulationProtocol
    (src/modeling/electrophysiology.jl:260-283), tested by needs_update (src/discretization/operator.jl:17-26)."""
    form = L.TB_FORM_SOURCE

    def __init__(self, source, qorder=0, nonzero_intervals=None):
        self.coef, self.qorder, self.nonzero_intervals = source, qorder, nonzero_intervals


class _Form:
    def __init__(self, dmesh, kind, qorder, c):
        self.h = C.c_void_p()
        check(lib().tb_form_create(dmesh.h, kind, qorder, C.byref(c), C.byref(self.h)))

    def __del__(self):
        try:
            if self.h:
                lib().tb_form_destroy(self.h)
        except Exception:
            pass


class BilinearOperator:
    """Assembled bilinear operator: `.A` is the CSR nzval vector on device, pattern shared per DofHandler."""

    def __init__(self, strategy, integrator, dh, pattern):
        self.strategy, self.integrator, self.dh = strategy, integrator, dh
        self.dmesh = dh.device_mesh(strategy.device)
        self.pattern = self.dmesh.pattern(pattern)
        c, self._keep = _lower_coef(integrator.coef)
        self.form = _Form(self.dmesh, integrator.form, integrator.qorder, c)
        self.A = DeviceVector(strategy.device, pattern.nnz)

    def update(self, t):
        check(lib().tb_assemble_matrix(self.form.h, self.pattern.h, self.strategy.code, float(t), self.A.ptr))

    def mul(self, y, x, alpha=1.0, beta=0.0):
        """mul!(y, op, x, α, β) (src/utils.jl:185-231)."""
        check(lib().tb_spmv_csr(self.pattern.h, self.A.ptr, _ptr(x), alpha, beta, _ptr(y)))


class LinearOperator:
    """Assembled linear operator: `.b` on device (test/gpu/test_operators.jl:24-30)."""

    def __init__(self, strategy, integrator, dh):
        self.strategy, self.integrator, self.dh = strategy, integrator, dh
        self.dmesh = dh.device_mesh(strategy.device)
        src = integrator.coef
        c = L.tb_coef()
        self._callable = None
        if callable(src.f):
            c.kind = L.TB_SRC_TABULATED
            self._callable = src.f
        else:
            c.kind = AnalyticalCoefficient.KINDS[src.f]
            c.p[0] = float(src.value)
        self.form = _Form(self.dmesh, L.TB_FORM_SOURCE, integrator.qorder, c)
        self.b = DeviceVector(strategy.device, dh.ndofs)

    def _tabulate(self, t):
        # host evaluation of the closure at every quadrature point: x_q = Σ Mₐ(ξ_q) Xₐ
        g = self.dh.grid
        if g.cell_kind != Hexahedron:
            raise NotImplementedError("callable sources are tabulated on hexahedra only")
        q = self.integrator.qorder or 2
        pts = {2: [-0.5773502691896258, 0.5773502691896258], 3: [-0.7745966692414834, 0.0, 0.7745966692414834]}[q]
        sx = np.array([-1, 1, 1, -1, -1, 1, 1, -1.0])
        sy = np.array([-1, -1, 1, 1, -1, -1, 1, 1.0])
        sz = np.array([-1, -1, -1, -1, 1, 1, 1, 1.0])
        X = g.xyz[g.conn]  # (nc, 8, 3)
        vals = np.empty((g.n_cells, q ** 3))
        k = 0
        for c_ in pts:
            for b_ in pts:
                for a_ in pts:
                    M = 0.125 * (1 + sx * a_) * (1 + sy * b_) * (1 + sz * c_)
                    xq = np.einsum("a,cad->cd", M, X)
                    vals[:, k] = [self._callable(x, t) for x in xq]
                    k += 1
        check(lib().tb_form_set_table(self.form.h, vals.ctypes.data_as(L.c_dp), vals.size))

    def update(self, t):
        if self._callable is not None:
            self._tabulate(t)
        check(lib().tb_assemble_vector(self.form.h, self.strategy.code, float(t), self.b.ptr))


def setup_operator(strategy, integrator, dh, pattern=None, local_solver=None):
    """setup_operator(strategy, integrator, [solver,] dh) (src/solver/interface.jl:17-94)."""
    if isinstance(integrator, LinearIntegrator):
        return LinearOperator(strategy, integrator, dh)
    if isinstance(integrator, QuasiStaticModel) or (isinstance(integrator, dict) and all(isinstance(v, QuasiStaticModel) for v in integrator.values())):
        return NonlinearOperator(strategy, integrator, dh, pattern or allocate_matrix(dh), local_solver=local_solver)
    if pattern is None:
        pattern = allocate_matrix(dh)
    return BilinearOperator(strategy, integrator, dh, pattern)


def _rebind_mirrors(*ops):
    """Assembling into a value array drops its sliced mirror (tb_spmv_mirror): operators whose array this host mirror had bound are bound again."""
    for op in ops:
        pat, A = getattr(op, "pattern", None), getattr(op, "A", None)
        if pat is not None and A is not None and any(v is A for v in getattr(pat, "_mirrored", [])):
            pat.mirror(A)


def update_operator(op, t):
    """update_operator!(op, t) (src/solver/time/euler.jl:172-176)."""
    op.update(t)
    _rebind_mirrors(op)
    return op


def update_operators(M, K, t):
    """update_operator!(cache.M, t); update_operator!(cache.K, t) of the heat stage (src/solver/time/euler.jl:172-176) in one pass
    over the mesh: M a mass operator, K a diffusion operator of the same DofHandler / pattern / strategy."""
    if (isinstance(M, BilinearOperator) and isinstance(K, BilinearOperator) and M.pattern is K.pattern and M.strategy.code == K.strategy.code
            and M.integrator.form == L.TB_FORM_MASS and K.integrator.form == L.TB_FORM_DIFFUSION):
        check(lib().tb_assemble_matrix_pair(M.form.h, K.form.h, M.pattern.h, M.strategy.code, float(t), M.A.ptr, K.A.ptr))
    else:
        M.update(t)
        K.update(t)
    _rebind_mirrors(M, K)
    return M, K


def needs_update(op, t):
    """needs_update(op::LinearOperator, t) (src/discretization/operator.jl:17-26): closed-interval test."""
    iv = getattr(op.integrator, "nonzero_intervals", None)
    if iv is None:
        return True
    return any(a <= t <= b for a, b in iv)


def heat_system_matrix(device, M, K, dt, A=None):
    """_implicit_euler_heat_solver_update_system_matrix!(A, M, K, Δt): Anz = Mnz − Δt·Knz (euler.jl:110-116)."""
    A = A or DeviceVector(device, M.A.n)
    check(lib().tb_heat_matrix(device.h, M.A.n, M.A.ptr, K.A.ptr, float(dt), A.ptr))
    return A


def add(b, op, device):
    """add!(b, op) (src/solver/time/euler.jl:90)."""
    check(lib().tb_axpy(device.h, op.b.n, 1.0, op.b.ptr, _ptr(b)))


# --------------------------------------------------------------------------------------- reaction
class _IonicModel:
    model_id = None

    def __init__(self, **params):
        ns, npar, phi = C.c_int(), C.c_int(), C.c_int()
        check(lib().tb_cell_model_info(self.model_id, C.byref(ns), C.byref(npar), C.byref(phi)))
        self.nstates, self.phi_index = ns.value, phi.value
        p = np.zeros(npar.value)
        u0 = np.zeros(ns.value)
        check(lib().tb_cell_model_defaults(self.model_id, p.ctypes.data_as(L.c_dp), u0.ctypes.data_as(L.c_dp)))
        for k, v in params.items():
            p[self.param_names.index(k)] = v
        self.params, self._u0 = p, u0

    def default_initial_state(self):
        if self.model_id == L.TB_CELL_PCG2019 and True:
            # recompute from the (possibly modified) parameters, pcg2019.jl:137-152
            p = self.params
            sig = lambda phi, E, k, s: 1.0 / (1.0 + np.exp(s * (phi - E) / k))  # noqa: E731
            n = self.param_names.index
            u0 = np.zeros(7)
            u0[0] = p[n("E_K")]
            u0[1] = sig(u0[0], p[n("E_h")], p[n("k_h")], 1.0)
            u0[2] = sig(u0[0], p[n("E_m")], p[n("k_m")], -1.0)
            u0[3] = sig(u0[0], p[n("E_f")], p[n("k_f")], 1.0)
            u0[4] = sig(u0[0], p[n("E_s")], p[n("k_s")], 1.0)
            u0[5] = sig(u0[0], p[n("E_xs")], p[n("k_xs")], -1.0)
            u0[6] = sig(u0[0], p[n("E_xr")], p[n("k_xr")], -1.0)
            return u0
        return self._u0.copy()


def num_states(model):
    if hasattr(model, "sid"):                  # sarcomere models with internal state
        ns = C.c_int()
        check(lib().tb_sarcomere_model_info(model.sid, C.byref(ns), None))
        return ns.value
    return model.nstates


def transmembranepotential_index(model):
    """1-based like the reference (src/modeling/electrophysiology.jl:149-153)."""
    return model.phi_index + 1


class FHNModel(_IonicModel):
    model_id = L.TB_CELL_FHN
    param_names = ["a", "b", "c", "d", "e", "f"]
    state_symbols = ("φₘ", "s")


class HeterogeneousFHNModel(_IonicModel):
    """HeterogeneousFHNModel of the reference's how-to (docs/src/literate-howto/custom-ep-cell-model.jl:8-56) with e(x) = e0 + g·x: the built-in model
    that reads the point coordinate `x` handed to cell_rhs! (PointwiseODEFunction(npoints, ode, x))."""
    model_id = L.TB_CELL_FHN_HETEROGENEOUS
    param_names = ["a", "b", "c", "d", "e0", "gx", "gy", "gz"]
    state_symbols = ("φₘ", "s")


class AlievPanfilovModel(_IonicModel):
    model_id = L.TB_CELL_ALIEV_PANFILOV
    param_names = ["c_t", "k", "a", "eps0", "mu1", "mu2"]
    state_symbols = ("s", "φₘ")


class PCG2019(_IonicModel):
    model_id = L.TB_CELL_PCG2019
    param_names = ["g_Na", "E_m", "k_m", "tau_m", "E_h", "k_h", "delta_h", "tau_h0", "g_K1", "E_z", "k_z", "g_to", "E_r",
                   "k_r", "E_s", "k_s", "tau_s", "g_CaL", "E_d", "k_d", "E_f", "k_f", "tau_f", "g_Kr", "E_xr", "k_xr",
                   "tau_xr", "E_y", "k_y", "g_Ks", "E_xs", "k_xs", "tau_xs", "E_Na", "E_K", "E_Ca"]
    state_symbols = ("φₘ", "h", "m", "f", "s", "xs", "xr")


class TT06(_IonicModel):
    """ten Tusscher–Panfilov 2006 epicardial cell — extension (the reference has no TT06)."""
    model_id = L.TB_CELL_TT06
    param_names = ["G_Na", "G_K1", "G_Kr", "G_Ks", "G_to", "G_CaL", "G_bNa", "G_bCa", "G_pCa", "G_pK", "k_NaK", "k_NaCa", "K_o", "Ca_o",
                   "Na_o", "V_c", "V_sr", "V_ss", "Buf_c", "K_bufc", "Buf_sr", "K_bufsr", "Buf_ss", "K_bufss", "V_maxup", "K_up", "V_rel",
                   "k1p", "k2p", "k3", "k4", "EC", "max_sr", "min_sr", "V_leak", "V_xfer", "C_m", "p_KNa", "K_mK", "K_mNa", "K_mNai", "K_mCa",
                   "k_sat", "n", "K_pCa", "R", "T", "F"]
    state_symbols = ("φₘ", "Ca_i", "Ca_SR", "Ca_ss", "Na_i", "K_i", "m", "h", "j", "xr1", "xr2", "xs", "r", "s", "d", "f", "f2", "fCass", "R̄")


class ORd2011(_IonicModel):
    """O'Hara–Virág–Varró–Rudy 2011 human ventricular cell (41 states) — extension (SURVEY §8 f4 names it; the reference has the hooks only).
    Parameters: thirteen conductance scalings (1 = published), extracellular concentrations, cell type (0 endo, 1 epi, 2 M)."""
    model_id = L.TB_CELL_ORD11
    param_names = ["s_GNa", "s_GNaL", "s_Gto", "s_PCa", "s_GKr", "s_GKs", "s_GK1", "s_Gncx", "s_Pnak", "s_GKb", "s_PNab", "s_PCab", "s_GpCa",
                   "nao", "cao", "ko", "celltype"]
    state_symbols = ("φₘ", "nai", "nass", "ki", "kss", "cai", "cass", "cansr", "cajsr", "m", "hf", "hs", "j", "hsp", "jp", "mL", "hL", "hLp", "a", "iF",
                     "iS", "ap", "iFp", "iSp", "d", "ff", "fs", "fcaf", "fcas", "jca", "nca", "ffp", "fcafp", "xrf", "xrs", "xs1", "xs2", "xk1",
                     "Jrelnp", "Jrelp", "CaMKt")


class StateBlockedLayout:
    code = L.TB_LAYOUT_SOA


class PointBlockedLayout:
    code = L.TB_LAYOUT_AOS


class PointwiseODEFunction:
    """PointwiseODEFunction(npoints, ode[, x]) (src/modeling/functions.jl:46-65); SoA by default (fem.jl:385-408)."""

    def __init__(self, npoints, ode, x=None, layout=None):
        self.npoints, self.ode, self.x = int(npoints), ode, x
        self.layout = layout or StateBlockedLayout()


def solution_size(f):
    return f.npoints * f.ode.nstates


class ForwardEulerCellSolver:
    def __init__(self, device, batch_size_hint=32):
        self.device, self.batch_size_hint = device, batch_size_hint
        self.substeps, self.reaction_threshold = 1, 0.0


class AdaptiveForwardEulerSubstepper:
    def __init__(self, device, substeps=10, reaction_threshold=0.1, batch_size_hint=32):
        self.device, self.substeps, self.reaction_threshold = device, substeps, reaction_threshold
        self.batch_size_hint = batch_size_hint


class RushLarsenCellSolver:
    """Rush–Larsen stepper for gate-type ionic models (extension; the reference has the reaction_rhs! / state_rhs!
    hooks only): gates exactly for frozen φₘ, the rest forward Euler."""

    def __init__(self, device, batch_size_hint=32):
        self.device, self.batch_size_hint = device, batch_size_hint
        self.substeps, self.reaction_threshold = 1, 0.0
        self.rush_larsen = True


class PointwiseSolverCache:
    """ForwardEulerCellSolverCache / AdaptiveForwardEulerSubstepperCache (partitioned_solver.jl:63-77,178-194):
    `du` is materialised (dumat), `un` is advanced in place."""

    def __init__(self, f, solver, u=None, keep_du=True):
        self.solver = solver
        self.un = u if u is not None else solver.device.zeros(solution_size(f))
        self.du = solver.device.zeros(solution_size(f)) if keep_du else None
        self.substeps, self.reaction_threshold = solver.substeps, solver.reaction_threshold
        # xs: coordinate of every point, Vec{sdim, Float32} (partitioned_solver.jl:63-77; coordinate_systems.jl:43-49), or None
        self.xs, self.sdim = None, 0
        if f.x is not None:
            x = np.ascontiguousarray(np.asarray(f.x, dtype=np.float32).reshape(f.npoints, -1))
            self.sdim = x.shape[1]
            self.xs = DeviceVector(solver.device, x.size, dtype=np.float32)
            self.xs.copy_from_host(x.ravel())


def setup_solver_cache(f, solver, t0=0.0, u=None, keep_du=True):
    return PointwiseSolverCache(f, solver, u=u, keep_du=keep_du)


def pointwise_step_outer_kernel(f, t, dt, cache):
    """_pointwise_step_outer_kernel!(f, t, Δt, cache, ::DeviceVector) → Bool (partitioned_solver.jl:38-52)."""
    m = f.ode
    if getattr(cache.solver, "rush_larsen", False):
        check(lib().tb_reaction_step_rl(cache.solver.device.h, m.model_id, m.params.ctypes.data_as(L.c_dp), len(m.params),
                                        _ptr(cache.un), f.npoints, m.nstates, f.layout.code, float(t), float(dt)))
        return True
    if cache.xs is not None or m.model_id == L.TB_CELL_FHN_HETEROGENEOUS:      # cell_rhs!(du, u, x, t, p) with x = getcoordinate(cache, i)
        check(lib().tb_reaction_step_x(cache.solver.device.h, m.model_id, m.params.ctypes.data_as(L.c_dp), len(m.params),
                                       _ptr(cache.un), _ptr(cache.du), f.npoints, m.nstates, f.layout.code, _ptr(cache.xs), int(cache.sdim),
                                       float(t), float(dt), int(cache.substeps), float(cache.reaction_threshold)))
        return True
    check(lib().tb_reaction_step(cache.solver.device.h, m.model_id, m.params.ctypes.data_as(L.c_dp), len(m.params),
                                 _ptr(cache.un), _ptr(cache.du), f.npoints, m.nstates, f.layout.code, float(t),
                                 float(dt), int(cache.substeps), float(cache.reaction_threshold)))
    return True


def perform_step(f, cache, t, dt):
    """perform_step!(f::PointwiseODEFunction, cache, t, Δt) (partitioned_solver.jl:14-21)."""
    return pointwise_step_outer_kernel(f, t, dt, cache)


def _phi_slice(f, cache):
    m = f.ode
    if f.layout.code == L.TB_LAYOUT_SOA:
        return C.c_void_p(cache.du.data_ptr() + 8 * m.phi_index * f.npoints), 1
    return C.c_void_p(cache.du.data_ptr() + 8 * m.phi_index), m.nstates


def get_reaction_tangent(device, f, cache):
    """R = maximum(@view dumat[:, φₘidx]) — the *signed* maximum, as get_reaction_tangent reads it
    (src/solver/time/rtc.jl:55-73)."""
    base, stride = _phi_slice(f, cache)
    out = C.c_double()
    check(lib().tb_max(device.h, f.npoints, base, stride, C.byref(out)))
    return out.value


def reaction_rate_max(device, f, cache):
    """max |dumat[:, φₘidx]| (diagnostic; the controller itself uses get_reaction_tangent)."""
    base, stride = _phi_slice(f, cache)
    out = C.c_double()
    check(lib().tb_absmax(device.h, f.npoints, base, stride, C.byref(out)))
    return out.value


def perform_step_with_reaction_tangent(f, cache, t, dt):
    """One pointwise step with the reaction tangent reduced inside the kernel (tb_reaction_step_rtc): returns
    (ok, R) with R as get_reaction_tangent would give after the step; `cache.du` may be None."""
    m = f.ode
    out = C.c_double()
    check(lib().tb_reaction_step_rtc(cache.solver.device.h, m.model_id, m.params.ctypes.data_as(L.c_dp), len(m.params),
                                     _ptr(cache.un), _ptr(cache.du), f.npoints, m.nstates, f.layout.code, float(t),
                                     float(dt), int(cache.substeps), float(cache.reaction_threshold), C.byref(out)))
    return True, out.value


# --------------------------------------------------------------------------------------- heat step + operator splitting
def solve_converged(pattern, resnorm):
    """True iff the latest Krylov solve on `pattern` met its stopping test ‖r‖ ≤ atol + rtol·‖r₀‖ (tb_solver_last_tolerance) — also when it
    did so exactly at the iteration limit; the reference fails a step whose linear solve hit MaxIters (newton_raphson.jl)."""
    tol = C.c_double()
    check(lib().tb_solver_last_tolerance(pattern.h, C.byref(tol)))
    return resnorm <= tol.value


def cg_solve(pattern, A, b, x, rtol=1e-5, atol=1e-6, maxiter=1000, jacobi=True, b_is_residual=False):
    """LinearSolve.solve!(linear_solver) with KrylovJL_CG(atol, rtol) (euler.jl:94-100, ep01_spiral-wave.jl:126-128).  b_is_residual: `b`
    already holds b − A·x₀ for the initial guess in `x` (tb_cg_solve_from_residual)."""
    it, res = C.c_int(), C.c_double()
    check((lib().tb_cg_solve_from_residual if b_is_residual else lib().tb_cg_solve)(pattern.h, _ptr(A), _ptr(b), _ptr(x), float(rtol), float(atol), int(maxiter), int(jacobi),
                            C.byref(it), C.byref(res)))
    return it.value, res.value


class L1GSPrecBuilder:
    """L1GSPrecBuilder(partsize) (Thunderbolt.Preconditioners, docs/src/api-reference/solver.md:13-22): ℓ₁ Gauss–Seidel with a symmetric
    sweep over partitions of `partsize` consecutive rows, as a preconditioner of cg_solve / the Newton inner solver."""

    def __init__(self, partsize=64):
        self.partsize = int(partsize)


class ChebyshevPrecBuilder:
    """Chebyshev polynomial preconditioner of degree `degree` over Jacobi scaling (TB_PRECOND_CHEBYSHEV): the smoother of the reference's
    multigrid extension (src/solver/linear/multigrid.jl:28-33) used as a preconditioner of its own — for the elasticity tangents, where
    Jacobi-CG needs thousands of iterations, it cuts the outer iterations (and with them the inner products and host looks) about degree-fold."""

    def __init__(self, degree=16):
        self.degree = int(degree)


def pcg_solve(pattern, A, b, x, rtol=1e-5, atol=1e-6, maxiter=1000, precond="jacobi"):
    """CG with precond = None | "jacobi" | L1GSPrecBuilder(partsize) | ChebyshevPrecBuilder(degree)"""
    if isinstance(precond, ChebyshevPrecBuilder):
        kind, ps = L.TB_PRECOND_CHEBYSHEV, precond.degree
    else:
        kind, ps = (L.TB_PRECOND_NONE, 1) if precond is None else (L.TB_PRECOND_JACOBI, 1) if precond == "jacobi" else (L.TB_PRECOND_L1GS, precond.partsize)
    it, res = C.c_int(), C.c_double()
    check(lib().tb_pcg_solve(pattern.h, _ptr(A), _ptr(b), _ptr(x), float(rtol), float(atol), int(maxiter), kind, ps, C.byref(it), C.byref(res)))
    return it.value, res.value


def l1gs_apply(pattern, A, r, z, partsize=64, sweep="symmetric"):
    """z = M⁻¹ r of the ℓ₁ Gauss–Seidel preconditioner (ForwardSweep / SymmetricSweep)"""
    check(lib().tb_l1gs_apply(pattern.h, _ptr(A), int(partsize), L.TB_SWEEP_SYMMETRIC if sweep == "symmetric" else L.TB_SWEEP_FORWARD, _ptr(r), _ptr(z)))
    return z


def gmres_solve(pattern, A, b, x, rtol=1e-8, atol=1e-14, maxiter=5000, restart=50, jacobi=True):
    """LinearSolve.solve! with KrylovJL_GMRES (the Newton default, newton_raphson.jl:61): restarted GMRES on the device."""
    it, res = C.c_int(), C.c_double()
    check(lib().tb_gmres_solve(pattern.h, _ptr(A), _ptr(b), _ptr(x), float(rtol), float(atol), int(maxiter), int(restart), int(jacobi),
                               C.byref(it), C.byref(res)))
    return it.value, res.value


class BackwardEulerSolver:
    """BackwardEulerSolver(; inner_solver = KrylovJL_CG(atol, rtol)) for an AffineODEFunction (euler.jl:4-15)."""

    def __init__(self, rtol=1e-5, atol=1e-6, maxiter=1000, jacobi=True, mirror=True):
        self.rtol, self.atol, self.maxiter, self.jacobi = rtol, atol, maxiter, jacobi
        self.mirror = mirror   # keep a sliced mirror of the system matrix for the products of the CG (tb_spmv_mirror): a second copy of its values


class BackwardEulerStage:
    """BackwardEulerSolverCache + BackwardEulerAffineODEStage (euler.jl:44-69): M, K, source, A, b, Δt_last;
    setup assembles all operators once at t₀ (euler.jl:143-176)."""

    def __init__(self, solver, strategy, dh, diffusion, source=None, pattern=None, t0=0.0):
        self.solver, self.device = solver, strategy.device
        self.sp = pattern or allocate_matrix(dh)
        self.M = update_operator(setup_operator(strategy, BilinearMassIntegrator(ConstantCoefficient(1.0)), dh, self.sp), t0)
        self.K = update_operator(setup_operator(strategy, BilinearDiffusionIntegrator(diffusion), dh, self.sp), t0)
        self.source = None if source is None else update_operator(setup_operator(strategy, source, dh), t0)
        self.A = DeviceVector(self.device, self.sp.nnz)
        self.b = DeviceVector(self.device, dh.ndofs)
        if solver.mirror:                                                       # K multiplies uₙ₋₁ once per step (the right-hand side below); re-bind after update_operator(K)
            self.K.pattern.mirror(self.K.A)
        self.dt_last = None
        self.last_iters = 0

    def perform_step(self, u, t, dt):
        """perform_backward_euler_step!(f, cache, stage, t, Δt) (euler.jl:71-101): True on success."""
        rebuilt = self.dt_last is None or abs(dt - self.dt_last) > 1e-14 * abs(dt)
        if rebuilt:
            heat_system_matrix(self.device, self.M, self.K, dt, self.A)         # euler.jl:104-116
            self.dt_last = dt
            if self.solver.mirror:                                              # A stays as it is until Δt changes: its products stream the mirror
                self.M.pattern.mirror(self.A)
        # A uₙ = b with b = M uₙ₋₁ (+ f) and the initial guess uₙ₋₁ (euler.jl:85-100): the initial residual b − A uₙ₋₁ is Δt·K·uₙ₋₁ (+ f),
        # so one product with K stands for the two with M and A
        check(lib().tb_spmv_csr(self.K.pattern.h, self.K.A.ptr, _ptr(u), float(dt), 0.0, self.b.ptr))
        if self.source is not None:
            if needs_update(self.source, t + dt):                               # euler.jl:118-120
                update_operator(self.source, t + dt)
            add(self.b, self.source, self.device)
        its, res = cg_solve(self.M.pattern, self.A, self.b, u, self.solver.rtol, self.solver.atol, self.solver.maxiter,
                            (1 if rebuilt else 2) if self.solver.jacobi else 0, b_is_residual=True)     # 2 = TB_JACOBI_REUSE: same A as last step
        self.last_iters = its
        return solve_converged(self.M.pattern, res)


class LieTrotterGodunov:
    """LieTrotterGodunov((BackwardEulerSolver(), ForwardEulerCellSolver())) on the monodomain split
    (OrdinaryDiffEqOperatorSplitting, third party; sequencing per integrator/operatorsplitting-interface.jl):
    heat step on u[heat_dofrange] (the φₘ block of the SoA state, fem.jl:399-408), then the reaction step."""

    def __init__(self, heat_stage, odefun, cell_cache):
        self.heat, self.f, self.cell = heat_stage, odefun, cell_cache
        m = odefun.ode
        if odefun.layout.code != L.TB_LAYOUT_SOA:
            raise NotImplementedError("the split problem stores its state StateBlocked (SoA)")
        self.phi = cell_cache.un.view(m.phi_index * odefun.npoints, odefun.npoints)

    def step(self, t, dt):
        ok = self.heat.perform_step(self.phi, t, dt)
        return ok and perform_step(self.f, self.cell, t, dt)


class ReactionTangentController:
    """ReactionTangentController(ltg, σ_s, σ_c, Δt_bounds) (src/solver/time/rtc.jl:1-125): steps exactly like
    LieTrotterGodunov; after every (always accepted) step the next Δt is σ(R) of the reaction tangent R of the
    pointwise sub-problem.  R comes out of the reaction kernel itself (fused reduction), so `dumat` need not exist."""

    def __init__(self, ltg, sigma_s, sigma_c, dt_bounds):
        self.ltg, self.sigma_s, self.sigma_c = ltg, float(sigma_s), float(sigma_c)
        self.dt_bounds = (float(dt_bounds[0]), float(dt_bounds[1]))
        self.R = 0.0

    def stepsize(self, R):
        """step_accept_controller! (rtc.jl:104-120)."""
        lo, hi = self.dt_bounds
        if np.isinf(self.sigma_s):
            return lo if R > self.sigma_c else hi
        return (1.0 - 1.0 / (1.0 + np.exp((self.sigma_c - R) * self.sigma_s))) * (hi - lo) + lo

    def step(self, t, dt):
        """One LTG step of length dt; returns (ok, next dt)."""
        ok = self.ltg.heat.perform_step(self.ltg.phi, t, dt)
        if not ok:
            return False, dt
        ok, self.R = perform_step_with_reaction_tangent(self.ltg.f, self.ltg.cell, t, dt)
        return ok, self.stepsize(self.R)

    def solve(self, t0, t1, dt):
        """Advance from t0 to t1 (the last step is clipped to land on t1); returns the list of accepted (t, dt)."""
        t, hist = float(t0), []
        while t < t1 - 1e-14 * max(1.0, abs(t1)):
            h = min(dt, t1 - t)
            ok, dt_next = self.step(t, h)
            if not ok:
                raise RuntimeError("RTC: inner step failed at t = %g" % t)
            hist.append((t, h))
            t += h
            dt = dt_next
        self.dt_cache = dt
        return hist


# --------------------------------------------------------------------------------------- mechanics side (off the hot path of SURVEY §8)
# materials, weak boundary conditions, NonlinearOperator, constraints, Newton / load-path solvers, sarcomere models: solid.py.  Imported last (it
# uses the names above) and re-exported, so `setup_operator` finds QuasiStaticModel / NonlinearOperator and the package namespace is what it was.
from .solid import *  # noqa: E402,F401,F403
from .solid import _sync_active_tension  # noqa: E402,F401
