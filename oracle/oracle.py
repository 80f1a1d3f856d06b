"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py — never by the shipped package (thunderbolt.jl_amd).
The arithmetic lives in tb_oracle.c; this file only marshals numpy arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

LINE2, QUAD4, HEX8, TET4, HEX27 = 1, 2, 3, 4, 5
COEF_CONST_SCALAR, COEF_CONST_TENSOR, COEF_FIELD_SCALAR = 0, 1, 2
COEF_SPECTRAL_CONST, COEF_SPECTRAL_FIELD, COEF_TRANSVERSE_CONST = 3, 4, 5
SRC_CONST, SRC_NORM_PLUS_T, SRC_COS_EXP, SRC_TABULATED = 0, 1, 2, 3
CELL_FHN, CELL_ALIEV_PANFILOV, CELL_PCG2019, CELL_TT06, CELL_FHN_HETEROGENEOUS, CELL_ORD11 = 0, 1, 2, 3, 4, 5
LAYOUT_SOA, LAYOUT_AOS = 0, 1

_dp = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


class _Mesh(C.Structure):
    _fields_ = [("kind", C.c_int), ("qorder", C.c_int), ("n_cells", C.c_int64), ("n_nodes", C.c_int64),
                ("xyz", _dp), ("conn", _i32p), ("cell_dofs", _i32p)]


class _Coef(C.Structure):
    _fields_ = [("kind", C.c_int), ("p", _dp), ("field", _dp), ("Cm", C.c_double), ("chi", C.c_double),
                ("wrap", C.c_int)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "tb_oracle.c")
    if force or not os.path.exists(so) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_close_dofs.restype = C.c_int64
        _LIB.orc_build_pattern.restype = C.c_int64
    return _LIB


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i32(a):
    return None if a is None else a.ctypes.data_as(_i32p)


def _i64(a):
    return None if a is None else a.ctypes.data_as(_i64p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---------------------------------------------------------------- FE substrate
def elem_info(kind):
    rd, nb = C.c_int(), C.c_int()
    assert lib().orc_elem_info(kind, C.byref(rd), C.byref(nb)) == 0
    return rd.value, nb.value


def quadrature(kind, order):
    rd, _ = elem_info(kind)
    xi = np.zeros(64 * 3)
    w = np.zeros(64)
    nq = lib().orc_quadrature(kind, order, _d(xi), _d(w))
    assert nq > 0
    return xi[:nq * rd].reshape(nq, rd).copy(), w[:nq].copy()


def shape(kind, xi):
    rd, nb = elem_info(kind)
    xi = _f64(xi)
    N = np.zeros(nb)
    dN = np.zeros((nb, rd))
    assert lib().orc_shape(kind, _d(xi), _d(N), _d(dN)) == 0
    return N, dN


def mapping(x, dM):
    x = _f64(x)
    dM = _f64(dM)
    ngeo, dim = x.shape
    J = np.zeros((dim, dim))
    Jinv = np.zeros((dim, dim))
    det = C.c_double()
    rc = lib().orc_mapping(dim, ngeo, _d(x), _d(dM), _d(J), C.byref(det), _d(Jinv))
    return rc, J, det.value, Jinv


def eval_field(Nq, data_cell):
    Nq = _f64(Nq)
    data = _f64(data_cell).reshape(len(Nq), -1)
    out = np.zeros(data.shape[1])
    lib().orc_eval_field(len(Nq), data.shape[1], _d(Nq), _d(data), _d(out))
    return out


def eval_cartesian(Nq, coords):
    return eval_field(Nq, coords)


def eval_spectral(vecs, lam):
    vecs = _f64(np.atleast_2d(vecs))
    lam = _f64(lam)
    nvec, dim = vecs.shape
    D = np.zeros((dim, dim))
    lib().orc_eval_spectral(dim, nvec, _d(vecs), _d(lam), _d(D))
    return D


def orthogonalize(f, s, n=None):
    f, s = _f64(f).copy(), _f64(s).copy()
    n = None if n is None else _f64(n).copy()
    lib().orc_orthogonalize(len(f), _d(f), _d(s), _d(n))
    return (f, s) if n is None else (f, s, n)


def conductivity_to_diffusivity(kappa, Cm, chi):
    k = _f64(kappa)
    D = np.zeros_like(k)
    lib().orc_conductivity_to_diffusivity(k.size, _d(k), C.c_double(Cm), C.c_double(chi), _d(D))
    return D


def homogeneous_data_index(timings, t):
    tm = _f64(timings)
    return lib().orc_eval_homogeneous_data_index(len(tm), _d(tm), C.c_double(t))


# ---------------------------------------------------------------- generators
def generate_grid_hex(nx, ny, nz, left=(-1.0, -1.0, -1.0), right=(1.0, 1.0, 1.0)):
    nn = (nx + 1) * (ny + 1) * (nz + 1)
    xyz = np.zeros((nn, 3))
    conn = np.zeros((nx * ny * nz, 8), dtype=np.int32)
    lib().orc_generate_grid_hex(nx, ny, nz, _d(_f64(left)), _d(_f64(right)), _d(xyz), _i32(conn))
    return xyz, conn


def close_dofs(kind, ncomp, conn, n_nodes):
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    _, nb = elem_info(kind)
    cd = np.zeros((conn.shape[0], nb * ncomp), dtype=np.int32)
    nd = lib().orc_close_dofs(kind, ncomp, C.c_int64(conn.shape[0]), C.c_int64(n_nodes), _i32(conn), _i32(cd))
    return cd, int(nd)


def build_pattern(cell_dofs, ndofs):
    cd = np.ascontiguousarray(cell_dofs, dtype=np.int32)
    rowptr = np.zeros(ndofs + 1, dtype=np.int64)
    nnz = lib().orc_build_pattern(C.c_int64(cd.shape[0]), cd.shape[1], _i32(cd), C.c_int64(ndofs), _i64(rowptr), None)
    colidx = np.zeros(nnz, dtype=np.int32)
    lib().orc_build_pattern(C.c_int64(cd.shape[0]), cd.shape[1], _i32(cd), C.c_int64(ndofs), _i64(rowptr), _i32(colidx))
    return rowptr, colidx


def color_cells(cell_dofs, ndofs):
    cd = np.ascontiguousarray(cell_dofs, dtype=np.int32)
    color = np.zeros(cd.shape[0], dtype=np.int32)
    nc = lib().orc_color_cells(C.c_int64(cd.shape[0]), cd.shape[1], _i32(cd), C.c_int64(ndofs), _i32(color))
    assert nc > 0
    return color, nc


# ---------------------------------------------------------------- element kernels / drivers
class Mesh:
    def __init__(self, kind, qorder, xyz, conn, cell_dofs):
        self.kind, self.qorder = kind, qorder
        self.xyz = _f64(xyz)
        self.conn = np.ascontiguousarray(conn, dtype=np.int32)
        self.cell_dofs = np.ascontiguousarray(cell_dofs, dtype=np.int32)
        self.ndofs = int(self.cell_dofs.max()) + 1
        self.c = _Mesh(kind, qorder, self.conn.shape[0], self.xyz.shape[0], _d(self.xyz), _i32(self.conn),
                       _i32(self.cell_dofs))


class Coef:
    def __init__(self, kind, p=None, field=None, Cm=1.0, chi=1.0, wrap=False):
        self.p = _f64(p if p is not None else [0.0])
        self.field = None if field is None else _f64(field)
        self.c = _Coef(kind, _d(self.p), _d(self.field), Cm, chi, int(wrap))


def element_matrix(mesh, form, coef, cell, t=0.0):
    nb = mesh.cell_dofs.shape[1]
    Ke = np.zeros((nb, nb))
    f = lib().orc_element_mass if form == 0 else lib().orc_element_diffusion
    rc = f(C.byref(mesh.c), C.c_int64(cell), C.byref(coef.c), C.c_double(t), _d(Ke))
    assert rc == 0, rc
    return Ke


def element_source(mesh, cell, src_kind, p=None, table=None, t=0.0):
    nb = mesh.cell_dofs.shape[1]
    be = np.zeros(nb)
    p = _f64(p if p is not None else [0.0])
    table = None if table is None else _f64(table)
    rc = lib().orc_element_source(C.byref(mesh.c), C.c_int64(cell), src_kind, _d(p), _d(table), C.c_double(t), _d(be))
    assert rc == 0, rc
    return be


def assemble_matrix(mesh, form, coef, rowptr, colidx, t=0.0, nthreads=1, color=None, ncolors=0):
    nz = np.zeros(int(rowptr[-1]))
    rc = lib().orc_assemble_matrix(C.byref(mesh.c), form, C.byref(coef.c), C.c_double(t), _i64(rowptr),
                                   _i32(colidx), _d(nz), nthreads, _i32(color), ncolors)
    assert rc == 0, rc
    return nz


class AssemblyPlan:
    """CPU-baseline form of the per-colour loop (bench.py's cpu_baseline): scatter positions looked up once, per-colour cell lists, first-touch zero
    fill — orc_assembly_plan_*.  `assemble(form, coef, nz)` fills (and returns) nz; pass the same `nz` array every time to keep its pages where the
    first call put them."""

    def __init__(self, mesh, rowptr, colidx, color, ncolors, nthreads):
        self.mesh, self.nthreads = mesh, int(nthreads)
        self.rowptr, self.colidx, self.color = rowptr, colidx, np.ascontiguousarray(color, dtype=np.int32)
        self.h = C.c_void_p()
        L = lib()
        L.orc_assembly_plan_create.restype = C.c_int
        L.orc_assembly_plan_destroy.restype = None
        L.orc_assembly_plan_destroy.argtypes = [C.c_void_p]
        rc = L.orc_assembly_plan_create(C.byref(mesh.c), _i64(rowptr), _i32(colidx), _i32(self.color), int(ncolors), self.nthreads, C.byref(self.h))
        if rc:
            raise RuntimeError("orc_assembly_plan_create -> %d" % rc)
        self.nnz = int(rowptr[-1])

    def new_values(self):
        return np.empty(self.nnz)                       # untouched pages: the first assembly's zero fill places them

    def assemble(self, form, coef, nz):
        rc = lib().orc_assemble_matrix_planned(self.h, C.byref(self.mesh.c), int(form), C.byref(coef.c), _d(nz), self.nthreads)
        if rc:
            raise RuntimeError("orc_assemble_matrix_planned -> %d" % rc)
        return nz

    def assemble_source(self, src_kind, b, p=None, t=0.0):
        par = _f64(p if p is not None else [0.0])
        rc = lib().orc_assemble_source_planned(self.h, C.byref(self.mesh.c), int(src_kind), _d(par), None, C.c_double(t), _d(b), self.nthreads)
        if rc:
            raise RuntimeError("orc_assemble_source_planned -> %d" % rc)
        return b

    def __del__(self):
        try:
            if self.h:
                lib().orc_assembly_plan_destroy(self.h)
        except Exception:
            pass


def assemble_source(mesh, src_kind, p=None, table=None, t=0.0, nthreads=1):
    b = np.zeros(mesh.ndofs)
    p = _f64(p if p is not None else [0.0])
    table = None if table is None else _f64(table)
    rc = lib().orc_assemble_source(C.byref(mesh.c), src_kind, _d(p), _d(table), C.c_double(t), _d(b), nthreads)
    assert rc == 0, rc
    return b


# ---------------------------------------------------------------- reaction
def cell_nstates(model):
    return lib().orc_cell_nstates(model)


def cell_default_params(model):
    p = np.zeros(lib().orc_cell_nparams(model))
    lib().orc_cell_default_params(model, _d(p))
    return p


def cell_default_state(model, p=None):
    p = cell_default_params(model) if p is None else _f64(p)
    u0 = np.zeros(cell_nstates(model))
    lib().orc_cell_default_state(model, _d(p), _d(u0))
    return u0


def cell_rhs(model, p, u, t=0.0):
    p, u = _f64(p), _f64(u)
    du = np.zeros_like(u)
    lib().orc_cell_rhs(model, _d(p), _d(u), C.c_double(t), _d(du))
    return du


def reaction_step(model, p, u, npoints, layout=LAYOUT_SOA, t=0.0, dt=1.0, substeps=1, threshold=0.1, nthreads=1,
                  want_du=True):
    """In place on `u` (float64, contiguous). Returns du (or None)."""
    assert u.dtype == np.float64 and u.flags.c_contiguous
    p = _f64(p)
    du = np.zeros_like(u) if want_du else None
    rc = lib().orc_reaction_step(model, _d(p), _d(u), _d(du), C.c_int64(npoints), layout, C.c_double(t),
                                 C.c_double(dt), substeps, C.c_double(threshold), nthreads)
    assert rc == 0
    return du


def reaction_step_x(model, p, u, npoints, xs, layout=LAYOUT_SOA, t=0.0, dt=1.0, substeps=1, threshold=0.1, want_du=True):
    """reaction_step with the point coordinates xs (npoints × sdim, Float32) handed to cell_rhs! — in place on `u`; returns du (or None)."""
    assert u.dtype == np.float64 and u.flags.c_contiguous
    p = _f64(p)
    xs = np.ascontiguousarray(np.asarray(xs, dtype=np.float32).reshape(npoints, -1))
    du = np.zeros_like(u) if want_du else None
    f = lib().orc_reaction_step_x
    f.restype = C.c_int
    rc = f(C.c_int(model), _d(p), _d(u), _d(du), C.c_int64(npoints), C.c_int(layout), xs.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(xs.shape[1]),
           C.c_double(t), C.c_double(dt), C.c_int(substeps), C.c_double(threshold), C.c_int(1))
    assert rc == 0, rc
    return du


def reaction_step_rl(model, p, u, npoints, layout=LAYOUT_SOA, t=0.0, dt=1.0, nthreads=1):
    """Rush–Larsen step, in place on `u` (TT06 and PCG2019)."""
    assert u.dtype == np.float64 and u.flags.c_contiguous
    p = _f64(p)
    rc = lib().orc_reaction_step_rl(model, _d(p), _d(u), C.c_int64(npoints), layout, C.c_double(t), C.c_double(dt), nthreads)
    assert rc == 0, rc


# ---------------------------------------------------------------- heat-step algebra
def heat_matrix(Mnz, Knz, dt):
    A = np.zeros_like(Mnz)
    lib().orc_heat_matrix(C.c_int64(Mnz.size), _d(_f64(Mnz)), _d(_f64(Knz)), C.c_double(dt), _d(A))
    return A


def spmv_csr(rowptr, colidx, nz, x, alpha=1.0, beta=0.0, y=None, nthreads=1):
    n = len(rowptr) - 1
    y = np.zeros(n) if y is None else y
    lib().orc_spmv_csr(C.c_int64(n), _i64(rowptr), _i32(colidx), _d(_f64(nz)), _d(_f64(x)), C.c_double(alpha),
                       C.c_double(beta), _d(y), nthreads)
    return y


# ---------------------------------------------------------------- quasi-static hyperelasticity
HO_DEFAULTS = np.array([0.059, 8.023, 18.472, 16.026, 2.581, 11.120, 0.216, 11.436, 1.0])  # energies.jl:136-146, :80-82


def ho_energy(F, p=HO_DEFAULTS, fsn=np.eye(3)):
    F, p, fsn = _f64(F), _f64(p), _f64(fsn)
    P = np.zeros((3, 3))
    A = np.zeros((9, 9))
    lib().orc_ho_energy.restype = C.c_double
    psi = lib().orc_ho_energy(_d(p), _d(fsn), _d(F), _d(P), _d(A))
    return psi, P, A


def set_microstructure_field(field):
    """field: (n_cells, 8, 3, 3) nodal f,s,n or None (constant frame)."""
    global _FSN_KEEP
    _FSN_KEEP = None if field is None else _f64(field)
    lib().orc_set_microstructure_field(_d(_FSN_KEEP))


_FSN_KEEP = None


def element_hyperelastic(mesh, cell, ue, p=HO_DEFAULTS, fsn=np.eye(3), want_K=True, want_r=True):
    nd = mesh.cell_dofs.shape[1]
    Ke = np.zeros((nd, nd)) if want_K else None
    re = np.zeros(nd) if want_r else None
    rc = lib().orc_element_hyperelastic(C.byref(mesh.c), C.c_int64(cell), _d(_f64(p)), _d(_f64(fsn)), _d(_f64(ue)), _d(Ke), _d(re))
    assert rc == 0, rc
    return Ke, re


EN_HO, EN_NULL, EN_BIO_NEOHOOKEAN, EN_TI_NEOHOOKEAN, EN_LIN_YIN_PASSIVE, EN_LIN_YIN_ACTIVE, EN_HSY, EN_LINEAR_SPRING, EN_GUCCIONE = range(9)
PEN_SIMPLE, PEN_NULL, PEN_HN1, PEN_HN2, PEN_HN3 = range(5)


def energy(energy, penalty, p, up, F, fsn=np.eye(3)):
    """(Ψ, P, 𝔸) of any reference energy by hyper-dual AD."""
    pp = np.zeros(9); pp[:len(p)] = p
    uu = np.zeros(3); uu[:len(up)] = up
    P, A = np.zeros(9), np.zeros(81)
    lib().orc_energy.restype = C.c_double
    psi = lib().orc_energy(int(energy), int(penalty), _d(pp), _d(uu), _d(_f64(fsn)), _d(_f64(F)), _d(P), _d(A))
    return psi, P.reshape(3, 3), A.reshape(9, 9)


def set_material(energy=0, penalty=0, p=None, up=None):
    """material of element_hyperelastic / assemble_hyperelastic (global; set_material() restores HO + SimpleCompressionPenalty)."""
    pp = np.zeros(9); uu = np.zeros(3)
    if p is not None: pp[:len(p)] = p
    if up is not None: uu[:len(up)] = up
    lib().orc_set_material.restype = None
    lib().orc_set_material(int(energy), int(penalty), _d(pp), _d(uu))


HILL_NONE, HILL_GENERALIZED, HILL_EXTENDED = 0, 1, 2
ACT_SIMPLE_ACTIVE_SPRING = 100
ADG_GMK, ADG_GMK_INCOMPRESSIBLE, ADG_RLRSQ = 0, 1, 2
SARC_PSL1995, SARC_CONSTANT_STRETCH = 0, 1


def set_hill(framework=0, act_energy=0, act_penalty=0, act_p=None, adg=0, kappa=0.0, sarc=0, sarc_p=(0.0, 0.0)):
    """Generalized / ExtendedHillModel over the material of set_material (global; set_hill() switches it off).  The calcium
    state is given with set_active_tension(Ca[, nodal field])."""
    ap = np.zeros(12)
    if act_p is not None: ap[:len(act_p)] = act_p
    sp = np.zeros(2); sp[:len(sarc_p)] = sarc_p
    lib().orc_set_hill.restype = None
    lib().orc_set_hill.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p]
    lib().orc_set_hill(int(framework), int(act_energy), int(act_penalty), ap.ctypes.data, int(adg), float(kappa), int(sarc), sp.ctypes.data)


def set_point_activation(a):
    """activation seen by direct energy() calls: Ta of the active stress or the calcium state of a Hill framework"""
    lib().orc_set_point_activation.restype = None
    lib().orc_set_point_activation(C.c_double(float(a)))


BC_ROBIN, BC_NORMAL_SPRING, BC_PRESSURE, BC_BENDING_SPRING, BC_PRESSURE_FIELD = 0, 1, 2, 3, 4
_PF_KEEP = [None]


def set_facet_pressure_field(field=None):
    f = None if field is None else np.ascontiguousarray(field, dtype=np.float64)
    _PF_KEEP[0] = f
    lib().orc_set_facet_pressure_field.restype = None
    lib().orc_set_facet_pressure_field(_d(f))


_ACT_KEEP = [None]


def set_active_tension(tension, field=None):
    """ActiveStressModel + SimpleActiveStress: Ta = tension·(nodal calcium field per cell, or 1).  Global state of the
    oracle (like set_microstructure_field); reset with set_active_tension(0.0)."""
    f = None if field is None else np.ascontiguousarray(field, dtype=np.float64)
    _ACT_KEEP[0] = f
    lib().orc_set_active_tension.restype = None
    lib().orc_set_active_tension(C.c_double(tension), _d(f))



def element_facet(mesh, cell, local_facet, kind, param, fq, ue, want_K=True, want_r=True):
    """assemble_facet! of one (cell, local facet 0…5) (weak_boundary_conditions.jl): returns (Ke, re) contributions."""
    nd = mesh.cell_dofs.shape[1]
    Ke = np.zeros((nd, nd)) if want_K else None
    re = np.zeros(nd) if want_r else None
    rc = lib().orc_element_facet(C.byref(mesh.c), C.c_int64(cell), int(local_facet), int(kind), C.c_double(param), int(fq), _d(_f64(ue)),
                                 _d(Ke), _d(re))
    assert rc == 0, rc
    return Ke, re


def assemble_facets(mesh, kind, param, fq, facets, u, rowptr=None, colidx=None, nz=None, r=None):
    """Adds the facet integrals of `facets` ((cell, local facet) pairs) to nz / r (created zeroed when None is passed
    together with a pattern / as r=True)."""
    facets = np.ascontiguousarray(facets, dtype=np.int32).reshape(-1, 2)
    if nz is None and rowptr is not None:
        nz = np.zeros(int(rowptr[-1]))
    if r is None or r is True:
        r = np.zeros(mesh.ndofs)
    rc = lib().orc_assemble_facets(C.byref(mesh.c), int(kind), C.c_double(param), int(fq), _i32(facets), C.c_int64(len(facets)), _d(_f64(u)),
                                   _i64(rowptr), _i32(colidx), _d(nz), _d(r))
    assert rc == 0, rc
    return nz, r


def assemble_hyperelastic(mesh, u, rowptr=None, colidx=None, p=HO_DEFAULTS, fsn=np.eye(3), want_K=True, want_r=True,
                          nthreads=1, color=None, ncolors=0):
    nz = np.zeros(int(rowptr[-1])) if want_K else None
    r = np.zeros(mesh.ndofs) if want_r else None
    rc = lib().orc_assemble_hyperelastic(C.byref(mesh.c), _d(_f64(p)), _d(_f64(fsn)), _d(_f64(u)), _i64(rowptr), _i32(colidx),
                                         _d(nz), _d(r), nthreads, _i32(color), ncolors)
    assert rc == 0, rc
    return nz, r


# ------------------------------------------------------------------------------------------- RDQ20-MF sarcomere model
# RDQ20MFModel defaults in struct field order (contraction.jl:337-369)
RDQ20MF_DEFAULTS = np.array([1.25, 1.65, 0.18, 2.2, 2.0, 0.381, -0.571, 10.0, 12.0, 0.1, 0.013, 0.13431, 25.184, 0.032653, 0.000778, 22.894e3, 1.0e-6])


def rdq20mf_rhs(u, lam, dlam, ca, p=RDQ20MF_DEFAULTS):
    du = np.zeros(20)
    lib().orc_rdq20mf_rhs.restype = None
    lib().orc_rdq20mf_rhs(_d(_f64(p)), _d(_f64(u)), C.c_double(lam), C.c_double(dlam), C.c_double(ca), _d(du))
    return du


def rdq20mf_tension(u, lam, p=RDQ20MF_DEFAULTS):
    lib().orc_rdq20mf_tension.restype = C.c_double
    return lib().orc_rdq20mf_tension(_d(_f64(p)), _d(_f64(u)), C.c_double(lam))


def rdq20mf_stiffness(u, lam, p=RDQ20MF_DEFAULTS):
    lib().orc_rdq20mf_stiffness.restype = C.c_double
    return lib().orc_rdq20mf_stiffness(_d(_f64(p)), _d(_f64(u)), C.c_double(lam))


def rdq20mf_trajectory(u0, dt, lam, dlam, ca, sample, p=RDQ20MF_DEFAULTS):
    """forward Euler over len(lam) steps; returns (final state, states after the steps flagged in `sample`)"""
    u = _f64(u0).copy()
    lam, dlam, ca = _f64(lam), _f64(dlam), _f64(ca)
    sample = np.ascontiguousarray(sample, dtype=np.uint8)
    out = np.zeros((int(sample.sum()), 20))
    lib().orc_rdq20mf_trajectory.restype = None
    lib().orc_rdq20mf_trajectory(_d(_f64(p)), _d(u), C.c_int64(len(lam)), C.c_double(dt), _d(lam), _d(dlam), _d(ca),
                                 sample.ctypes.data_as(C.c_void_p), _d(out))
    return u, out


def rdq20mf_local_solve(Qguess, Qknown, lam, ca, dt, tol=1e-4, max_iters=10, dlam=0.0, p=RDQ20MF_DEFAULTS, rate=False):
    """backward-Euler local problem + corrector(s) → (status, Q, dQ/dλ, iterations, last residual norm[, dQ/dλ̇ with rate=True])"""
    Q = _f64(Qguess).copy()
    dQdl, dQdv = np.zeros(20), np.zeros(20)
    it, rn = C.c_int(), C.c_double()
    lib().orc_rdq20mf_local_solve_rate.restype = C.c_int
    code = lib().orc_rdq20mf_local_solve_rate(_d(_f64(p)), _d(Q), _d(_f64(Qknown)), C.c_double(lam), C.c_double(dlam), C.c_double(ca), C.c_double(dt),
                                              C.c_double(tol), C.c_int(max_iters), _d(dQdl), _d(dQdv) if rate else None, C.byref(it), C.byref(rn))
    return (code, Q, dQdl, it.value, rn.value, dQdv) if rate else (code, Q, dQdl, it.value, rn.value)


_COND_KEEP = [None]


def set_condensation(Q=None, Qknown=None, dt=1.0, tmax=1.0, tol=1e-4, max_iters=10, p=RDQ20MF_DEFAULTS, status=None, u_prev=None):
    """Condensed RDQ20-MF internal variable in element_hyperelastic / assemble_hyperelastic (global; set_condensation() switches it
    off).  Q (20 × n_points, point = cell·n_qp + q, C-contiguous float64) is the initial guess and is overwritten with the solution;
    calcium comes from set_active_tension(scale[, nodal field])."""
    lib().orc_set_condensation.restype = None
    lib().orc_set_condensation.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_void_p]
    if Q is None:
        _COND_KEEP[0] = None
        lib().orc_set_condensation(None, 0.0, None, None, 0, 1.0, 0.0, 1, None)
        return
    assert Q.flags.c_contiguous and Q.dtype == np.float64 and Q.shape[0] == 20
    Qk = np.ascontiguousarray(Qknown, dtype=np.float64)
    pp = _f64(p)
    up = None if u_prev is None else np.ascontiguousarray(u_prev, dtype=np.float64)
    _COND_KEEP[0] = (Q, Qk, pp, status, up)
    lib().orc_set_condensation(pp.ctypes.data, float(tmax), Q.ctypes.data, Qk.ctypes.data, Q.shape[1], float(dt), float(tol), int(max_iters),
                               None if status is None else status.ctypes.data)
    lib().orc_set_condensation_rate.restype = None
    lib().orc_set_condensation_rate.argtypes = [C.c_void_p]
    lib().orc_set_condensation_rate(None if up is None else up.ctypes.data)   # rate-coupled form: Ḟ = (∇u − ∇u_prev)/Δt


def set_prestress(F0inv=None):
    """PrestressedMechanicalModel with a constant field in element_hyperelastic / assemble_hyperelastic (global; set_prestress() removes it)."""
    lib().orc_set_prestress.restype = None
    lib().orc_set_prestress.argtypes = [C.c_void_p]
    if F0inv is None:
        lib().orc_set_prestress(None)
    else:
        G = _f64(F0inv).ravel()
        lib().orc_set_prestress(G.ctypes.data)
