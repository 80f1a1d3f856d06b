"""Harmonic ventricular coordinates by Laplace solves on the device, and a rule-based fibre field built on them — the use the reference makes
of its sequential assembly loop (`_assemble_laplacian`, `_solve_dirichlet_laplace`, `_lumped_gradient`, src/modeling/core/coordinate_systems.jl:
145-233).  The Laplacian is the device diffusion form with D = −I (so that Kₑ[i,j] = +∇Nᵢ·∇Nⱼ dΩ), the Dirichlet data are
eliminated on the device CSR arrays, the solve is the device Jacobi-CG; the lumped gradient recovery is a host (numpy) setup step.
Transmural (endocardium 0 → epicardium 1) and apicobasal (apex 0 → base 1) coordinates; the rotational coordinate is built in its
azimuthal chart (element-wise discontinuous values, `_compute_rotational_from_azimuth!`), not in the ridge-based chart of a bi-ventricular mesh.
`create_microstructure_model` is the reference's generator: local axes from the in-cell gradients at the cell's nodes and the ODB25LT angle rule."""
from dataclasses import dataclass
import numpy as np

from . import api as tb

_SGN = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], dtype=float)


def assemble_laplacian(strategy, dh, sp):
    """_assemble_laplacian(dh, ip): Kₑ[i,j] += (∇v·∇u) dΩ on the device"""
    return tb.update_operator(tb.setup_operator(strategy, tb.BilinearDiffusionIntegrator(tb.ConstantCoefficient(-np.eye(3))), dh, sp), 0.0)


def solve_dirichlet_laplace(K, dh, constraints, rtol=1e-12, maxiter=20000):
    """_solve_dirichlet_laplace(K, dh, solver, constraints): Δu = 0 with u = value on each (dofs, value) pair; the prescribed values are
    written back exactly (coordinate_systems.jl:176-201)."""
    dev = K.A.dev
    dofs = np.concatenate([np.asarray(d, dtype=np.int64) for d, _ in constraints])
    vals = np.concatenate([np.full(len(d), float(v)) for d, v in constraints])
    dofs, first = np.unique(dofs, return_index=True)
    ch = tb.ConstraintHandler(dh, dofs, vals[first])
    u = dev.zeros(dh.ndofs)
    tb.apply(u, ch)
    # K (u_D + δ) = 0 on the free dofs: right-hand side −K u_D, prescribed rows and columns eliminated, δ = 0 there
    A = dev.zeros(K.A.n)
    tb.check(tb.lib().tb_memcpy_d2d(dev.h, A.ptr, K.A.ptr, K.A.nbytes))
    b = dev.zeros(dh.ndofs)
    tb.check(tb.lib().tb_spmv_csr(K.pattern.h, A.ptr, u.ptr, -1.0, 0.0, b.ptr))
    tb.apply_zero(A, b, ch, pattern=K.pattern)
    delta = dev.zeros(dh.ndofs)
    its, res = tb.cg_solve(K.pattern, A, b, delta, rtol=rtol, atol=0.0, maxiter=maxiter)
    tb.check(tb.lib().tb_axpy(dev.h, dh.ndofs, 1.0, delta.ptr, u.ptr))
    tb.apply(u, ch)
    return u.to_host(), its


def lumped_gradient(g, dh, u):
    """_lumped_gradient(dh, ip, u) for first-order hexahedra (coordinate_systems.jl:203-233): gᵢ = Σ ∫ ∇u Nᵢ dΩ / Σ ∫ Nᵢ dΩ, 2×2×2 Gauss points"""
    gp = _SGN / np.sqrt(3.0)
    X = g.xyz[g.conn]                                                   # (cells, 8, 3)
    ue = u[dh.cell_dofs]                                                # (cells, 8)
    grad = np.zeros((dh.ndofs, 3))
    mass = np.zeros(dh.ndofs)
    for xi in gp:
        N = 0.125 * np.prod(1 + _SGN * xi, axis=1)
        dN = np.stack([0.125 * _SGN[:, d] * np.prod(np.delete(1 + _SGN * xi, d, axis=1), axis=1) for d in range(3)], axis=1)      # (8, 3) ∂N/∂ξ
        J = np.einsum("cai,ad->cid", X, dN)
        detJ = np.linalg.det(J)
        dNdx = np.einsum("ad,cdi->cai", dN, np.linalg.inv(J))
        gu = np.einsum("ca,cai->ci", ue, dNdx)                          # ∇u at the point
        w = N[None, :] * detJ[:, None]                                  # (cells, 8)
        np.add.at(mass, dh.cell_dofs, w)
        for d in range(3):
            np.add.at(grad[:, d], dh.cell_dofs, w * gu[:, d:d + 1])
    nz = mass > 0
    grad[nz] /= mass[nz, None]
    return grad, mass


class LVCoordinateSystem:
    """nodal transmural and apicobasal coordinates (dof order of `dh`) of a ventricle with "Endocardium", "Epicardium", "Base" facet sets
    and an "Apex" node set"""

    def __init__(self, grid, dh, transmural, apicobasal, rotational=None):
        self.grid, self.dh, self.u_transmural, self.u_apicobasal = grid, dh, transmural, apicobasal
        self.u_rotational = rotational                                  # (cells, 8): one value per cell-node (DiscontinuousLagrange{1}), or None


def _facet_dofs(g, dh, name):
    fs = g.facetset(name)
    nodes = np.unique(np.concatenate([g.conn[c, list(tb.Grid.HEX_FACETS[lf])] for c, lf in fs]))
    n2d = np.full(g.n_nodes, -1, dtype=np.int64)
    n2d[g.conn.ravel()] = dh.cell_dofs.ravel()
    return n2d[nodes], n2d


def apicobasal_from_laplace(grid, dh, u_laplace, nbins=200):
    """apicobasal_from_laplace(dh, ip, u; nbins) (coordinate_systems.jl:238-300): recalibrate the apicobasal Laplace field to arc length along its
    own trajectories, ab = 1 − F(u)/F(0) with F(u) = ∫ᵤ¹ dū/⟨‖∇u‖⟩(ū), the level-set average ⟨‖∇u‖⟩ taken per bin with the lumped weights of
    `_lumped_gradient`; exactly 0 / 1 on the two Dirichlet sets, monotone in u"""
    u = np.clip(np.asarray(u_laplace, dtype=float), 0.0, 1.0)
    grad, weight = lumped_gradient(grid, dh, u)
    gradnorm = np.linalg.norm(grad, axis=1)
    edges = np.linspace(0.0, 1.0, nbins + 1)
    du = 1.0 / nbins
    b = np.clip(np.searchsorted(edges, u, side="right"), 1, nbins) - 1   # searchsortedlast, 0-based
    num = np.bincount(b, weights=weight * gradnorm, minlength=nbins)
    den = np.bincount(b, weights=weight, minlength=nbins)
    gbar = np.where(den > 0, num / np.where(den > 0, den, 1.0), 0.0)
    for k in range(1, nbins):                                            # empty bins inherit a populated neighbour, both ways
        if gbar[k] == 0:
            gbar[k] = gbar[k - 1]
    for k in range(nbins - 2, -1, -1):
        if gbar[k] == 0:
            gbar[k] = gbar[k + 1]
    F = np.zeros(nbins + 1)
    for k in range(nbins - 1, -1, -1):
        F[k] = F[k + 1] + du / max(gbar[k], np.finfo(float).eps)
    if not F[0] > 0:
        return np.zeros_like(u)
    lam = (u - edges[b]) / du
    return np.clip(1.0 - (F[b] + lam * (F[b + 1] - F[b])) / F[0], 0.0, 1.0)


def _apical_cap_facets(grid, apex, longitudinal, height, surface_names):
    """_apical_cap_facets (coordinate_systems.jl:752-783): facets of the named surfaces whose nodes all lie within `height` of the apex along the
    long axis — a Dirichlet set of positive capacity for the apical end of the apicobasal problem"""
    below = (grid.xyz - apex) @ longitudinal <= height
    cap = []
    for name in surface_names:
        if name not in grid.facetsets:
            continue
        for c, lf in grid.facetset(name):
            if below[grid.conn[c, list(tb.Grid.HEX_FACETS[lf])]].all():
                cap.append((c, lf))
    return cap


def compute_lv_coordinate_system(grid, device, strategy=None, apex_nodeset="Apex", apex_facetset=None, apical_cap_fraction=0.0,
                                 apicobasal_bins=200, recalibrate=True):
    """compute_lv_coordinate_system(mesh; apex_nodeset, apex_facetset, apical_cap_fraction, apicobasal_bins) (coordinate_systems.jl:886-952):
    transmural and apicobasal coordinates by Dirichlet–Laplace solves on the device; the apical end is pinned on a named facet set, on the
    apical cap cut off by a plane at `apical_cap_fraction` of the chamber length, or (the default, as in the reference) on the apex node set;
    the apicobasal Laplace field is then recalibrated to arc length (apicobasal_from_laplace).  The long axis runs from the apex node to the
    centroid of the base nodes (the reference's compute_lv_axes fits it to the basal plane).  `u_apicobasal_laplace` keeps the raw field."""
    dh = tb.DofHandler(grid)
    sp = tb.allocate_matrix(dh)
    K = assemble_laplacian(strategy or tb.PerColorAssemblyStrategy(device), dh, sp)
    endo, n2d = _facet_dofs(grid, dh, "Endocardium")
    epi, _ = _facet_dofs(grid, dh, "Epicardium")
    base, _ = _facet_dofs(grid, dh, "Base")
    transmural, _ = solve_dirichlet_laplace(K, dh, [(endo, 0.0), (epi, 1.0)])
    transmural = np.clip(transmural, 0.0, 1.0)
    apex_nodes = np.asarray(grid.getnodeset(apex_nodeset))
    apical = n2d[apex_nodes]
    if apex_facetset is not None:
        if apex_facetset not in grid.facetsets:
            raise ValueError('No facetset "%s" on this mesh.' % apex_facetset)
        apical, _ = _facet_dofs(grid, dh, apex_facetset)
    elif apical_cap_fraction > 0:
        apex = grid.xyz[apex_nodes].mean(axis=0)
        d2n = np.empty(dh.ndofs, dtype=np.int64)
        d2n[n2d] = np.arange(grid.n_nodes)
        base_center = grid.xyz[d2n[base]].mean(axis=0)
        l = (base_center - apex) / np.linalg.norm(base_center - apex)
        cap = _apical_cap_facets(grid, apex, l, apical_cap_fraction * abs((base_center - apex) @ l), ("Endocardium", "Epicardium"))
        if cap:
            apical = n2d[np.unique(np.concatenate([grid.conn[c, list(tb.Grid.HEX_FACETS[lf])] for c, lf in cap]))]
    laplace, _ = solve_dirichlet_laplace(K, dh, [(apical, 0.0), (base, 1.0)])
    apicobasal = apicobasal_from_laplace(grid, dh, laplace, apicobasal_bins) if recalibrate else laplace
    cs = LVCoordinateSystem(grid, dh, transmural, apicobasal)
    cs.u_apicobasal_laplace = laplace
    return cs


def _orthogonal_to(axis):
    e = np.eye(3)[np.argmin(np.abs(axis))]
    e = e - (e @ axis) * axis
    return e / np.linalg.norm(e)


def wrap_rotational(r):
    return np.mod(r, 1.0)


def rotational_from_azimuth(grid, origin, axis, zero_direction):
    """_compute_rotational_from_azimuth! + _unwrap_cell_angles! (coordinate_systems.jl:515-590): r = θ/2π around `axis` per cell-node, each
    cell unwrapped so that the branch cut falls outside of it and placed on the turn of its mean"""
    l = np.asarray(axis, dtype=float) / np.linalg.norm(axis)
    e1 = np.asarray(zero_direction, dtype=float)
    e1 = e1 - (e1 @ l) * l
    if np.linalg.norm(e1) < np.sqrt(np.finfo(float).eps):
        raise ValueError("The azimuth reference direction must not be collinear with the long axis.")
    e1 /= np.linalg.norm(e1)
    e2 = np.cross(l, e1)
    X = grid.xyz[grid.conn]                                             # (cells, 8, 3)
    d = X - np.asarray(origin, dtype=float)
    d = d - (d @ l)[..., None] * l
    tol = 1e-6 * np.linalg.norm(X - X[:, :1], axis=2).max(axis=1)
    defined = np.linalg.norm(d, axis=2) > tol[:, None]
    ang = np.where(defined, np.mod(np.arctan2(d @ e2, d @ e1), 2 * np.pi) / (2 * np.pi), 0.0)
    first = np.argmax(defined, axis=1)
    ref = np.take_along_axis(ang, first[:, None], axis=1)
    delta = ang - ref
    ang = ref + (delta - np.round(delta))                               # rem(·, 1, RoundNearest)
    cnt = defined.sum(axis=1)
    mean = np.where(cnt > 0, (ang * defined).sum(axis=1) / np.maximum(cnt, 1), 0.0)
    turn = np.floor(mean)
    out = np.where(defined, ang - turn[:, None], (mean - turn)[:, None])
    out[cnt == 0] = 0.0
    return out


def _sheet_direction(grid, name, origin, axis):
    """mean radial direction of the facets of `name` (coordinate_systems.jl:341-352), or None"""
    if name not in grid.facetsets:
        return None
    nodes = np.unique(np.concatenate([grid.conn[c, list(tb.Grid.HEX_FACETS[lf])] for c, lf in grid.facetset(name)]))
    d = grid.xyz[nodes] - origin
    d = d - (d @ axis)[:, None] * axis
    nrm = np.linalg.norm(d, axis=1)
    keep = nrm >= np.finfo(float).eps
    direction = (d[keep] / nrm[keep, None]).sum(axis=0)
    if np.linalg.norm(direction) < 0.5:
        return None
    return direction / np.linalg.norm(direction)


def compute_midmyocardial_section_coordinate_system(grid, device, up=(0.0, 0.0, 1.0), apicobasal_lower=0.4, apicobasal_upper=0.6,
                                                    epicardium_name="Epicardium", endocardium_name="Endocardium",
                                                    rotational_zero_direction=None, strategy=None):
    """compute_midmyocardial_section_coordinate_system(mesh; ridge_anterior = nothing, ridge_posterior = nothing)
    (coordinate_systems.jl:971-1027): transmural coordinate by a Dirichlet–Laplace solve on the device (clamped to [0,1] as the reference
    does after its Krylov solve), apicobasal = height along `up` rescaled into [lower, upper], rotational = azimuth around `up` through the
    node centroid, zero on the "RotationalSeam" sheet if the mesh has one."""
    dh = tb.DofHandler(grid)
    sp = tb.allocate_matrix(dh)
    K = assemble_laplacian(strategy or tb.PerColorAssemblyStrategy(device), dh, sp)
    endo, n2d = _facet_dofs(grid, dh, endocardium_name)
    epi, _ = _facet_dofs(grid, dh, epicardium_name)
    transmural, _ = solve_dirichlet_laplace(K, dh, [(endo, 0.0), (epi, 1.0)])
    transmural = np.clip(transmural, 0.0, 1.0)
    up = np.asarray(up, dtype=float)
    apicobasal = np.zeros(dh.ndofs)
    apicobasal[n2d[np.arange(grid.n_nodes)]] = grid.xyz @ up
    apicobasal -= apicobasal.min()
    apicobasal /= apicobasal.max()
    apicobasal = apicobasal * (apicobasal_upper - apicobasal_lower) + apicobasal_lower
    l = up / np.linalg.norm(up)
    origin = grid.xyz.sum(axis=0) / grid.n_nodes
    zero = None if rotational_zero_direction is None else np.asarray(rotational_zero_direction, dtype=float)
    if zero is None:
        zero = _sheet_direction(grid, "RotationalSeam", origin, l)
    if zero is None:
        zero = _orthogonal_to(l)
    return LVCoordinateSystem(grid, dh, transmural, apicobasal, rotational_from_azimuth(grid, origin, l, zero))


def _shape_gradients(X, xi):
    """∂Nₐ/∂x of the trilinear hexahedron at reference point ξ for every cell: (cells, 8, 3)"""
    f = 1 + _SGN * np.asarray(xi, dtype=float)
    dN = np.stack([0.125 * _SGN[:, d] * np.prod(np.delete(f, d, axis=1), axis=1) for d in range(3)], axis=1)
    J = np.einsum("cai,ad->cid", X, dN)
    return np.einsum("ad,cdi->cai", dN, np.linalg.inv(J)), 0.125 * np.prod(f, axis=1)


def evaluate_coordinate_axes(cs, xi):
    """evaluate_coordinate_axes + _local_axes (coefficients.jl:427-446, coordinate_systems.jl:30-36) at reference point ξ of every cell:
    transmural = ∇φ_t/|∇φ_t|, apicobasal = ∇φ_ab made orthogonal to it, rotational = transmural × apicobasal"""
    dNdx, _ = _shape_gradients(cs.grid.xyz[cs.grid.conn], xi)
    gt = np.einsum("ca,cai->ci", cs.u_transmural[cs.dh.cell_dofs], dNdx)
    ga = np.einsum("ca,cai->ci", cs.u_apicobasal[cs.dh.cell_dofs], dNdx)
    t = gt / np.linalg.norm(gt, axis=1, keepdims=True)
    a = ga / np.linalg.norm(ga, axis=1, keepdims=True)
    a = a - np.einsum("ci,ci->c", a, t)[:, None] * t
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    r = np.cross(t, a)
    return t, a, r / np.linalg.norm(r, axis=1, keepdims=True)


def evaluate_coordinate(cs, xi):
    """evaluate_coefficient(::LVCoordinateSystem…) at ξ of every cell: (transmural, apicobasal[, rotational])"""
    N = 0.125 * np.prod(1 + _SGN * np.asarray(xi, dtype=float), axis=1)
    out = [cs.u_transmural[cs.dh.cell_dofs] @ N, cs.u_apicobasal[cs.dh.cell_dofs] @ N]
    if cs.u_rotational is not None:
        out.append(wrap_rotational(cs.u_rotational @ N))
    return tuple(out)


@dataclass
class ODB25LTMicrostructureParameters:
    """ODB25LTMicrostructureParameters (microstructure.jl:192-199): helix α, transversal β and sheetlet γ angles, linear across the wall"""
    αendo: float = np.deg2rad(60.0)
    αepi: float = np.deg2rad(-60.0)
    βendo: float = 0.0
    βepi: float = 0.0
    γendo: float = 0.0
    γepi: float = 0.0


def _rotate_around(v, a, theta):
    """Rodrigues rotation of v around the unit axis a (utils.jl:98-102)"""
    c, s = np.cos(theta)[..., None], np.sin(theta)[..., None]
    return v * c + np.cross(a, v) * s + a * np.einsum("...i,...i->...", a, v)[..., None] * (1 - c)


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def compute_local_microstructure(p, transmural, axes):
    """compute_local_microstructure(::ODB25LTMicrostructureParameters, x, axes) (microstructure.jl:208-245)"""
    t_dir, a_dir, c_dir = axes
    x = np.asarray(transmural, dtype=float)
    alpha = (1 - x) * p.αendo + x * p.αepi
    beta = (1 - x) * p.βendo + x * p.βepi
    gamma = (1 - x) * p.γendo + x * p.γepi
    f = _unit(_rotate_around(c_dir, t_dir, alpha))
    f = _unit(_rotate_around(f, a_dir, -beta))
    s = _unit(_rotate_around(c_dir, t_dir, alpha + np.pi / 2.0))
    s = _unit(s - np.einsum("...i,...i->...", s, f)[..., None] * f)
    s = _unit(_rotate_around(s, f, -gamma))
    return f, s, _unit(np.cross(f, s))


def create_microstructure_model(cs, parameters=None):
    """create_microstructure_model(coordinate_system, ip_collection, parameters) (microstructure.jl:279-333): the rule evaluated at the nodal
    quadrature points of every cell (the cell's own gradients there, not a recovered nodal gradient); returns (f, s, n), each (cells, 8, 3)
    in the [basis, cell] layout a FieldCoefficient / OrthotropicMicrostructureModel takes."""
    p = parameters or ODB25LTMicrostructureParameters()
    nc = cs.grid.n_cells
    f, s, n = (np.zeros((nc, 8, 3)) for _ in range(3))
    for a in range(8):
        axes = evaluate_coordinate_axes(cs, _SGN[a])
        f[:, a], s[:, a], n[:, a] = compute_local_microstructure(p, cs.u_transmural[cs.dh.cell_dofs[:, a]], axes)
    return f, s, n


def create_lumped_microstructure_model(cs, alpha_endo=np.deg2rad(80.0), alpha_epi=np.deg2rad(-65.0)):
    """Helix-angle rule on *recovered nodal* gradients (`_lumped_gradient`), a continuous variant of create_microstructure_model that stays
    defined at the apex, where the in-cell gradients of the two coordinates become parallel: the transmural direction
    is ∇φ_t, the longitudinal one ∇φ_ab made orthogonal to it, the circumferential one their cross product; f = cos α c + sin α l,
    n = transmural, s = n × f.  Returns nodal (f, s, n) per cell for OrthotropicMicrostructureModel."""
    g, dh = cs.grid, cs.dh
    gt, _ = lumped_gradient(g, dh, cs.u_transmural)
    ga, _ = lumped_gradient(g, dh, cs.u_apicobasal)

    def unit(v):
        return v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-300)
    bad_t = np.linalg.norm(gt, axis=1) < 1e-12
    gt[bad_t] = [1.0, 0.0, 0.0]
    t = unit(gt)
    l = ga - np.einsum("ni,ni->n", ga, t)[:, None] * t
    # where the apicobasal gradient vanishes or is parallel to the transmural direction (the apex): complete the frame with the
    # coordinate axis least aligned with t
    deg = np.linalg.norm(l, axis=1) < 1e-8 * max(1.0, np.linalg.norm(ga, axis=1).max())
    axis = np.eye(3)[np.argmin(np.abs(t), axis=1)]
    l[deg] = axis[deg] - np.einsum("ni,ni->n", axis[deg], t[deg])[:, None] * t[deg]
    l = unit(l)
    c = np.cross(t, l)                                                  # circumferential, counter-clockwise seen from the base side of the apex
    alpha = (alpha_endo * (1.0 - cs.u_transmural) + alpha_epi * cs.u_transmural)[:, None]
    f = np.cos(alpha) * c + np.sin(alpha) * l
    s = np.cross(t, f)
    cd = dh.cell_dofs
    return f[cd], s[cd], t[cd]
