"""Harmonic ventricular coordinates by Laplace solves on the device, and a rule-based fibre field built on them — the use the reference makes
of its sequential assembly loop (`_assemble_laplacian`, `_solve_dirichlet_laplace`, `_lumped_gradient`, src/modeling/core/coordinate_systems.jl:
145-233).  The Laplacian is the device diffusion form with D = −I (so that Kₑ[i,j] = +∇Nᵢ·∇Nⱼ dΩ), the Dirichlet data are
eliminated on the device CSR arrays, the solve is the device Jacobi-CG; the lumped gradient recovery is a host (numpy) setup step.
Transmural (endocardium 0 → epicardium 1) and apicobasal (apex 0 → base 1) coordinates; the rotational coordinate of the reference
(discontinuous interpolation across the ridge sheets) is not built."""
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

    def __init__(self, grid, dh, transmural, apicobasal):
        self.grid, self.dh, self.u_transmural, self.u_apicobasal = grid, dh, transmural, apicobasal


def _facet_dofs(g, dh, name):
    fs = g.facetset(name)
    nodes = np.unique(np.concatenate([g.conn[c, list(tb.Grid.HEX_FACETS[lf])] for c, lf in fs]))
    n2d = np.full(g.n_nodes, -1, dtype=np.int64)
    n2d[g.conn.ravel()] = dh.cell_dofs.ravel()
    return n2d[nodes], n2d


def compute_lv_coordinate_system(grid, device, strategy=None):
    """compute_lv_coordinate_system(mesh) (coordinate_systems.jl): two harmonic coordinates by Dirichlet–Laplace solves on the device"""
    dh = tb.DofHandler(grid)
    sp = tb.allocate_matrix(dh)
    K = assemble_laplacian(strategy or tb.PerColorAssemblyStrategy(device), dh, sp)
    endo, n2d = _facet_dofs(grid, dh, "Endocardium")
    epi, _ = _facet_dofs(grid, dh, "Epicardium")
    base, _ = _facet_dofs(grid, dh, "Base")
    apex = n2d[np.asarray(grid.getnodeset("Apex"))]
    transmural, _ = solve_dirichlet_laplace(K, dh, [(endo, 0.0), (epi, 1.0)])
    apicobasal, _ = solve_dirichlet_laplace(K, dh, [(apex, 0.0), (base, 1.0)])
    return LVCoordinateSystem(grid, dh, transmural, apicobasal)


def create_microstructure_model(cs, alpha_endo=np.deg2rad(80.0), alpha_epi=np.deg2rad(-65.0)):
    """Rule-based fibres from the harmonic coordinates, in the spirit of the reference's create_microstructure_model(cs, …, ODB25LT…)
    (microstructure.jl; simplified: helix angle linear in the transmural coordinate, no transverse or sheet angle): the transmural direction
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
