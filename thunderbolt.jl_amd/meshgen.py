"""Idealised cardiac geometries as all-hexahedral grids: the hollow ring and the truncated-ellipsoid left ventricle with an O-grid apex —
generate_ring_mesh / generate_open_ring_mesh / generate_ideal_lv_mesh_hex of the reference (src/mesh/generators.jl:17-250,689-943), with
their facet sets ("Endocardium", "Epicardium", "Base", …; (cell, local facet) pairs, 0-based, Ferrite's hexahedron facet order) and node
sets ("MyocardialAnchor1…4", "Apex", …).  Host-side setup: numpy, independent of the device library."""
import numpy as np

from .api import Grid, Hexahedron


def _wall_cells(node_array, periodic=True):
    """hexahedra of an (nc [× periodic], nr, nl) structured shell; vertex order as Ferrite's Hexahedron"""
    nc, nr, nl = node_array.shape
    nec = nc if periodic else nc - 1
    i = np.arange(nec)
    inext = (i + 1) % nc
    cells = np.empty((nl - 1, nr - 1, nec, 8), dtype=np.int32)
    for k in range(nl - 1):
        for j in range(nr - 1):
            cells[k, j, :, 0] = node_array[i, j, k];         cells[k, j, :, 1] = node_array[inext, j, k]
            cells[k, j, :, 2] = node_array[inext, j + 1, k]; cells[k, j, :, 3] = node_array[i, j + 1, k]
            cells[k, j, :, 4] = node_array[i, j, k + 1];     cells[k, j, :, 5] = node_array[inext, j, k + 1]
            cells[k, j, :, 6] = node_array[inext, j + 1, k + 1]; cells[k, j, :, 7] = node_array[i, j + 1, k + 1]
    return cells.reshape(-1, 8)


def _facets(cell_ids, local):
    c = np.asarray(cell_ids, dtype=np.int32).ravel()
    return np.stack([c, np.full_like(c, local)], axis=1)


def _ring(nc, nr, nl, angles, periodic, inner_radius, outer_radius, longitudinal_lower, longitudinal_upper, apicobasal_tilt):
    radial = np.linspace(inner_radius, outer_radius, nr + 1)
    z = np.linspace(longitudinal_upper, longitudinal_lower, nl + 1)
    n_ang = len(angles)
    # node (i, j, k): angle fastest, then radius, then the axial layer (top layer first)
    rad = radial[None, :, None] - apicobasal_tilt * z[None, None, :] / np.abs(z).max()
    x = rad * np.cos(angles)[:, None, None]
    y = rad * np.sin(angles)[:, None, None]
    zz = np.broadcast_to(z[None, None, :], x.shape)
    xyz = np.stack([x, y, zz], axis=-1).transpose(2, 1, 0, 3).reshape(-1, 3)
    node_array = np.arange(n_ang * (nr + 1) * (nl + 1)).reshape(nl + 1, nr + 1, n_ang).transpose(2, 1, 0)
    cells = _wall_cells(node_array, periodic)
    nec = n_ang if periodic else n_ang - 1
    cell_array = np.arange(nec * nr * nl).reshape(nl, nr, nec).transpose(2, 1, 0)     # [i, j, k]
    g = Grid(Hexahedron, xyz, cells)
    # FacetIndex(cell, 1 / 2 / 4 / 6) of the reference → 0-based local facets 0 / 1 / 3 / 5
    g.facetsets = {"Myocardium": _facets(cell_array[:, :, 0].T, 0), "Endocardium": _facets(cell_array[:, 0, :].T, 1),
                   "Epicardium": _facets(cell_array[:, -1, :].T, 3), "Base": _facets(cell_array[:, :, -1].T, 5)}
    return g, node_array, cell_array


def generate_ring_mesh(num_elements_circumferential, num_elements_radial, num_elements_longitudinal, inner_radius=0.75, outer_radius=1.0,
                       longitudinal_lower=-0.2, longitudinal_upper=0.2, apicobasal_tilt=0.0):
    """generate_ring_mesh(nc, nr, nl; …) (generators.jl:17-129): closed hexahedral ring, linear geometry."""
    nc = num_elements_circumferential
    angles = np.linspace(0.0, 2 * np.pi, nc + 1)[:-1]
    g, node_array, cell_array = _ring(nc, num_elements_radial, num_elements_longitudinal, angles, True, inner_radius, outer_radius,
                                      longitudinal_lower, longitudinal_upper, apicobasal_tilt)
    g.facetsets["RotationalSeam"] = _facets(cell_array[0, :, :].T, 4)
    q1, q3 = int(np.ceil(1 + nc / 4)) - 1, int(np.ceil(1 + 3 * nc / 4)) - 1
    g.nodesets = {"MyocardialAnchor1": np.array([node_array[0, 0, 0]]), "MyocardialAnchor2": np.array([node_array[0, -1, 0]]),
                  "MyocardialAnchor3": np.array([node_array[q1, 0, 0]]), "MyocardialAnchor4": np.array([node_array[q3, 0, 0]])}
    return g


def generate_open_ring_mesh(num_elements_circumferential, num_elements_radial, num_elements_longitudinal, opening_angle, inner_radius=0.75,
                            outer_radius=1.0, longitudinal_lower=-0.2, longitudinal_upper=0.2, apicobasal_tilt=0.0):
    """generate_open_ring_mesh(nc, nr, nl, opening_angle; …) (generators.jl:132-250): the ring opened along the x–z plane."""
    nc = num_elements_circumferential
    angles = np.linspace(opening_angle / 2, 2 * np.pi - opening_angle / 2, nc + 1)
    g, node_array, cell_array = _ring(nc, num_elements_radial, num_elements_longitudinal, angles, False, inner_radius, outer_radius,
                                      longitudinal_lower, longitudinal_upper, apicobasal_tilt)
    g.facetsets["Open1"] = _facets(cell_array[-1, :, :].T, 2)
    g.facetsets["Open2"] = _facets(cell_array[0, :, :].T, 4)
    n_nodes_c = nc + 1
    q1, q3 = int(np.ceil(1 + n_nodes_c / 4)) - 1, int(np.ceil(1 + 3 * n_nodes_c / 4)) - 1
    g.nodesets = {"MyocardialAnchor1": np.array([node_array[0, 0, 0]]), "MyocardialAnchor2": np.array([node_array[0, -1, 0]]),
                  "MyocardialAnchor3": np.array([node_array[q1, 0, 0]]), "MyocardialAnchor4": np.array([node_array[q3, 0, 0]])}
    return g


def _ellipsoid_point(theta, phi, rp, inner_radius, outer_radius, apex_inner, apex_outer, septum_flatness, axis_ratio, eccentricity):
    """point of the idealised wall at longitudinal angle θ (0 = apex), circumferential angle φ, transmural fraction rp (generators.jl:738-767)"""
    theta, phi, rp = np.broadcast_arrays(np.asarray(theta, dtype=float), np.asarray(phi, dtype=float), np.asarray(rp, dtype=float))
    r = inner_radius * (1.0 - rp) + outer_radius * rp
    r1, r2 = r * axis_ratio, r / axis_ratio
    z = np.where(theta < np.pi / 2, (apex_inner * (1.0 - rp) + apex_outer * rp) * np.cos(theta), apex_outer * np.cos(theta))
    x = r1 * np.cos(phi) * np.sin(theta) + np.sin(septum_flatness * theta) * inner_radius
    y = r2 * np.sin(phi) * np.sin(theta) + eccentricity * x * (1.0 - rp)
    x = x - septum_flatness * 0.125 * y ** 2
    return np.stack([x, y, z], axis=-1)


def _ogrid_perimeter(nc, size, roundness):
    phi = 2 * np.pi * np.arange(nc) / nc
    rho = (1 - roundness) / (np.abs(np.cos(phi)) + np.abs(np.sin(phi))) + roundness
    return size * rho[:, None] * np.stack([np.cos(phi), np.sin(phi)], axis=1)


def _ogrid_perimeter_index(k, m):
    """lattice index (a, b), 0-based, of the core node at perimeter position k (counter-clockwise from the corner at angle 0)"""
    k %= 4 * m
    if k <= m:
        return k, 0
    if k <= 2 * m:
        return m, k - m
    if k <= 3 * m:
        return 3 * m - k, m
    return 0, 4 * m - k


def _ogrid_core(nc, size, roundness):
    """(m+1)×(m+1) core lattice by transfinite interpolation of its four perimeter sides (generators.jl:710-735)"""
    m = nc // 4
    P = _ogrid_perimeter(nc, size, roundness)
    at = lambda k: P[k % nc]                                                   # noqa: E731
    lat = np.empty((m + 1, m + 1, 2))
    for b in range(m + 1):
        for a in range(m + 1):
            u, v = a / m, b / m
            south, north, west, east = at(a), at(3 * m - a), at(-b), at(m + b)
            lat[a, b] = ((1 - v) * south + v * north + (1 - u) * west + u * east
                         - ((1 - u) * (1 - v) * at(0) + u * (1 - v) * at(m) + (1 - u) * v * at(3 * m) + u * v * at(2 * m)))
    return lat


def generate_ideal_lv_mesh_hex(num_elements_circumferential, num_elements_radial, num_elements_longitudinal, inner_radius=0.7, outer_radius=1.0,
                               longitudinal_upper=0.2, apex_inner=1.3, apex_outer=1.5, septum_flatness=0.6, axis_ratio=1.2, eccentricity=0.0,
                               core_size=None, core_roundness=0.45, septum_fraction=1.0 / 3.0):
    """generate_ideal_lv_mesh_hex(nc, nr, nl; …) (generators.jl:769-941): truncated ellipsoid, all hexahedra, O-grid cap over the apex."""
    nc, nr, nl = num_elements_circumferential, num_elements_radial, num_elements_longitudinal
    if nc % 4:
        raise ValueError("the O-grid apex needs num_elements_circumferential divisible by 4, got %d" % nc)
    m = nc // 4
    if core_size is None:
        core_size = min(max(1 - 2 * np.pi / nc, 0.35), 0.9)
    i_ant = min(max(int(round(nc * septum_fraction)), 1), nc - 1)              # 0-based index of the first cell past the septum
    shape = dict(inner_radius=inner_radius, outer_radius=outer_radius, apex_inner=apex_inner, apex_outer=apex_outer,
                 septum_flatness=septum_flatness, axis_ratio=axis_ratio, eccentricity=eccentricity)
    phi = np.linspace(0.0, 2 * np.pi, nc + 1)[:-1]
    rp = np.linspace(0.0, 1.0, nr + 1)
    theta = np.linspace(0.0, (1.0 + longitudinal_upper) * np.pi / 2, nl + 2)
    # wall: rings from the one above the apex to the base; node (i, j, k): angle fastest, then transmural, then ring
    wall = _ellipsoid_point(theta[1:, None, None], phi[None, None, :], rp[None, :, None], **shape).reshape(-1, 3)
    n_wall = nc * (nr + 1) * (nl + 1)
    node_array = np.arange(n_wall).reshape(nl + 1, nr + 1, nc).transpose(2, 1, 0)
    # one copy of the core per transmural shell, mapped into the cap between the apex and the first ring
    lat = _ogrid_core(nc, core_size, core_roundness)
    rho, ang = np.linalg.norm(lat, axis=2), np.arctan2(lat[..., 1], lat[..., 0])
    core = np.empty((nr + 1, m + 1, m + 1, 3))
    for j in range(nr + 1):
        core[j] = _ellipsoid_point(rho.T * theta[1], ang.T, rp[j], **shape)     # stored [j][b][a]: a fastest
    core_array = (n_wall + np.arange((m + 1) ** 2 * (nr + 1))).reshape(nr + 1, m + 1, m + 1).transpose(2, 1, 0)     # [a, b, j]
    xyz = np.concatenate([wall, core.reshape(-1, 3)])
    cells = [_wall_cells(node_array, True)]
    ne_wall = nc * nr * nl
    cell_array = np.arange(ne_wall).reshape(nl, nr, nc).transpose(2, 1, 0)
    fs = {"Endocardium": [_facets(cell_array[:, 0, :].T, 1)], "Epicardium": [_facets(cell_array[:, -1, :].T, 3)],
          "Base": [_facets(cell_array[:, :, -1].T, 5)], "SRidgePost": [_facets(cell_array[0, :, :].T, 4)],
          "SRidgeAnt": [_facets(cell_array[i_ant - 1, :, :].T, 2)]}
    # the ring of cells joining the first wall ring to the core: extruded transmurally, so endo / epi are their bottom / top facets
    ncell = ne_wall
    collar = np.empty((nr, nc, 8), dtype=np.int32)
    for j in range(nr):
        for i in range(nc):
            inext = (i + 1) % nc
            a, b = _ogrid_perimeter_index(i, m)
            an, bn = _ogrid_perimeter_index(i + 1, m)
            collar[j, i] = (node_array[i, j, 0], node_array[inext, j, 0], core_array[an, bn, j], core_array[a, b, j],
                            node_array[i, j + 1, 0], node_array[inext, j + 1, 0], core_array[an, bn, j + 1], core_array[a, b, j + 1])
            c = ncell + j * nc + i
            if j == 0: fs["Endocardium"].append(_facets([c], 0))
            if j == nr - 1: fs["Epicardium"].append(_facets([c], 5))
            if i == 0: fs["SRidgePost"].append(_facets([c], 4))
            if i == i_ant - 1: fs["SRidgeAnt"].append(_facets([c], 2))
    cells.append(collar.reshape(-1, 8))
    ncell += nr * nc
    inner = np.empty((nr, m, m, 8), dtype=np.int32)
    for j in range(nr):
        for b in range(m):
            for a in range(m):
                inner[j, b, a] = (core_array[a, b, j], core_array[a + 1, b, j], core_array[a + 1, b + 1, j], core_array[a, b + 1, j],
                                  core_array[a, b, j + 1], core_array[a + 1, b, j + 1], core_array[a + 1, b + 1, j + 1], core_array[a, b + 1, j + 1])
                c = ncell + (j * m + b) * m + a
                if j == 0: fs["Endocardium"].append(_facets([c], 0))
                if j == nr - 1: fs["Epicardium"].append(_facets([c], 5))
    cells.append(inner.reshape(-1, 8))
    g = Grid(Hexahedron, xyz, np.concatenate(cells))
    g.facetsets = {k: np.concatenate(v) for k, v in fs.items()}
    ca, cb = np.unravel_index(np.argmin(rho), rho.shape)
    q1, q3 = int(np.ceil(1 + nc / 4)) - 1, int(np.ceil(1 + 3 * nc / 4)) - 1
    g.nodesets = {"MyocardialAnchor1": np.array([node_array[0, 0, -1]]), "MyocardialAnchor2": np.array([node_array[0, -1, -1]]),
                  "MyocardialAnchor3": np.array([node_array[q1, 0, -1]]), "MyocardialAnchor4": np.array([node_array[q3, 0, -1]]),
                  "Apex": np.array([core_array[ca, cb, -1]]), "ApexInOut": np.array([core_array[ca, cb, 0], core_array[ca, cb, -1]])}
    # parametric coordinates of every node (θ from the apex, φ around, transmural fraction): what the analytic fibre field below needs
    par = np.empty((len(xyz), 3))
    TH, RP, PH = np.meshgrid(theta[1:], rp, phi, indexing="ij")
    par[:n_wall] = np.stack([TH.ravel(), PH.ravel(), RP.ravel()], axis=1)
    cpar = np.empty((nr + 1, m + 1, m + 1, 3))
    for j in range(nr + 1):
        cpar[j, ..., 0], cpar[j, ..., 1], cpar[j, ..., 2] = rho.T * theta[1], ang.T, rp[j]
    par[n_wall:] = cpar.reshape(-1, 3)
    g.parametric = par
    g.shape_parameters = shape
    return g


def ideal_lv_microstructure(g, alpha_endo=np.deg2rad(80.0), alpha_epi=np.deg2rad(-65.0)):
    """Analytic rule-based fibre field on a generate_ideal_lv_mesh_hex grid: at every node the fibre lies in the tangent plane of its
    transmural shell, rotated from the circumferential direction by a helix angle that varies linearly from α_endo to α_epi across the
    wall; sheet normal = shell normal.  (The reference builds the same kind of field from Laplace-solve coordinates — ODB25LT,
    microstructure.jl — which is host setup outside the hot path; this closed form serves examples and tests.)  Returns nodal (f, s, n)
    per cell, the input of OrthotropicMicrostructureModel."""
    th, ph, rp = g.parametric.T
    h = 1e-6
    pt = lambda t, p, r: _ellipsoid_point(t, p, r, **g.shape_parameters)                       # noqa: E731
    e_phi = (pt(th, ph + h, rp) - pt(th, ph - h, rp)) / (2 * h)
    e_th = (pt(th + h, ph, rp) - pt(np.maximum(th - h, 0.0), ph, rp)) / (h + np.minimum(h, th))[:, None]
    sing = np.linalg.norm(e_phi, axis=1) < 1e-8                                                 # the apex: the azimuth does not exist there
    e_phi[sing] = np.cross(np.array([0.0, 0.0, 1.0]), e_th[sing])
    nrm = np.cross(e_phi, e_th)
    bad = np.linalg.norm(nrm, axis=1) < 1e-12
    e_phi[bad], e_th[bad], nrm[bad] = [1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]
    unit = lambda v: v / np.linalg.norm(v, axis=1, keepdims=True)                               # noqa: E731
    nrm, c = unit(nrm), unit(e_phi)
    l = np.cross(nrm, c)                                                                         # longitudinal direction in the shell
    alpha = (alpha_endo * (1.0 - rp) + alpha_epi * rp)[:, None]
    f = np.cos(alpha) * c + np.sin(alpha) * l
    s = np.cross(nrm, f)
    return f[g.conn], s[g.conn], nrm[g.conn]



def uniform_refinement(g):
    """uniform_refinement(grid) for hexahedral grids (src/mesh/tools.jl:257-300): every hexahedron is cut into eight by its edge, face
    and cell centres; new nodes are appended after the old ones — edge centres, face centres, cell centres — shared between
    neighbouring cells.  Sets are not transferred (the reference leaves that as a TODO as well)."""
    if g.cell_kind != Hexahedron:
        raise NotImplementedError("uniform_refinement: hexahedral grids")
    conn = g.conn.astype(np.int64)
    nc = len(conn)
    # Ferrite's reference edges / faces of the hexahedron, 0-based vertex ids
    E = ((0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7))
    F = ((0, 3, 2, 1), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (0, 4, 7, 3), (4, 5, 6, 7))
    ekeys = np.sort(np.stack([conn[:, list(e)] for e in E], axis=1), axis=2).reshape(-1, 2)
    eu, einv = np.unique(ekeys, axis=0, return_inverse=True)
    fkeys = np.sort(np.stack([conn[:, list(f)] for f in F], axis=1), axis=2).reshape(-1, 4)
    fu, finv = np.unique(fkeys, axis=0, return_inverse=True)
    n0 = g.n_nodes
    edge_id = n0 + einv.reshape(nc, 12)
    face_id = n0 + len(eu) + finv.reshape(nc, 6)
    cell_id = n0 + len(eu) + len(fu) + np.arange(nc)
    xyz = np.concatenate([g.xyz, g.xyz[eu].mean(axis=1), g.xyz[fu].mean(axis=1), g.xyz[conn].mean(axis=1)])
    # the 27 points of a cell on the 3×3×3 lattice (index 0, 1, 2 per direction = vertex, centre, vertex)
    lat = np.empty((nc, 3, 3, 3), dtype=np.int64)
    vpos = ((0, 0, 0), (2, 0, 0), (2, 2, 0), (0, 2, 0), (0, 0, 2), (2, 0, 2), (2, 2, 2), (0, 2, 2))
    for v, (i, j, k) in enumerate(vpos):
        lat[:, i, j, k] = conn[:, v]
    for e, (a, b) in enumerate(E):
        i, j, k = [(vpos[a][d] + vpos[b][d]) // 2 for d in range(3)]
        lat[:, i, j, k] = edge_id[:, e]
    for f, vs in enumerate(F):
        i, j, k = [sum(vpos[v][d] for v in vs) // 4 for d in range(3)]
        lat[:, i, j, k] = face_id[:, f]
    lat[:, 1, 1, 1] = cell_id
    new = np.empty((nc, 8, 8), dtype=np.int32)
    sub = 0
    for k in range(2):
        for j in range(2):
            for i in range(2):
                for v, (a, b, c) in enumerate(((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1))):
                    new[:, sub, v] = lat[:, i + a, j + b, k + c]
                sub += 1
    return Grid(Hexahedron, xyz, new.reshape(-1, 8))
