"""Host-side mesh readers for the formats Thunderbolt.jl loads (src/mesh/tools.jl:429-665): openCARP (.elem/.pts),
MFEM mesh v1.0 and the voom2 legacy format (.ele/.nodes/.fsn).  Pure parsing — no device work; the result feeds
`Grid` / `DofHandler` / the C ABI.  Node ids are returned 0-based, vertex order is the reference's (i.e. Ferrite's)
for every cell type, including the two permutations the reference applies (MFEM triangle and pyramid).

The hot path integrates hexahedra and tetrahedra; other cell types are parsed (so a file's cell numbering and its
domain sets stay intact) and can be inspected, but `MixedGrid.grid(kind)` only hands out Hexahedron / Tetrahedron."""
import numpy as np

from . import _lib as L

LINE, TRIANGLE, QUADRILATERAL, TETRAHEDRON, HEXAHEDRON, WEDGE, PYRAMID = "Line", "Triangle", "Quadrilateral", "Tetrahedron", "Hexahedron", "Wedge", "Pyramid"
NVERTS = {LINE: 2, TRIANGLE: 3, QUADRILATERAL: 4, TETRAHEDRON: 4, HEXAHEDRON: 8, WEDGE: 6, PYRAMID: 5}
_ABI_KIND = {HEXAHEDRON: L.TB_HEX8, TETRAHEDRON: L.TB_TET4}


class MixedGrid:
    """Grid(elements, nodes; cellsets) of the loaders: cells keep the file order; `cellsets` maps the attribute
    (as a string, like the reference) to the ordered list of 0-based cell indices carrying it."""

    def __init__(self, cell_types, cells, nodes, cellsets=None):
        self.cell_types = list(cell_types)          # per cell: one of the type names above (None = skipped / unknown)
        self.cells = [None if c is None else tuple(int(v) for v in c) for c in cells]
        self.nodes = np.ascontiguousarray(nodes, dtype=np.float64)
        self.cellsets = cellsets or {}

    def __len__(self):
        return len(self.cells)

    def cells_of(self, type_name):
        return [i for i, t in enumerate(self.cell_types) if t == type_name]

    def grid(self, type_name=None):
        """The cells of one integrable type as an api.Grid (3-D coordinates; 2-D files are padded with z = 0)."""
        from .api import Grid
        if type_name is None:
            kinds = {t for t in self.cell_types if t in _ABI_KIND}
            if len(kinds) != 1:
                raise ValueError("grid(): say which of %s to extract" % sorted(kinds))
            type_name = kinds.pop()
        if type_name not in _ABI_KIND:
            raise NotImplementedError("only Hexahedron and Tetrahedron cells are integrated on the device (got %s)" % type_name)
        idx = self.cells_of(type_name)
        conn = np.array([self.cells[i] for i in idx], dtype=np.int32).reshape(len(idx), NVERTS[type_name])
        xyz = self.nodes if self.nodes.shape[1] == 3 else np.hstack([self.nodes, np.zeros((len(self.nodes), 3 - self.nodes.shape[1]))])
        g = Grid(_ABI_KIND[type_name], xyz, conn)
        g.file_cell_index = np.array(idx, dtype=np.int64)
        return g


def _lines(path):
    with open(path, "r") as fh:
        for raw in fh:
            yield raw.strip()


def _add(sets, attr, ei):
    sets.setdefault(str(attr), []).append(ei)


# ------------------------------------------------------------------------------------------------ openCARP
_CARP = {"Ln": LINE, "Tr": TRIANGLE, "Qd": QUADRILATERAL, "Tt": TETRAHEDRON, "Pr": WEDGE, "Hx": HEXAHEDRON}


def load_carp_elements(filename):
    """load_carp_elements (src/mesh/tools.jl:585-643): `<n>` then one `<tag> v… [attr]` line per element, 0-based ids."""
    it = _lines(filename)
    ne = int(next(it).split()[0])
    types, cells, sets = [None] * ne, [None] * ne, {}
    for ei in range(ne):
        try:
            tok = next(it).split()
        except StopIteration:
            raise ValueError("Premature end of input file")
        t = _CARP.get(tok[0])
        if t is None:
            continue                                   # unknown element type: skipped, like the reference
        nv = NVERTS[t]
        types[ei], cells[ei] = t, [int(v) for v in tok[1:1 + nv]]
        if len(tok) == nv + 2:
            _add(sets, int(tok[-1]), ei)
    return types, cells, sets


def load_carp_nodes(filename):
    it = _lines(filename)
    nv = int(next(it).split()[0])
    nodes = np.empty((nv, 3))
    for ni in range(nv):
        try:
            tok = next(it).split()
        except StopIteration:
            raise ValueError("Premature end of input file")
        nodes[ni] = [float(tok[0]), float(tok[1]), float(tok[2])]
    return nodes


def load_carp_fibres(filename):
    """openCARP .lon: first line = number of direction vectors per element (1: f, 2: f and s), then one line per
    element.  (The reference has no .lon reader; the nodal/elementwise frames enter through coefficient fields.)"""
    it = _lines(filename)
    nvec = int(next(it).split()[0])
    rows = [[float(v) for v in ln.split()] for ln in it if ln]
    a = np.array(rows, dtype=np.float64).reshape(len(rows), 3 * nvec)
    return (a[:, :3],) if nvec == 1 else (a[:, :3], a[:, 3:6])


def load_carp_grid(filename):
    """load_carp_grid(filename) (src/mesh/tools.jl:660-665): `filename.elem` + `filename.pts`."""
    types, cells, sets = load_carp_elements(filename + ".elem")
    return MixedGrid(types, cells, load_carp_nodes(filename + ".pts"), sets)


# ------------------------------------------------------------------------------------------------ MFEM v1.0
def load_mfem_grid(filename):
    """load_mfem_grid (src/mesh/tools.jl:497-583): straight meshes, format v1.0; boundary section skipped."""
    it = _lines(filename)
    fmt = next(it)
    if fmt != "MFEM mesh v1.0":
        raise ValueError("Unsupported mesh format '%s'" % fmt)

    def seek(word):
        for ln in it:
            if ln == word:
                return
        raise ValueError("Missing '%s' specification" % word)

    seek("dimension")
    sdim = int(next(it))
    seek("elements")
    ne = int(next(it))
    types, cells, sets = [None] * ne, [None] * ne, {}
    for ei in range(ne):
        tok = [int(v) for v in next(it).split()]
        attr, etype, v = tok[0], tok[1], tok[2:]
        if etype == 1:
            types[ei], cells[ei] = LINE, v[:2]
        elif etype == 2:
            types[ei], cells[ei] = TRIANGLE, (v[1], v[2], v[0])          # tools.jl:531
        elif etype == 3:
            types[ei], cells[ei] = QUADRILATERAL, v[:4]
        elif etype == 4:
            types[ei], cells[ei] = TETRAHEDRON, v[:4]
        elif etype == 5:
            types[ei], cells[ei] = HEXAHEDRON, v[:8]
        elif etype == 6:
            types[ei], cells[ei] = WEDGE, v[:6]
        elif etype == 7:
            types[ei], cells[ei] = PYRAMID, (v[0], v[1], v[3], v[2], v[4])  # tools.jl:541
        _add(sets, attr, ei)
    seek("vertices")
    nv = int(next(it))
    if int(next(it)) != sdim:
        raise ValueError("vertex dimension does not match 'dimension'")
    nodes = np.empty((nv, sdim))
    for vi in range(nv):
        nodes[vi] = [float(x) for x in next(it).split()[:sdim]]
    return MixedGrid(types, cells, nodes, sets)


# ------------------------------------------------------------------------------------------------ voom2
def load_voom2_elements(filename):
    """load_voom2_elements (src/mesh/tools.jl:429-451): `<n> …` then `<id> <type> v…`, 1-based ids; 8 = hex, 2 = line."""
    it = _lines(filename)
    ne = int(next(it).split()[0])
    types, cells = [None] * ne, [None] * ne
    for ln in it:
        if not ln:
            continue
        tok = [int(v) for v in ln.split()]
        ei, etype = tok[0] - 1, tok[1]
        if etype == 8:
            types[ei], cells[ei] = HEXAHEDRON, [v - 1 for v in tok[2:10]]
        elif etype == 2:
            types[ei], cells[ei] = LINE, [v - 1 for v in tok[2:4]]
    return types, cells


def load_voom2_nodes(filename):
    it = _lines(filename)
    nn = int(next(it).split()[0])
    nodes = np.zeros((nn, 3))
    for ln in it:
        if not ln:
            continue
        tok = ln.split()
        nodes[int(tok[0]) - 1] = [float(tok[1]), float(tok[2]), float(tok[3])]
    return nodes


def load_voom2_fsn(filename):
    """nine numbers per line: f, s, n (src/mesh/tools.jl:470-484)."""
    a = np.array([[float(v) for v in ln.split()] for ln in _lines(filename) if ln], dtype=np.float64).reshape(-1, 9)
    return a[:, :3], a[:, 3:6], a[:, 6:9]


def load_voom2_grid(filename):
    types, cells = load_voom2_elements(filename + ".ele")
    return MixedGrid(types, cells, load_voom2_nodes(filename + ".nodes"))
