"""Shared test helpers (test infrastructure): comparison norms and small mesh builders."""
import numpy as np


def rel_err(got, ref):
    """SURVEY §8 parity norms: relative 2-norm (Frobenius for nz vectors) and max-entrywise relative to max|ref|."""
    got, ref = np.asarray(got), np.asarray(ref)
    d = np.abs(got - ref)
    scale2 = np.linalg.norm(ref) or 1.0
    scalem = np.abs(ref).max() or 1.0
    return max(np.linalg.norm(d) / scale2, d.max() / scalem)


def hex_to_tets(xyz, conn):
    """Split every hexahedron into 6 positively oriented tetrahedra around the 0–6 diagonal."""
    T = [(0, 1, 2, 6), (0, 2, 3, 6), (0, 3, 7, 6), (0, 7, 4, 6), (0, 4, 5, 6), (0, 5, 1, 6)]
    tets = np.concatenate([conn[:, list(t)] for t in T], axis=0).astype(np.int32)
    # orient: det > 0
    X = xyz[tets]
    det = np.einsum("ij,ij->i", np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), X[:, 3] - X[:, 0])
    flip = det < 0
    tets[flip] = tets[flip][:, [0, 2, 1, 3]]
    return np.ascontiguousarray(tets)
