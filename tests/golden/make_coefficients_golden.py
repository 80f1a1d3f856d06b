"""Writes tests/golden/coefficients.json: the known-answer vectors that the reference's own test-suite
holds for the coefficient evaluators (Thunderbolt.jl test/test_coefficients.jl:24-188), as data:
inputs (mesh, quadrature points, coefficient parameters) and expected outputs.  Line numbers cite where
each expectation is asserted in the reference.  Mesh: generate_grid(Line,(2,)) on [-1,1] → cell 1 =
[-1,0], cell 2 = [0,1]; quadrature points ξ = 0.0 and ξ = 0.1 (test_coefficients.jl:6-10)."""
import json
import os

cells = [[[-1.0], [0.0]], [[0.0], [1.0]]]
xi = [0.0, 0.1]
G = {"mesh": {"kind": "Line", "cells": cells, "xi": xi}, "cases": []}
add = G["cases"].append

# ConstantCoefficient, :24-37
for val in (1.0, [[1.0, 0.0], [0.0, 1.0]]):
    add({"name": "constant", "ref": "test_coefficients.jl:24-37", "value": val,
         "expect": [{"cell": c, "qp": q, "t": t, "out": val} for c in (0, 1) for q, t in ((0, 0.0), (1, 1.0))]})

# FieldCoefficient scalar, :39-54   data[basis, cell]
add({"name": "field_scalar", "ref": "test_coefficients.jl:39-54",
     "data": [[1.0, -1.0], [-1.0, 0.0]],  # [cell][basis]
     "expect": [{"cell": 0, "qp": 0, "out": 0.0}, {"cell": 0, "qp": 1, "out": -0.1},
                {"cell": 1, "qp": 0, "out": -0.5}, {"cell": 1, "qp": 1, "out": (0.1 + 1.0) / 2.0 - 1.0}]})

# FieldCoefficient vector, :56-69
add({"name": "field_vector", "ref": "test_coefficients.jl:56-69",
     "data": [[[1.0, 0.0], [-1.0, -0.0]], [[0.0, -1.0], [0.0, 0.0]]],  # [cell][basis][comp]
     "expect": [{"cell": 0, "qp": 0, "out": [0.0, 0.0]}, {"cell": 0, "qp": 1, "out": [-0.1, 0.0]},
                {"cell": 1, "qp": 0, "out": [0.0, -0.5]}, {"cell": 1, "qp": 1, "out": [0.0, (0.1 + 1.0) / 2.0 - 1.0]}]})

# CartesianCoordinateSystem, :74-89
add({"name": "cartesian", "ref": "test_coefficients.jl:74-89",
     "expect": [{"cell": 0, "qp": 0, "out": [-0.5]}, {"cell": 0, "qp": 1, "out": [-0.45]},
                {"cell": 1, "qp": 0, "out": [0.5]}, {"cell": 1, "qp": 1, "out": [0.55]}]})

# AnalyticalCoefficient (x,t) -> norm(x)+t, :91-106
add({"name": "analytical_norm_plus_t", "ref": "test_coefficients.jl:91-106",
     "expect": [{"cell": 0, "qp": 0, "t": 0.0, "out": 0.5}, {"cell": 0, "qp": 1, "t": 0.0, "out": 0.45},
                {"cell": 0, "qp": 0, "t": 1.0, "out": 1.5}, {"cell": 0, "qp": 1, "t": 1.0, "out": 1.45},
                {"cell": 1, "qp": 0, "t": 0.0, "out": 0.5}, {"cell": 1, "qp": 1, "t": 0.0, "out": 0.55},
                {"cell": 1, "qp": 0, "t": 1.0, "out": 1.5}, {"cell": 1, "qp": 1, "t": 1.0, "out": 1.55}]})

# SpectralTensorCoefficient, :108-142
add({"name": "spectral_transverse", "ref": "test_coefficients.jl:108-125", "f": [1.0, 0.0], "lambda": [-1.0, 0.0],
     "out": [[-1.0, 0.0], [0.0, 0.0]]})
add({"name": "spectral_transverse", "ref": "test_coefficients.jl:127-133", "f": [1.0, 0.0], "lambda": [-1.0, -1.0],
     "out": [[-1.0, 0.0], [0.0, -1.0]]})
add({"name": "spectral_planar", "ref": "test_coefficients.jl:135-140", "f": [1.0, 0.0], "s": [0.0, 1.0],
     "lambda": [-1.0, -1.0], "out": [[-1.0, 0.0], [0.0, -1.0]]})

# SpatiallyHomogeneousDataField, :144-163
add({"name": "homogeneous_data", "ref": "test_coefficients.jl:144-163", "timings": [1.0, 2.0], "data": [0.1, 0.2, 0.3],
     "expect": [{"t": 0.0, "out": 0.1}, {"t": 1.0, "out": 0.1}, {"t": 1.1, "out": 0.2}, {"t": 2.0, "out": 0.2},
                {"t": 2.1, "out": 0.3}]})

# ConductivityToDiffusivityCoefficient, :165-188   κ/(Cₘ·χ) with Cₘ=2, χ=0.5
add({"name": "conductivity_to_diffusivity", "ref": "test_coefficients.jl:165-188", "f": [1.0, 0.0], "lambda": [-1.0, 0.0],
     "Cm": 2.0, "chi": 0.5, "out": [[-1.0, 0.0], [0.0, 0.0]]})

# distorted cells of the "Static interpolation values" test, :190-218 (inputs only; the expectations there
# are identities against Ferrite CellValues, restated in tests as Σ detJ·w = volume, Σ∇N = 0, Σ x⊗∇N = I)
G["distorted_cells"] = {
    "ref": "test_coefficients.jl:195-218",
    "hex8": [[0.0, 0.0, 0.0], [1.3, 0.1, 0.0], [1.1, 1.4, -0.2], [0.2, 1.0, 0.1], [-0.1, 0.2, 1.2], [1.5, 0.0, 1.0],
             [1.2, 1.1, 1.4], [0.0, 1.3, 1.1]],
    "tet4": [[0.0, 0.0, 0.0], [1.7, 0.2, 0.1], [0.3, 1.4, -0.1], [0.1, 0.2, 1.9]],
    "ue_hex8": [0.3, -1.2, 0.7, 2.1, 0.4, -0.6, 1.5, 0.9],  # :251
    "ue_tet4": [0.4, 1.1, -0.8, 0.25],  # :284
}

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "coefficients.json"), "w") as fh:
    json.dump(G, fh, indent=1)
