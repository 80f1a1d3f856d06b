# make_golden.jl — maintainer-side pin for the conventions this repository could not verify ([UNPINNED] in DESIGN.md §2):
# Ferrite's node / cell order of generate_grid, the dof numbering of close!(dh), the column order of allocate_matrix,
# the Gauss-point order of the default rules, and whole-mesh values of M, K and b.
#
# NOT runnable in the build image (no Julia, no network).  Run it where Thunderbolt.jl v0.0.4 and its Ferrite are installed:
#
#     julia --project=<Thunderbolt checkout> julia/make_golden.jl tests/golden
#
# It writes   tests/golden/ferrite_box_4x3x5.json   (perturbed hexahedral box, first-order scalar field) and
#             tests/golden/ferrite_quad_287x1.json  (the mesh of test/gpu/test_operators.jl:1-31)
# and tests/test_ferrite_golden.py then checks the oracle — mesh generator, dof table, sparsity pattern, quadrature order,
# mass / diffusion matrices and the two linear forms — against them entry by entry (it reports "unpinned" while the files
# are absent).  Everything is written 0-based.  No JSON package is needed.
#
# What each block follows in the reference:
#   mesh + dofs      Ferrite.generate_grid, DofHandler / add! / close!           (test/gpu/test_operators.jl:4-7)
#   pattern          transpose(allocate_matrix(dh)) read as CSR                    (src/solver/interface.jl:162-168)
#   cell loop        start_assemble / CellIterator / assemble!                     (src/modeling/core/coordinate_systems.jl:145-171)
#   element routines Thunderbolt.assemble_element! of the mass, diffusion and analytical-coefficient caches
#                    (src/modeling/core/mass.jl:28-43, diffusion.jl:28-50, analytical_coefficient.jl:80-101)
using Thunderbolt, Ferrite, SparseArrays, LinearAlgebra, StaticArrays, Tensors

outdir = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "tests", "golden")

# ---- minimal JSON writer (numbers, vectors, vectors of vectors, string keys) ----
jnum(x::Integer) = string(x)
jnum(x::AbstractFloat) = repr(Float64(x))            # shortest round-trip form
jval(x::Number) = jnum(x)
jval(x::AbstractString) = "\"" * x * "\""
jval(x::Union{AbstractVector, Tuple}) = "[" * join((jval(v) for v in x), ",") * "]"
function write_json(path, pairs)
    open(path, "w") do io
        println(io, "{")
        for (k, (key, val)) in enumerate(pairs)
            print(io, "  \"", key, "\": ", jval(val))
            println(io, k == length(pairs) ? "" : ",")
        end
        println(io, "}")
    end
    @info "wrote $path"
end

# the node perturbation of tb_host_perturb_nodes (thunderbolt.jl_amd/csrc/tb_hostgen.cpp): lattice index (i, j, k), x fastest
function perturbed(grid::Grid{3}, nel, amplitude)
    nx, ny, nz = nel
    px, py = nx + 1, ny + 1
    X = [collect(n.x) for n in grid.nodes]
    last = X[end]; first_ = X[1]
    h = ((last[1] - first_[1]) / nx, (last[2] - first_[2]) / ny, (last[3] - first_[3]) / nz)
    nodes = similar(grid.nodes)
    for id0 in 0:length(X)-1
        i = id0 % px; j = (id0 ÷ px) % py; k = id0 ÷ (px * py)
        s = sin(2π * i / nx) * sin(2π * j / ny) * sin(2π * k / nz)
        x = X[id0 + 1]
        nodes[id0 + 1] = Node(Vec((x[1] + amplitude * h[1] * s, x[2] - 0.5 * amplitude * h[2] * s, x[3] + 0.75 * amplitude * h[3] * s)))
    end
    return Grid(grid.cells, nodes; facetsets = grid.facetsets)
end

# the sequential loop of coordinate_systems.jl:145-171 around one of Thunderbolt's element caches
function assemble_bilinear(dh, integrator)
    K = allocate_matrix(dh)
    assembler = start_assemble(K)
    sdh = first(dh.subdofhandlers)
    cache = Thunderbolt.setup_element_cache(integrator, sdh)
    n = ndofs_per_cell(sdh)
    Ke = zeros(n, n)
    for cell in CellIterator(sdh)
        fill!(Ke, 0.0)
        Thunderbolt.assemble_element!(Ke, cell, cache, 0.0)
        assemble!(assembler, celldofs(cell), Ke)
    end
    return K
end

function assemble_linear(dh, f, qrc, t)
    grid = dh.grid
    sdh = first(dh.subdofhandlers)
    ac = AnalyticalCoefficient(f, CoordinateSystemCoefficient(CartesianCoordinateSystem(grid)))
    qr = getquadraturerule(qrc, sdh)
    ip = Ferrite.getfieldinterpolation(sdh, first(sdh.dh.field_names))
    cv = CellValues(qr, ip)
    cc = Thunderbolt.setup_coefficient_cache(ac, qr, sdh)
    cache = Thunderbolt.AnalyticalCoefficientElementCache(cc, [SVector((-Inf, Inf))], cv)
    b = zeros(ndofs(dh))
    n = ndofs_per_cell(sdh)
    be = zeros(n)
    for cell in CellIterator(sdh)
        fill!(be, 0.0)
        Thunderbolt.assemble_element!(be, cell, cache, t)
        b[celldofs(cell)] .+= be
    end
    return b
end

# CSR arrays of a Ferrite CSC matrix, the way create_system_matrix makes them (interface.jl:162-168)
function csr(K::SparseMatrixCSC)
    Kt = SparseMatrixCSC(transpose(K))
    return Kt.colptr .- 1, Kt.rowval .- 1, Kt.nzval
end

function mesh_block(grid, dh)
    sdh = first(dh.subdofhandlers)
    cd = [collect(celldofs(c)) .- 1 for c in CellIterator(sdh)]
    return [
        "nodes" => [collect(n.x) for n in grid.nodes],
        "cells" => [collect(c.nodes) .- 1 for c in grid.cells],
        "celldofs" => cd,
        "ndofs" => ndofs(dh),
    ]
end

# ---- 1. perturbed 4 × 3 × 5 hexahedral box, Q1 scalar field ----
let nel = (4, 3, 5)
    grid = perturbed(generate_grid(Hexahedron, nel, Vec((0.0, 0.0, 0.0)), Vec((1.0, 1.0, 1.0))), nel, 0.1)
    dh = DofHandler(grid)
    add!(dh, :u, Lagrange{RefHexahedron, 1}())
    close!(dh)
    qrc = QuadratureRuleCollection(2)
    sdh = first(dh.subdofhandlers)
    qr = getquadraturerule(qrc, sdh)
    D = SymmetricTensor{2, 3}((4.5e-5, 1.0e-5, 0.5e-5, 2.0e-5, 0.3e-5, 1.0e-5)) # the KAPPA_FULL of tests/test_gpu_parity.py (order: xx, yx, zx, yy, zy, zz)
    M = assemble_bilinear(dh, BilinearMassIntegrator(ConstantCoefficient(1.0), qrc, :u))
    K = assemble_bilinear(dh, BilinearDiffusionIntegrator(ConstantCoefficient(D), qrc, :u))
    rowptr, colval, Mnz = csr(M)
    _, _, Knz = csr(K)
    b = assemble_linear(dh, (x, t) -> norm(x) + t, qrc, 0.0)   # benchmarks/benchmarks-linear-form.jl:16-27
    write_json(joinpath(outdir, "ferrite_box_4x3x5.json"), vcat(
        ["generator" => "julia/make_golden.jl", "nel" => collect(nel), "perturb" => 0.1],
        mesh_block(grid, dh),
        ["rowptr" => rowptr, "colval" => colval,
         "gauss_points" => [collect(p) for p in Ferrite.getpoints(qr)], "gauss_weights" => collect(Ferrite.getweights(qr)),
         "D" => [D[1, 1], D[1, 2], D[1, 3], D[2, 1], D[2, 2], D[2, 3], D[3, 1], D[3, 2], D[3, 3]],
         "M" => Mnz, "K" => Knz, "b_norm_x_plus_t" => b]))
end

# ---- 2. the 287 × 1 quadrilateral mesh of test/gpu/test_operators.jl ----
let nel = (287, 1)
    grid = generate_grid(Quadrilateral, nel, Vec((-1.0, -1.0)), Vec((1.0, 1.0)))
    dh = DofHandler(grid)
    add!(dh, :u, Lagrange{RefQuadrilateral, 1}())
    close!(dh)
    qrc = QuadratureRuleCollection(2)
    K = allocate_matrix(dh)
    rowptr, colval, _ = csr(K)
    b = assemble_linear(dh, (x, t) -> cos(2π * t) * exp(-norm(x)^2), qrc, 0.0)
    write_json(joinpath(outdir, "ferrite_quad_287x1.json"), vcat(
        ["generator" => "julia/make_golden.jl", "nel" => collect(nel)],
        mesh_block(grid, dh),
        ["rowptr" => rowptr, "colval" => colval, "b_cos_exp" => b]))
end
