# ThunderboltHIPBackend.jl — the reference-side binding a Thunderbolt.jl maintainer would add (as a package extension next to
# ext/CuThunderboltExt.jl) to use libtbhip.so as an `AbstractGPUDevice` backend for the assembly + reaction hot path.
#
# NOT executed in the build container (no Julia toolchain there; stated in DESIGN.md and INTEGRATION.md).  Every `ccall` targets a symbol
# declared in include/tbhip.h with the argument order of that header; tests/abi_driver.c exercises the same entry points in the same order
# from C, and thunderbolt.jl_amd/api.py is the executable mirror.
module ThunderboltHIPBackend

using Thunderbolt, Ferrite, Tensors, SparseArrays, SparseMatricesCSR, LinearAlgebra
import LinearAlgebra: mul!
import Thunderbolt: AbstractGPUDevice, AbstractAssemblyStrategy, AbstractSolver, AbstractSemidiscreteFunction, setup_operator,
    update_operator!, update_linearization!, residual!, add!, needs_update, _pointwise_step_outer_kernel!, create_system_vector,
    create_system_matrix, __add_to_vector!, adapt_vector_type, value_type, index_type, solution_size, AbstractPointwiseSolverCache,
    AbstractPointwiseFunction, num_states, BilinearMassIntegrator, BilinearDiffusionIntegrator, LinearIntegrator, ConstantCoefficient,
    FieldCoefficient, ConductivityToDiffusivityCoefficient, SpectralTensorCoefficient, OrthotropicMicrostructure,
    TransverselyIsotropicMicrostructure, OrthotropicMicrostructureModel, PerColorAssemblyStrategy, ElementAssemblyStrategy,
    QuasiStaticModel, PK1Model, HolzapfelOgden2009Model

const libtbhip = get(ENV, "TBHIP_LIBRARY", "libtbhip.so")

# revision of include/tbhip.h these ccalls were written against (TB_ABI_REVISION); a library of another revision reads / writes other buffer sizes
const TB_ABI_REVISION = 6
const TB_ERR_UNSUPPORTED = Cint(-5) # include/tbhip.h
function __init__()
    have = ccall((:tb_abi_revision, libtbhip), Cint, ())
    have == TB_ABI_REVISION || error("libtbhip ABI revision $have, this binding was written against $TB_ABI_REVISION")
end

check(rc::Cint) = rc == 0 ? nothing :
    error("libtbhip error $rc: " * unsafe_string(ccall((:tb_last_error_string, libtbhip), Cstring, ())))

# ---------------------------------------------------------------- device (replaces FerriteOperators.CudaDevice, ext/CuThunderboltExt.jl:48-49)
mutable struct MI355XDevice{Tv, Ti} <: AbstractGPUDevice
    handle::Ptr{Cvoid}
    function MI355XDevice{Tv, Ti}(id::Integer = 0) where {Tv, Ti}
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:tb_device_create, libtbhip), Cint, (Cint, Ref{Ptr{Cvoid}}), id, h))
        dev = new{Tv, Ti}(h[])
        finalizer(d -> ccall((:tb_device_destroy, libtbhip), Cint, (Ptr{Cvoid},), d.handle), dev)
        return dev
    end
end
# the kernel instance the latest assembly call launched (benchmark lines)
last_kernel_name() = unsafe_string(ccall((:tb_last_kernel_name, libtbhip), Cstring, ()))
MI355XDevice(id = 0) = MI355XDevice{Float64, Int32}(id)
value_type(::MI355XDevice{Tv}) where {Tv} = Tv
index_type(::MI355XDevice{Tv, Ti}) where {Tv, Ti} = Ti
# deferred status: inside a time loop on a fixed mesh the assembly calls then only enqueue their kernels; `poll_status` synchronises and throws the first
# assembly error (detJ ≤ 0, coupling missing from the pattern) raised since the last poll
defer_status!(dev::MI355XDevice, on::Bool = true) = check(ccall((:tb_device_defer_status, libtbhip), Cint, (Ptr{Cvoid}, Cint), dev.handle, on ? 1 : 0))
poll_status(dev::MI355XDevice) = check(ccall((:tb_device_poll_status, libtbhip), Cint, (Ptr{Cvoid},), dev.handle))

# ---------------------------------------------------------------- vectors: an opaque, GC-managed device buffer (finaliser → tb_free).
# Deliberately NOT an AbstractVector: there is no scalar indexing on the device, and the solvers that consume it are the library's own
# (tb_cg_solve …), reached through the methods below — not LinearSolve's generic fallbacks.
mutable struct HIPVector{T}
    dev::MI355XDevice
    ptr::Ptr{T}
    n::Int
    function HIPVector{T}(dev::MI355XDevice, n::Integer) where {T}
        p = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:tb_malloc, libtbhip), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), dev.handle, n * sizeof(T), p))
        v = new{T}(dev, Ptr{T}(p[]), n)
        finalizer(x -> ccall((:tb_free, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), x.dev.handle, x.ptr), v)
        return fill!(v, zero(T))
    end
end
Base.length(v::HIPVector) = v.n
Base.eltype(::HIPVector{T}) where {T} = T
Base.similar(v::HIPVector{T}) where {T} = HIPVector{T}(v.dev, v.n)
function Base.fill!(v::HIPVector{T}, x) where {T}
    iszero(x) || error("HIPVector: only zero-fill is provided")
    check(ccall((:tb_memset, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Csize_t), v.dev.handle, v.ptr, 0, v.n * sizeof(T)))
    return v
end
function Base.copyto!(v::HIPVector{T}, a::Vector{T}) where {T}
    length(a) == v.n || throw(DimensionMismatch())
    check(ccall((:tb_memcpy_h2d, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), v.dev.handle, v.ptr, a, sizeof(a)))
    return v
end
function Base.copyto!(v::HIPVector{T}, w::HIPVector{T}) where {T}
    check(ccall((:tb_memcpy_d2d, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), v.dev.handle, v.ptr, w.ptr, v.n * sizeof(T)))
    return v
end
function Base.Vector(v::HIPVector{T}) where {T}
    a = Vector{T}(undef, v.n)
    check(ccall((:tb_memcpy_d2h, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), v.dev.handle, a, v.ptr, sizeof(a)))
    return a
end
HIPVector(dev::MI355XDevice, a::Vector{T}) where {T} = copyto!(HIPVector{T}(dev, length(a)), a)

# the device a vector type lives on: one process drives one GPU (one rank per GPU under torch.distributed / MPI)
const CURRENT_DEVICE = Ref{Union{Nothing, MI355XDevice}}(nothing)
current_device() = something(CURRENT_DEVICE[], (CURRENT_DEVICE[] = MI355XDevice(); CURRENT_DEVICE[]))

# ext/CuThunderboltExt.jl:126-127
create_system_vector(::Type{<:HIPVector{T}}, f::AbstractSemidiscreteFunction) where {T} = HIPVector{T}(current_device(), solution_size(f))
create_system_vector(::Type{<:HIPVector{T}}, dh::DofHandler) where {T} = HIPVector{T}(current_device(), ndofs(dh))
# ext/CuThunderboltExt.jl:141-146
__add_to_vector!(b::Vector, a::HIPVector) = b .+= Vector(a)
function __add_to_vector!(b::HIPVector{T}, a::Vector{T}) where {T}
    tmp = HIPVector(b.dev, a)
    check(ccall((:tb_axpy, libtbhip), Cint, (Ptr{Cvoid}, Int64, Cdouble, Ptr{T}, Ptr{T}), b.dev.handle, b.n, 1.0, tmp.ptr, b.ptr))
    return b
end
adapt_vector_type(::Type{<:HIPVector}, v::Vector) = HIPVector(current_device(), v)

# ---------------------------------------------------------------- strategies (src/Thunderbolt.jl:22-32): the reference's own strategy types carry the
# device; the patch strategy is this backend's addition
struct PatchAssemblyStrategy{D <: MI355XDevice} <: AbstractAssemblyStrategy
    device::D
end
const HIPStrategy = Union{PatchAssemblyStrategy, PerColorAssemblyStrategy{<:MI355XDevice}, ElementAssemblyStrategy{<:MI355XDevice}}
strategy_code(::PerColorAssemblyStrategy) = Cint(1)      # TB_STRATEGY_PER_COLOR
strategy_code(::ElementAssemblyStrategy) = Cint(2)       # TB_STRATEGY_ELEMENT
strategy_code(::PatchAssemblyStrategy) = Cint(3)         # TB_STRATEGY_PATCH

# ---------------------------------------------------------------- mesh, dof table, pattern: Ferrite's own arrays go over as they are, once per DofHandler
# (all operators of one DofHandler share one sparsity pattern, src/solver/time/euler.jl:110-116, newmark.jl:105-110)
mutable struct DeviceDofHandler
    dev::MI355XDevice
    mesh::Ptr{Cvoid}
    pattern::Ptr{Cvoid}
    nnz::Int
    size::Tuple{Int, Int}
    cpu_pattern::SparseMatrixCSR{1, Float64, Int64}   # kept for hosts that want rowptr / colval on the CPU side
end
const DOFHANDLERS = IdDict{Any, DeviceDofHandler}()

function DeviceDofHandler(dev::MI355XDevice, dh::DofHandler)
    haskey(DOFHANDLERS, dh) && return DOFHANDLERS[dh]
    grid = Ferrite.get_grid(dh)
    sdh = only(dh.subdofhandlers)                       # one subdomain, one field (SURVEY §8b)
    sdim = Ferrite.getspatialdim(grid)
    xyz = Float64[i <= sdim ? n.x[i] : 0.0 for n in grid.nodes for i in 1:3]           # AoS, 3 per node; 2-D meshes travel with z = 0 (TB_QUAD4)
    conn = Int32[v for c in grid.cells for v in c.nodes]                                # 1-based
    celldofs = Int32.(dh.cell_dofs)                                                     # dh.cell_dofs / cell_dofs_offset, src/utils.jl:52-56
    kind = grid.cells[1] isa Quadrilateral ? Cint(2) : grid.cells[1] isa Hexahedron ? Cint(3) : Cint(4)   # TB_QUAD4 / TB_HEX8 / TB_TET4
    ip = Ferrite.getfieldinterpolation(sdh, first(sdh.field_names))
    fkind = (kind == 3 && Ferrite.getorder(ip) == 2) ? Cint(5) : kind                  # TB_HEX27
    ncomp = Ferrite.n_components(ip)
    mesh = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:tb_mesh_create, libtbhip), Cint,
        (Ptr{Cvoid}, Cint, Int64, Ptr{Float64}, Int64, Ptr{Int32}, Cint, Cint, Ptr{Int32}, Int64, Cint, Ref{Ptr{Cvoid}}),
        dev.handle, kind, length(grid.nodes), xyz, length(grid.cells), conn, fkind, ncomp, celldofs, ndofs(dh), 1, mesh))
    A = SparseMatrixCSR(transpose(allocate_matrix(dh)))                                 # src/solver/interface.jl:162-168
    pat = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:tb_pattern_create, libtbhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int32}, Cint, Ref{Ptr{Cvoid}}),
        mesh[], size(A, 1), Int64.(A.rowptr), Int32.(A.colval), 1, pat))
    ddh = DeviceDofHandler(dev, mesh[], pat[], nnz(A), size(A), A)
    finalizer(ddh) do d
        ccall((:tb_pattern_destroy, libtbhip), Cint, (Ptr{Cvoid},), d.pattern)
        ccall((:tb_mesh_destroy, libtbhip), Cint, (Ptr{Cvoid},), d.mesh)
    end
    return DOFHANDLERS[dh] = ddh
end

# Locality numbering for a DofHandler on an arbitrarily numbered grid (a mesh read from a file): call BEFORE the first setup_operator /
# create_system_matrix of this DofHandler.  The reference's cell loop does not care how a mesh is numbered (src/modeling/core/coordinate_systems.jl:145-171);
# the device plans do (shared scatter signatures, SpMV gathers that stay in the caches).  tb_host_locality_permutation sweeps the per-axis cell layers the
# patch planner cuts and numbers the dofs by first visit in that order — what close!(dh) would number on a grid stored in that order.
function locality_renumber!(dh::DofHandler)
    grid = Ferrite.get_grid(dh)
    sdim = Ferrite.getspatialdim(grid)
    xyz = Float64[i <= sdim ? n.x[i] : 0.0 for n in grid.nodes for i in 1:3]
    conn = Int32[v for c in grid.cells for v in c.nodes]
    celldofs = Int32.(dh.cell_dofs)
    kind = grid.cells[1] isa Quadrilateral ? Cint(2) : grid.cells[1] isa Hexahedron ? Cint(3) : Cint(4)
    perm = Vector{Int32}(undef, ndofs(dh))
    check(ccall((:tb_host_locality_permutation, libtbhip), Cint,
        (Cint, Int64, Ptr{Float64}, Int64, Ptr{Int32}, Cint, Ptr{Int32}, Int64, Cint, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
        kind, length(grid.nodes), xyz, length(grid.cells), conn, Ferrite.ndofs_per_cell(dh), celldofs, ndofs(dh), 1, C_NULL, C_NULL, perm))
    Ferrite.renumber!(dh, Int.(perm))                    # perm[d] = new number of dof d (Ferrite's convention)
    return dh
end

# device CSR matrix: the shared pattern + one nzval vector (rowptr / colval stay inside the library: tb_pattern_{rowptr,colidx}_device)
struct HIPSparseMatrixCSR{Tv}
    ddh::DeviceDofHandler
    nzval::HIPVector{Tv}
end
Base.size(A::HIPSparseMatrixCSR) = A.ddh.size
SparseArrays.nnz(A::HIPSparseMatrixCSR) = A.ddh.nnz
SparseArrays.nonzeros(A::HIPSparseMatrixCSR) = A.nzval
# ext/CuThunderboltExt.jl:129-139
function create_system_matrix(::Type{<:HIPSparseMatrixCSR{Tv}}, dh::DofHandler) where {Tv}
    ddh = DeviceDofHandler(current_device(), dh)
    return HIPSparseMatrixCSR{Tv}(ddh, HIPVector{Tv}(ddh.dev, ddh.nnz))
end
# y = α·A·x + β·y (src/utils.jl:185-231)
function mul!(y::HIPVector{Tv}, A::HIPSparseMatrixCSR{Tv}, x::HIPVector{Tv}, α::Number = true, β::Number = false) where {Tv}
    check(ccall((:tb_spmv_csr, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Tv}, Ptr{Tv}, Cdouble, Cdouble, Ptr{Tv}),
        A.ddh.pattern, A.nzval.ptr, x.ptr, α, β, y.ptr))
    return y
end

# ---------------------------------------------------------------- coefficients → tb_coef (closures cannot cross the ABI, SURVEY F10)
struct TbCoef
    kind::Int32; wrap::Int32; Cm::Float64; chi::Float64
    p::NTuple{16, Float64}; field::Ptr{Float64}; field_len::Int64
end
pad16(v) = ntuple(i -> i <= length(v) ? Float64(v[i]) : 0.0, 16)
v3(x::Vec{3}) = (x[1], x[2], x[3])
v3(x::Vec{2}) = (x[1], x[2], 0.0)
# returns (TbCoef, keepalive): the field array must outlive tb_form_create, which copies it
lower(c::ConstantCoefficient{<:Real}) = (TbCoef(0, 0, 1.0, 1.0, pad16((c.val,)), C_NULL, 0), nothing)                     # TB_COEF_CONST_SCALAR
function lower(c::ConstantCoefficient{<:Tensors.SecondOrderTensor})                                                         # TB_COEF_CONST_TENSOR, row-major 3×3
    D = zeros(3, 3); d = size(c.val, 1); D[1:d, 1:d] .= Matrix(c.val)
    return (TbCoef(1, 0, 1.0, 1.0, pad16(vec(permutedims(D))), C_NULL, 0), nothing)
end
function lower(c::FieldCoefficient)                                                                                         # TB_COEF_FIELD_SCALAR, field[basis + nb·cell]
    data = Float64.(vec(c.elementwise_data.data))
    return (TbCoef(2, 0, 1.0, 1.0, pad16(()), pointer(data), length(data)), data)
end
function lower(c::SpectralTensorCoefficient)
    λ = c.eigenvalues isa ConstantCoefficient ? Tuple(c.eigenvalues.val) : error("spectral coefficient: constant eigenvalues only")
    ev = c.eigenvectors
    if ev isa ConstantCoefficient{<:OrthotropicMicrostructure}                                                              # TB_COEF_SPECTRAL_CONST: f, s, n, λ
        m = ev.val
        return (TbCoef(3, 0, 1.0, 1.0, pad16((v3(m.f)..., v3(m.s)..., v3(m.n)..., λ...)), C_NULL, 0), nothing)
    elseif ev isa ConstantCoefficient{<:TransverselyIsotropicMicrostructure}                                                # TB_COEF_TRANSVERSE_CONST: f, λ₁, λ₂
        return (TbCoef(5, 0, 1.0, 1.0, pad16((v3(ev.val.f)..., λ[1], λ[2])), C_NULL, 0), nothing)
    elseif ev isa OrthotropicMicrostructureModel                                                                            # TB_COEF_SPECTRAL_FIELD: per cell, per basis f, s, n
        f, s, n = ev.fiber_coefficient.elementwise_data.data, ev.sheetlet_coefficient.elementwise_data.data, ev.normal_coefficient.elementwise_data.data
        data = Float64[x for k in eachindex(f) for vec3 in (f[k], s[k], n[k]) for x in v3(vec3)]                            # cell-major (ElementwiseData, collections.jl:195-233)
        return (TbCoef(4, 0, 1.0, 1.0, pad16(λ), pointer(data), length(data)), data)
    end
    error("spectral coefficient over $(typeof(ev)) is not lowered")
end
function lower(c::ConductivityToDiffusivityCoefficient)                                                                     # κ/(Cₘχ), coefficients.jl:152-162
    k, keep = lower(c.conductivity_tensor_coefficient)
    return (TbCoef(k.kind, 1, c.capacitance_coefficient.val, c.χ_coefficient.val, k.p, k.field, k.field_len), keep)
end

# ---------------------------------------------------------------- operators (src/solver/interface.jl:17-94, euler.jl:143-176)
mutable struct HIPBilinearOperator{Tv}
    form::Ptr{Cvoid}
    strategy::Cint
    A::HIPSparseMatrixCSR{Tv}            # op.A (euler.jl:112-114)
    is_mass::Bool
end
function setup_operator(strategy::HIPStrategy, integrator::Union{BilinearMassIntegrator, BilinearDiffusionIntegrator}, dh::DofHandler)
    ddh = DeviceDofHandler(strategy.device, dh)
    coef, keep = lower(integrator isa BilinearMassIntegrator ? integrator.ρ : integrator.D)
    form = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve keep check(ccall((:tb_form_create, libtbhip), Cint, (Ptr{Cvoid}, Cint, Cint, Ref{TbCoef}, Ref{Ptr{Cvoid}}),
        ddh.mesh, integrator isa BilinearMassIntegrator ? 0 : 1, 0, Ref(coef), form))
    Tv = value_type(strategy.device)
    op = HIPBilinearOperator{Tv}(form[], strategy_code(strategy), HIPSparseMatrixCSR{Tv}(ddh, HIPVector{Tv}(ddh.dev, ddh.nnz)), integrator isa BilinearMassIntegrator)
    finalizer(o -> ccall((:tb_form_destroy, libtbhip), Cint, (Ptr{Cvoid},), o.form), op)
    return op
end
# the 4-argument form the solvers call (interface.jl:66-94): the solver's system_matrix_type is this backend's CSR type already
setup_operator(strategy::HIPStrategy, integrator::Union{BilinearMassIntegrator, BilinearDiffusionIntegrator}, ::AbstractSolver, dh::DofHandler) =
    setup_operator(strategy, integrator, dh)

update_operator!(op::HIPBilinearOperator, t) =
    check(ccall((:tb_assemble_matrix, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Ptr{Float64}),
        op.form, op.A.ddh.pattern, op.strategy, t, op.A.nzval.ptr))
# update_operator!(mass, t); update_operator!(diffusion, t) of the heat stage's set-up (euler.jl:172-176) in one pass over the mesh
function update_operators!(M::HIPBilinearOperator, K::HIPBilinearOperator, t)
    (M.is_mass && !K.is_mass && M.A.ddh === K.A.ddh) || (update_operator!(M, t); update_operator!(K, t); return nothing)
    check(ccall((:tb_assemble_matrix_pair, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Ptr{Float64}, Ptr{Float64}),
        M.form, K.form, M.A.ddh.pattern, M.strategy, t, M.A.nzval.ptr, K.A.nzval.ptr))
end
mul!(y::HIPVector, op::HIPBilinearOperator, x::HIPVector, α::Number = true, β::Number = false) = mul!(y, op.A, x, α, β)   # mul!(b, M, uₙ₋₁), euler.jl:85

# linear operators: the source closure f(x, t) is evaluated on the host at every quadrature point and uploaded (TB_SRC_TABULATED)
mutable struct HIPLinearOperator{Tv, F, IV}
    form::Ptr{Cvoid}
    strategy::Cint
    b::HIPVector{Tv}                     # op.b (test/gpu/test_operators.jl:30)
    f::F                                 # (x, t) -> value
    nonzero_intervals::IV
    xq::Vector{Vec{3, Float64}}          # quadrature-point coordinates, [q + nq·(cell − 1)]
end
function setup_operator(strategy::HIPStrategy, integrator::LinearIntegrator, dh::DofHandler)
    ddh = DeviceDofHandler(strategy.device, dh)
    protocol = integrator.integrand                      # AnalyticalTransmembraneStimulationProtocol(f, nonzero_intervals), electrophysiology.jl:260-283
    sdh = only(dh.subdofhandlers)
    ip = Ferrite.getfieldinterpolation(sdh, first(sdh.field_names))
    qr = QuadratureRule{Ferrite.getrefshape(ip)}(max(2 * Ferrite.getorder(ip) - 1, 2))                 # fem.jl:52-55
    gip = Ferrite.geometric_interpolation(typeof(Ferrite.get_grid(dh).cells[1]))
    xq = Vec{3, Float64}[]
    for cell in CellIterator(sdh)
        X = getcoordinates(cell)
        for ξ in Ferrite.getpoints(qr)
            x = sum(Ferrite.reference_shape_value(gip, ξ, a) * X[a] for a in 1:length(X))
            push!(xq, Vec{3, Float64}(i -> i <= length(x) ? x[i] : 0.0))
        end
    end
    coef = TbCoef(3, 0, 1.0, 1.0, pad16(()), C_NULL, 0)                                                # TB_SRC_TABULATED
    form = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:tb_form_create, libtbhip), Cint, (Ptr{Cvoid}, Cint, Cint, Ref{TbCoef}, Ref{Ptr{Cvoid}}), ddh.mesh, 2, 0, Ref(coef), form))
    Tv = value_type(strategy.device)
    op = HIPLinearOperator(form[], strategy_code(strategy), HIPVector{Tv}(ddh.dev, ndofs(dh)), protocol.f.f, protocol.nonzero_intervals, xq)
    finalizer(o -> ccall((:tb_form_destroy, libtbhip), Cint, (Ptr{Cvoid},), o.form), op)
    return op
end
setup_operator(strategy::HIPStrategy, integrator::LinearIntegrator, ::AbstractSolver, dh::DofHandler) = setup_operator(strategy, integrator, dh)
needs_update(op::HIPLinearOperator, t) = any(iv -> iv[1] <= t <= iv[2], op.nonzero_intervals)          # src/discretization/operator.jl:17-26
function update_operator!(op::HIPLinearOperator, t)
    table = Float64[op.f(x, t) for x in op.xq]
    check(ccall((:tb_form_set_table, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64), op.form, table, length(table)))
    check(ccall((:tb_assemble_vector, libtbhip), Cint, (Ptr{Cvoid}, Cint, Float64, Ptr{Float64}), op.form, op.strategy, t, op.b.ptr))
end
function add!(b::HIPVector{Tv}, op::HIPLinearOperator{Tv}) where {Tv}                                   # add!(b, source), euler.jl:90
    check(ccall((:tb_axpy, libtbhip), Cint, (Ptr{Cvoid}, Int64, Cdouble, Ptr{Tv}, Ptr{Tv}), b.dev.handle, b.n, 1.0, op.b.ptr, b.ptr))
    return b
end

# ---------------------------------------------------------------- heat stage algebra and solve (euler.jl:71-116)
# _implicit_euler_heat_solver_update_system_matrix!: Anz = Mnz − Δt·Knz
function Thunderbolt._implicit_euler_heat_solver_update_system_matrix!(A::HIPSparseMatrixCSR, M::HIPBilinearOperator, K::HIPBilinearOperator, Δt)
    check(ccall((:tb_heat_matrix, libtbhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Cdouble, Ptr{Float64}),
        A.ddh.dev.handle, A.ddh.nnz, M.A.nzval.ptr, K.A.nzval.ptr, Δt, A.nzval.ptr))
end
# the linear solve of the stage for device matrices: Jacobi-CG inside the library (KrylovJL_CG(atol, rtol) in the tutorials); converged iff the
# residual met the solver's own stopping threshold
function solve_heat_system!(u::HIPVector, A::HIPSparseMatrixCSR, b::HIPVector; rtol = 1e-5, atol = 1e-6, maxiter = 1000, reuse_diagonal = false)
    iters = Ref{Cint}(0); res = Ref{Cdouble}(0.0); tol = Ref{Cdouble}(0.0)
    check(ccall((:tb_cg_solve, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble, Cint, Cint, Ref{Cint}, Ref{Cdouble}),
        A.ddh.pattern, A.nzval.ptr, b.ptr, u.ptr, rtol, atol, maxiter, reuse_diagonal ? 2 : 1, iters, res))
    check(ccall((:tb_solver_last_tolerance, libtbhip), Cint, (Ptr{Cvoid}, Ref{Cdouble}), A.ddh.pattern, tol))
    return res[] <= tol[], Int(iters[])
end

# ---------------------------------------------------------------- pointwise reaction step, dispatched on the vector type
# (src/solver/time/partitioned_solver.jl:38-44, ext/CuThunderboltExt.jl:111-117)
model_id(::Thunderbolt.ParametrizedFHNModel) = Cint(0)
model_id(::Thunderbolt.ParametrizedAlievPanfilovModel) = Cint(1)
model_id(::Thunderbolt.ParametrizedPCG2019Model) = Cint(2)
params(m) = Float64[getfield(m, f) for f in fieldnames(typeof(m))]      # struct field order == ABI parameter order

function _pointwise_step_outer_kernel!(f::AbstractPointwiseFunction, t::Real, Δt::Real, cache::AbstractPointwiseSolverCache, u::HIPVector)
    p = params(f.ode)
    substeps = hasproperty(cache, :substeps) ? cache.substeps : 1
    threshold = hasproperty(cache, :reaction_threshold) ? cache.reaction_threshold : 0.0
    du = cache.du === nothing ? Ptr{Float64}(C_NULL) : cache.du.ptr
    xs = cache.xs                                                        # Vector{Vec{sdim, Float32}} uploaded as an HIPVector{Float32}, or nothing
    if xs === nothing
        rc = ccall((:tb_reaction_step, libtbhip), Cint,
            (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Int64, Cint, Cint, Float64, Float64, Cint, Float64),
            u.dev.handle, model_id(f.ode), p, length(p), cache.uₙ.ptr, du, f.npoints, num_states(f.ode),
            0 #= StateBlockedLayout, fem.jl:385-408 =#, t, Δt, substeps, threshold)
    else
        rc = ccall((:tb_reaction_step_x, libtbhip), Cint,
            (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Int64, Cint, Cint, Ptr{Float32}, Cint, Float64, Float64, Cint, Float64),
            u.dev.handle, model_id(f.ode), p, length(p), cache.uₙ.ptr, du, f.npoints, num_states(f.ode), 0, xs.ptr, xs.n ÷ f.npoints, t, Δt, substeps, threshold)
    end
    return rc == 0
end

# reaction tangent of the RTC controller: `maximum(@view cache.dumat[:, φₘidx])` (src/solver/time/rtc.jl:64-73) on the device
function reaction_tangent(dev::MI355XDevice, dumat_phi::Ptr{Float64}, npoints::Integer, stride::Integer = 1)
    R = Ref{Float64}()
    check(ccall((:tb_max, libtbhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Int64, Ref{Float64}), dev.handle, npoints, dumat_phi, stride, R))
    return R[]
end

# ---------------------------------------------------------------- quasi-static mechanics (src/solver/nonlinear/newton_raphson.jl:234-238)
struct TbMaterial
    kind::Int32; penalty::Int32; p::NTuple{16, Float64}
    f::NTuple{3, Float64}; s::NTuple{3, Float64}; n::NTuple{3, Float64}
    fsn_field::Ptr{Float64}; fsn_field_len::Int64
end
energy_id(::HolzapfelOgden2009Model) = Int32(0); energy_id(::Thunderbolt.NullEnergyModel) = Int32(1)
energy_id(::Thunderbolt.BioNeoHookean) = Int32(2); energy_id(::Thunderbolt.TransverseIsotopicNeoHookeanModel) = Int32(3)
energy_id(::Thunderbolt.LinYinPassiveModel) = Int32(4); energy_id(::Thunderbolt.LinYinActiveModel) = Int32(5)
energy_id(::Thunderbolt.HumphreyStrumpfYinModel) = Int32(6); energy_id(::Thunderbolt.LinearSpringModel) = Int32(7)
energy_id(::Thunderbolt.Guccione1991PassiveModel) = Int32(8)
penalty_id(::Thunderbolt.SimpleCompressionPenalty) = Int32(0); penalty_id(::Thunderbolt.NullCompressionPenalty) = Int32(1)
penalty_id(::Thunderbolt.HartmannNeffCompressionPenalty1) = Int32(2); penalty_id(::Thunderbolt.HartmannNeffCompressionPenalty2) = Int32(3)
penalty_id(::Thunderbolt.HartmannNeffCompressionPenalty3) = Int32(4)
# energy parameters = the Float fields of the energy struct in field order (p[1:9]); penalty parameters (β, a, b) in p[11:13]
function lower(m::PK1Model)
    ψ = m.material
    ep = Float64[getfield(ψ, f) for f in fieldnames(typeof(ψ)) if getfield(ψ, f) isa Real]
    pen = hasproperty(ψ, :mpU) ? ψ.mpU : Thunderbolt.NullCompressionPenalty()
    pp = Float64[getfield(pen, f) for f in fieldnames(typeof(pen)) if getfield(pen, f) isa Real]
    p = zeros(16); p[1:length(ep)] .= ep; p[11:10 + length(pp)] .= pp
    energy_id(ψ) == 0 && penalty_id(pen) == 0 && (p[9] = isempty(pp) ? 1.0 : pp[1])       # HO2009 + SimpleCompressionPenalty: β also in p[9] (fast path)
    ms = m.coefficient isa ConstantCoefficient ? m.coefficient.val : error("mechanics: constant microstructure frames only in this binding")
    return TbMaterial(energy_id(ψ), penalty_id(pen), Tuple(p), v3(ms.f), v3(ms.s), v3(ms.n), C_NULL, 0)
end

mutable struct HIPNonlinearOperator{Tv}
    form::Ptr{Cvoid}
    strategy::Cint
    J::HIPSparseMatrixCSR{Tv}            # op.J (src/solver/interface.jl:9)
end
function setup_operator(strategy::HIPStrategy, model::QuasiStaticModel, ::AbstractSolver, dh::DofHandler)
    ddh = DeviceDofHandler(strategy.device, dh)
    form = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:tb_hyperelastic_create, libtbhip), Cint, (Ptr{Cvoid}, Cint, Ref{TbMaterial}, Ref{Ptr{Cvoid}}), ddh.mesh, 0, Ref(lower(model.material_model)), form))
    Tv = value_type(strategy.device)
    op = HIPNonlinearOperator{Tv}(form[], strategy_code(strategy), HIPSparseMatrixCSR{Tv}(ddh, HIPVector{Tv}(ddh.dev, ddh.nnz)))
    finalizer(o -> ccall((:tb_form_destroy, libtbhip), Cint, (Ptr{Cvoid},), o.form), op)
    return op
end
gettime(p::Real) = p
gettime(p) = p.t                                                         # GenericFirstOrderTimeParameters(p, t, Δt, uprev), euler.jl:490-493
function update_linearization!(op::HIPNonlinearOperator, residual::HIPVector, u::HIPVector, p)
    check(ccall((:tb_linearize, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}),
        op.form, op.J.ddh.pattern, op.strategy, u.ptr, gettime(p), op.J.nzval.ptr, residual.ptr))
end
function update_linearization!(op::HIPNonlinearOperator, u::HIPVector, p)
    check(ccall((:tb_linearize, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}),
        op.form, op.J.ddh.pattern, op.strategy, u.ptr, gettime(p), op.J.nzval.ptr, C_NULL))
end
function residual!(op::HIPNonlinearOperator, residual::HIPVector, u::HIPVector, p)
    check(ccall((:tb_residual, libtbhip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Float64, Ptr{Float64}), op.form, op.strategy, u.ptr, gettime(p), residual.ptr))
end
# apply_zero!(J, r, ch) on the device CSR matrix (src/solver/nonlinear/nlsolve_common.jl:12-26, src/utils.jl:263-278): flags = one byte per dof
function Ferrite.apply_zero!(J::HIPSparseMatrixCSR, r::HIPVector, prescribed_flags::HIPVector{UInt8}, diag::Real)
    check(ccall((:tb_apply_zero_csr, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{UInt8}, Cdouble), J.ddh.pattern, J.nzval.ptr, r.ptr, prescribed_flags.ptr, diag))
end
function LinearAlgebra.dot(a::HIPVector{Float64}, b::HIPVector{Float64})
    r = Ref{Cdouble}(0.0)
    check(ccall((:tb_dot, libtbhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ref{Cdouble}), a.dev.handle, a.n, a.ptr, b.ptr, r))
    return r[]
end
LinearAlgebra.norm(a::HIPVector{Float64}) = sqrt(dot(a, a))

# ---- Float32 value type: MI355XDevice{Float32,Int32} (ext/CuThunderboltExt.jl:126-127 types everything by value_type(device); the reference's GPU
# tests run Float32).  Storage is Float32, the kernels compute in Float64 and round once (include/tbhip.h, "Float32 value type").
function update_operator!(op::HIPBilinearOperator{Float32}, t)
    check(ccall((:tb_assemble_matrix_f32, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Ptr{Float32}),
        op.form, op.A.ddh.pattern, op.strategy, Float64(t), op.A.nzval.ptr))
end
function update_operators!(mass::HIPBilinearOperator{Float32}, diffusion::HIPBilinearOperator{Float32}, t)
    check(ccall((:tb_assemble_matrix_pair_f32, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Ptr{Float32}, Ptr{Float32}),
        mass.form, diffusion.form, mass.A.ddh.pattern, mass.strategy, Float64(t), mass.A.nzval.ptr, diffusion.A.nzval.ptr))
end
function update_operator!(op::HIPLinearOperator{Float32}, t)
    check(ccall((:tb_assemble_vector_f32, libtbhip), Cint, (Ptr{Cvoid}, Cint, Float64, Ptr{Float32}), op.form, op.strategy, Float64(t), op.b.ptr))
end
function LinearAlgebra.mul!(y::HIPVector{Float32}, A::HIPSparseMatrixCSR{Float32}, x::HIPVector{Float32}, α::Number, β::Number)
    check(ccall((:tb_spmv_csr_f32, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Cdouble, Cdouble, Ptr{Float32}),
        A.ddh.pattern, A.nzval.ptr, x.ptr, Float64(α), Float64(β), y.ptr))
    return y
end
function heat_system_matrix!(A::HIPSparseMatrixCSR{Float32}, M::HIPSparseMatrixCSR{Float32}, K::HIPSparseMatrixCSR{Float32}, Δt)
    check(ccall((:tb_heat_matrix_f32, libtbhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float32}, Ptr{Float32}, Cdouble, Ptr{Float32}),
        A.nzval.dev.handle, A.nzval.n, M.nzval.ptr, K.nzval.ptr, Float64(Δt), A.nzval.ptr))
end
function add!(b::HIPVector{Float32}, x::HIPVector{Float32})
    check(ccall((:tb_axpy_f32, libtbhip), Cint, (Ptr{Cvoid}, Int64, Cdouble, Ptr{Float32}, Ptr{Float32}), b.dev.handle, b.n, 1.0, x.ptr, b.ptr))
end
function cg!(x::HIPVector{Float32}, A::HIPSparseMatrixCSR{Float32}, b::HIPVector{Float32}; rtol = 1e-5, atol = 1e-6, maxiter = 1000)
    its = Ref{Cint}(0); res = Ref{Cdouble}(0.0)
    check(ccall((:tb_cg_solve_f32, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Cdouble, Cdouble, Cint, Cint, Ref{Cint}, Ref{Cdouble}),
        A.ddh.pattern, A.nzval.ptr, b.ptr, x.ptr, rtol, atol, maxiter, 1, its, res))
    return its[], res[]
end
# _pointwise_step_outer_kernel! for Float32 solution vectors (partitioned_solver.jl:38-44 dispatches on the vector type)
function pointwise_step_f32!(dev, model_id, p::Vector{Float64}, u::HIPVector{Float32}, du, npoints, nstates, xs, sdim, t, Δt, substeps, threshold)
    check(ccall((:tb_reaction_step_f32, libtbhip), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Ptr{Float32}, Ptr{Float32}, Int64, Cint, Cint, Ptr{Float32}, Cint, Float64, Float64, Cint, Float64),
        dev.handle, model_id, p, length(p), u.ptr, du === nothing ? C_NULL : du.ptr, npoints, nstates, 0, xs === nothing ? C_NULL : xs.ptr, sdim,
        Float64(t), Float64(Δt), substeps, Float64(threshold)))
    return true
end
Base.convert(::Type{HIPVector{Float32}}, v::HIPVector{Float64}) = (w = HIPVector{Float32}(v.dev, v.n);
    check(ccall((:tb_convert_f64_to_f32, libtbhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float32}), v.dev.handle, v.n, v.ptr, w.ptr)); w)
Base.convert(::Type{HIPVector{Float64}}, v::HIPVector{Float32}) = (w = HIPVector{Float64}(v.dev, v.n);
    check(ccall((:tb_convert_f32_to_f64, libtbhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float32}, Ptr{Float64}), v.dev.handle, v.n, v.ptr, w.ptr)); w)

# ---- multi-device path (new work: the reference is shared-memory only, README.md:7).  One Julia process per GPU (MPI.jl); a sub-domain vector holds
# the dofs shared with a neighbouring rank at the positions `idx` (0-based Int32 on the device, both sides in the same order).  The exchange is
# pack → MPI.Isend / MPI.Irecv! on the device buffers (GPU-aware MPI) → unpack; mirrored and tested in thunderbolt.jl_amd/distributed.py (HaloExchange).
struct HaloNeighbour
    peer::Int
    idx::HIPVector{Int32}
    send::HIPVector{Float64}
    recv::HIPVector{Float64}
end
function halo_pack!(nb::HaloNeighbour, v::HIPVector{Float64})
    check(ccall((:tb_gather_indexed, libtbhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}), v.dev.handle, nb.idx.n, v.ptr, nb.idx.ptr, nb.send.ptr))
end
function halo_unpack_add!(v::HIPVector{Float64}, nb::HaloNeighbour)
    check(ccall((:tb_scatter_add_indexed, libtbhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}), v.dev.handle, nb.idx.n, nb.recv.ptr, nb.idx.ptr, v.ptr))
end
# v[idx] = the rows this rank sent (same kernel as the peer's rows): called before halo_unpack_add! in the overlapped product, so that both sides of an
# interface add the same two numbers
function halo_unpack_own!(v::HIPVector{Float64}, nb::HaloNeighbour)
    check(ccall((:tb_scatter_indexed, libtbhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}), v.dev.handle, nb.idx.n, nb.send.ptr, nb.idx.ptr, v.ptr))
end
# interface rows of A·x straight into the send buffer (posted before the whole product is formed, so the transfer overlaps it)
function halo_pack_product_rows!(nb::HaloNeighbour, A::HIPSparseMatrixCSR{Float64}, x::HIPVector{Float64})
    check(ccall((:tb_spmv_csr_rows, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Int32}, Ptr{Float64}),
        A.ddh.pattern, A.nzval.ptr, x.ptr, nb.idx.n, nb.idx.ptr, nb.send.ptr))
end
# work statistics of the patch plan: (patches, cell instances, cells, max instances per patch, max rows per patch, LDS bytes per accumulator block)
function patch_stats(A::HIPSparseMatrixCSR)
    out = zeros(Int64, 6)
    check(ccall((:tb_pattern_patch_stats, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Int64}), A.ddh.pattern, out))
    return out
end
# Sliced mirror of A's values for repeated products (a Krylov solve on a fixed matrix): mul!, mul_dot! and cg! on A then stream the mirrored copy.
# Call again after A's values changed (update_operator!, a new M − Δt·K); unmirror! goes back to the CSR array.  false where the pattern has no mirror.
function mirror!(A::HIPSparseMatrixCSR{Float64})
    rc = ccall((:tb_spmv_mirror, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}), A.ddh.pattern, A.nzval.ptr)
    rc == TB_ERR_UNSUPPORTED && return false
    check(rc)
    return true
end
unmirror!(A::HIPSparseMatrixCSR{Float64}) = check(ccall((:tb_spmv_mirror, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}), A.ddh.pattern, C_NULL))
function diagonal!(d::HIPVector{Float64}, A::HIPSparseMatrixCSR{Float64})
    check(ccall((:tb_extract_diagonal, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), A.ddh.pattern, A.nzval.ptr, d.ptr))
end
# y = A·x and xy[] += xᵀ·y (device scalar): the local quadratic form of the distributed CG
function mul_dot!(y::HIPVector{Float64}, A::HIPSparseMatrixCSR{Float64}, x::HIPVector{Float64}, xy::HIPVector{Float64})
    check(ccall((:tb_spmv_csr_dot, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), A.ddh.pattern, A.nzval.ptr, x.ptr, y.ptr, xy.ptr))
end
# the vector work of one CG iteration with device-resident scalars S = (rz, pAp, rz_new, rr, flag, rr of the last finished iteration) — six doubles; the
# caller all-reduces S[2] and S[3:4] in between and ends the iteration with cgd_rotate! (rz ← rz_new, rr → S[6], accumulators zeroed)
function cgd_dot!(S::HIPVector{Float64}, slot::Int, w::HIPVector{Float64}, a::HIPVector{Float64}, b::HIPVector{Float64})
    check(ccall((:tb_cgd_dot, libtbhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), a.dev.handle, a.n, w.ptr, a.ptr, b.ptr, S.ptr + 8 * slot))
end
function cgd_update!(S::HIPVector{Float64}, w::HIPVector{Float64}, dinv::HIPVector{Float64}, p::HIPVector{Float64}, Ap::HIPVector{Float64}, x::HIPVector{Float64}, r::HIPVector{Float64})
    check(ccall((:tb_cgd_update, libtbhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        x.dev.handle, x.n, w.ptr, dinv.ptr, p.ptr, Ap.ptr, x.ptr, r.ptr, S.ptr, S.ptr + 8, S.ptr + 16))
end
function cgd_direction!(S::HIPVector{Float64}, dinv::HIPVector{Float64}, r::HIPVector{Float64}, p::HIPVector{Float64})
    check(ccall((:tb_cgd_direction, libtbhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), r.dev.handle, r.n, dinv.ptr, r.ptr, p.ptr, S.ptr, S.ptr + 16))
end
# ---------------------------------------------------------------- RCCL behind the ABI: the exchange of the multi-GPU path without GPU-aware MPI on the Julia side.
# Rank 0: id = comm_unique_id(); carry the 128 bytes to the other ranks (MPI.Bcast!, Distributed, a file); every rank: HIPComm(dev, id, rank, nranks).
mutable struct HIPComm
    handle::Ptr{Cvoid}
    dev::MI355XDevice
    function HIPComm(dev::MI355XDevice, id::Vector{UInt8}, rank::Integer, nranks::Integer)
        length(id) == 128 || error("communicator id must be 128 bytes")
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:tb_comm_create, libtbhip), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint, Ref{Ptr{Cvoid}}), dev.handle, id, rank, nranks, h))
        c = new(h[], dev)
        finalizer(x -> ccall((:tb_comm_destroy, libtbhip), Cint, (Ptr{Cvoid},), x.handle), c)
        return c
    end
end
function comm_unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:tb_comm_unique_id, libtbhip), Cint, (Ptr{UInt8},), id))
    return id
end
# sum of interface partials: pack (halo_pack! / halo_pack_product_rows!) → this → halo_unpack_add!
function halo_exchange!(c::HIPComm, nbs::Vector{HaloNeighbour})
    peers = Int32[nb.peer for nb in nbs]; counts = Int64[nb.idx.n for nb in nbs]
    send = Ptr{Float64}[nb.send.ptr for nb in nbs]; recv = Ptr{Float64}[nb.recv.ptr for nb in nbs]
    check(ccall((:tb_comm_exchange, libtbhip), Cint, (Ptr{Cvoid}, Cint, Ptr{Int32}, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}), c.handle, length(nbs), peers, counts, send, recv))
end
# the same exchange on the communicator's own queue: what is launched between begin and end (the whole-domain product) runs beside the transfer
function halo_exchange_begin!(c::HIPComm, nbs::Vector{HaloNeighbour})
    peers = Int32[nb.peer for nb in nbs]; counts = Int64[nb.idx.n for nb in nbs]
    send = Ptr{Float64}[nb.send.ptr for nb in nbs]; recv = Ptr{Float64}[nb.recv.ptr for nb in nbs]
    check(ccall((:tb_comm_exchange_begin, libtbhip), Cint, (Ptr{Cvoid}, Cint, Ptr{Int32}, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}), c.handle, length(nbs), peers, counts, send, recv))
end
halo_exchange_end!(c::HIPComm) = check(ccall((:tb_comm_exchange_end, libtbhip), Cint, (Ptr{Cvoid},), c.handle))
allreduce_sum!(c::HIPComm, S::HIPVector{Float64}, first::Int, n::Int) =
    check(ccall((:tb_comm_allreduce, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Cint), c.handle, S.ptr + 8 * first, n, 0))

# ---------------------------------------------------------------- HIP graphs: a fixed sequence of enqueue-only calls captured once, replayed with one launch
# g = capture_graph(dev) do … the calls of one time step / one CG iteration … end;  launch!(g, t)
mutable struct HIPGraph
    handle::Ptr{Cvoid}
    dev::MI355XDevice
end
function capture_graph(f::Function, dev::MI355XDevice)
    check(ccall((:tb_graph_begin, libtbhip), Cint, (Ptr{Cvoid},), dev.handle))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    try
        f()
    finally
        rc = ccall((:tb_graph_end, libtbhip), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), dev.handle, h)
        check(rc)
    end
    g = HIPGraph(h[], dev)
    finalizer(x -> ccall((:tb_graph_destroy, libtbhip), Cint, (Ptr{Cvoid},), x.handle), g)
    return g
end
launch!(g::HIPGraph, t::Real = 0.0) = check(ccall((:tb_graph_launch, libtbhip), Cint, (Ptr{Cvoid}, Cdouble), g.handle, t))
function node_count(g::HIPGraph)
    n = Ref{Cint}(0)
    check(ccall((:tb_graph_node_count, libtbhip), Cint, (Ptr{Cvoid}, Ref{Cint}), g.handle, n))
    return Int(n[])
end

# one whole iteration of the local CG (a rank without shared dofs): product + pᵀAp, update, direction, rotate in one call
function cgd_iteration!(A::HIPSparseMatrixCSR, dinv::HIPVector{Float64}, x::HIPVector{Float64}, r::HIPVector{Float64}, p::HIPVector{Float64}, Ap::HIPVector{Float64}, S::HIPVector{Float64})
    check(ccall((:tb_cgd_iteration, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        A.ddh.pattern, A.nzval.ptr, dinv.ptr, x.ptr, r.ptr, p.ptr, Ap.ptr, S.ptr))
end
function cgd_rotate!(S::HIPVector{Float64})
    check(ccall((:tb_cgd_rotate, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}), S.dev.handle, S.ptr))
end

# ---------------------------------------------------------------- round 6: the rest of the boundary the hot path's rows of SURVEY §8 use, bound the same way
# (device queue and events; pattern accessors; Rush–Larsen and RTC reaction steps; the other Krylov entries; cell sets and accumulation of a form; facet
# forms; sarcomere steps and the condensed-mechanics setters).  The host generators (tb_host_generate_grid_*, tb_host_close_dofs, tb_host_build_pattern,
# tb_host_perturb_nodes) and the host-side evaluators (tb_host_material_eval*, tb_host_sarcomere_*) stay unbound: Ferrite and the reference's own Julia
# routines are the host side of those.
libtbhip_version() = unsafe_string(ccall((:tb_version, libtbhip), Cstring, ()))
# the library's kernels on a queue the host framework owns (its own stream, or the legacy default stream) — ordering with the host's device work then needs no events
set_stream!(dev::MI355XDevice, hip_stream::Ptr{Cvoid}) = check(ccall((:tb_device_set_stream, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), dev.handle, hip_stream))
use_null_stream!(dev::MI355XDevice) = check(ccall((:tb_device_use_null_stream, libtbhip), Cint, (Ptr{Cvoid},), dev.handle))
synchronize(dev::MI355XDevice) = check(ccall((:tb_device_synchronize, libtbhip), Cint, (Ptr{Cvoid},), dev.handle))
function device_info(dev::MI355XDevice)
    name = Vector{UInt8}(undef, 64); ncu = Ref{Cint}(0); mem = Ref{Csize_t}(0)
    check(ccall((:tb_device_info, libtbhip), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Csize_t, Ref{Cint}, Ref{Csize_t}), dev.handle, name, length(name), ncu, mem))
    return (name = unsafe_string(pointer(name)), n_cu = Int(ncu[]), hbm_bytes = Int(mem[]))
end
mutable struct HIPEvent
    handle::Ptr{Cvoid}
    function HIPEvent(dev::MI355XDevice)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:tb_event_create, libtbhip), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), dev.handle, h))
        e = new(h[])
        finalizer(x -> ccall((:tb_event_destroy, libtbhip), Cint, (Ptr{Cvoid},), x.handle), e)
        return e
    end
end
record!(dev::MI355XDevice, e::HIPEvent) = check(ccall((:tb_event_record, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), dev.handle, e.handle))
function elapsed_ms(a::HIPEvent, b::HIPEvent)
    ms = Ref{Cfloat}(0)
    check(ccall((:tb_event_elapsed_ms, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Cfloat}), a.handle, b.handle, ms))
    return Float64(ms[])
end
mesh_ncells(ddh::DeviceDofHandler) = Int(ccall((:tb_mesh_ncells, libtbhip), Int64, (Ptr{Cvoid},), ddh.mesh))
mesh_ndofs(ddh::DeviceDofHandler) = Int(ccall((:tb_mesh_ndofs, libtbhip), Int64, (Ptr{Cvoid},), ddh.mesh))
pattern_nnz(ddh::DeviceDofHandler) = Int(ccall((:tb_pattern_nnz, libtbhip), Int64, (Ptr{Cvoid},), ddh.pattern))
rowptr_device(ddh::DeviceDofHandler) = ccall((:tb_pattern_rowptr_device, libtbhip), Ptr{Int64}, (Ptr{Cvoid},), ddh.pattern)      # 0-based, on the device
colidx_device(ddh::DeviceDofHandler) = ccall((:tb_pattern_colidx_device, libtbhip), Ptr{Int32}, (Ptr{Cvoid},), ddh.pattern)
function spmv_plan(A::HIPSparseMatrixCSR)   # (row signatures, signature entries): > 0 when the index-compressed product applies
    o = zeros(Int64, 2)
    check(ccall((:tb_pattern_spmv_plan, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Int64}), A.ddh.pattern, o))
    return (o[1], o[2])
end
# cell models: states / parameters / index of the transmembrane potential, defaults (src/modeling/cells/*.jl)
function cell_model_info(model::Integer)
    ns = Ref{Cint}(0); np = Ref{Cint}(0); phi = Ref{Cint}(0)
    check(ccall((:tb_cell_model_info, libtbhip), Cint, (Cint, Ref{Cint}, Ref{Cint}, Ref{Cint}), model, ns, np, phi))
    return (Int(ns[]), Int(np[]), Int(phi[]) + 1)
end
function cell_model_defaults(model::Integer)
    ns, np, _ = cell_model_info(model)
    p = zeros(Float64, np); u0 = zeros(Float64, ns)
    check(ccall((:tb_cell_model_defaults, libtbhip), Cint, (Cint, Ptr{Float64}, Ptr{Float64}), model, p, u0))
    return p, u0
end
# Rush–Larsen step (gates exponentially, the rest forward Euler) and the reaction-tangent-controlled step (src/solver/time/rtc.jl:55-125); states SoA
function reaction_step_rl!(u::HIPVector{Float64}, model::Integer, p::Vector{Float64}, npoints::Integer, nstates::Integer, t::Real, Δt::Real)
    check(ccall((:tb_reaction_step_rl, libtbhip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Int64, Cint, Cint, Cdouble, Cdouble),
        u.dev.handle, model, p, length(p), u.ptr, npoints, nstates, 0, t, Δt))
end
function reaction_step_rtc!(u::HIPVector{Float64}, du, model::Integer, p::Vector{Float64}, npoints::Integer, nstates::Integer, t::Real, Δt::Real, substeps::Integer, threshold::Real)
    rmax = Ref{Cdouble}(0)
    check(ccall((:tb_reaction_step_rtc, libtbhip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Int64, Cint, Cint, Cdouble, Cdouble, Cint, Cdouble, Ref{Cdouble}),
        u.dev.handle, model, p, length(p), u.ptr, du === nothing ? Ptr{Float64}(C_NULL) : du.ptr, npoints, nstates, 0, t, Δt, substeps, threshold, rmax))
    return rmax[]
end
# Krylov entries beside tb_cg_solve (euler.jl:94-100; newton_raphson.jl:61,215-320): CG started from a residual the caller holds, Chebyshev / ℓ₁-Gauss–Seidel
# preconditioned CG (precond 1 / 2), restarted GMRES with a Jacobi preconditioner for the non-symmetric tangents
function cg_from_residual!(x::HIPVector{Float64}, A::HIPSparseMatrixCSR{Float64}, r0::HIPVector{Float64}; rtol = 1e-5, atol = 1e-6, maxiter = 1000, jacobi = true)
    it = Ref{Cint}(0); res = Ref{Cdouble}(0)
    check(ccall((:tb_cg_solve_from_residual, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble, Cint, Cint, Ref{Cint}, Ref{Cdouble}),
        A.ddh.pattern, A.nzval.ptr, r0.ptr, x.ptr, rtol, atol, maxiter, jacobi ? 1 : 0, it, res))
    return Int(it[]), res[]
end
function pcg!(x::HIPVector{Float64}, A::HIPSparseMatrixCSR{Float64}, b::HIPVector{Float64}; rtol = 1e-5, atol = 1e-6, maxiter = 1000, precond = 1, partsize = 64)
    it = Ref{Cint}(0); res = Ref{Cdouble}(0)
    check(ccall((:tb_pcg_solve, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble, Cint, Cint, Cint, Ref{Cint}, Ref{Cdouble}),
        A.ddh.pattern, A.nzval.ptr, b.ptr, x.ptr, rtol, atol, maxiter, precond, partsize, it, res))
    return Int(it[]), res[]
end
function gmres!(x::HIPVector{Float64}, A::HIPSparseMatrixCSR{Float64}, b::HIPVector{Float64}; rtol = 1e-8, atol = 1e-10, maxiter = 1000, restart = 30, jacobi = true)
    it = Ref{Cint}(0); res = Ref{Cdouble}(0)
    check(ccall((:tb_gmres_solve, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble, Cint, Cint, Cint, Ref{Cint}, Ref{Cdouble}),
        A.ddh.pattern, A.nzval.ptr, b.ptr, x.ptr, rtol, atol, maxiter, restart, jacobi ? 1 : 0, it, res))
    return Int(it[]), res[]
end
l1gs_apply!(z::HIPVector{Float64}, A::HIPSparseMatrixCSR{Float64}, r::HIPVector{Float64}; partsize = 64, sweep = 0) =
    check(ccall((:tb_l1gs_apply, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint, Cint, Ptr{Float64}, Ptr{Float64}), A.ddh.pattern, A.nzval.ptr, partsize, sweep, r.ptr, z.ptr))
function absmax(v::HIPVector{Float64}; stride = 1)
    r = Ref{Cdouble}(0)
    check(ccall((:tb_absmax, libtbhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Int64, Ref{Cdouble}), v.dev.handle, v.n ÷ stride, v.ptr, stride, r))
    return r[]
end
function meandiag(A::HIPSparseMatrixCSR{Float64})
    r = Ref{Cdouble}(0)
    check(ccall((:tb_meandiag, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ref{Cdouble}), A.ddh.pattern, A.nzval.ptr, r))
    return r[]
end
# a form restricted to a cell set (OrderedSet of a SubDofHandler) and a form that adds to its output instead of overwriting it
set_cellset!(form::Ptr{Cvoid}, cells::Vector{Int32}) = check(ccall((:tb_form_set_cellset, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Int32}, Int64, Cint), form, cells, length(cells), 1))
clear_cellset!(form::Ptr{Cvoid}) = check(ccall((:tb_form_clear_cellset, libtbhip), Cint, (Ptr{Cvoid},), form))
set_accumulate!(form::Ptr{Cvoid}, on::Bool) = check(ccall((:tb_form_set_accumulate, libtbhip), Cint, (Ptr{Cvoid}, Cint), form, on ? 1 : 0))
# facet terms (src/modeling/solid/weak_boundary_conditions.jl): RobinBC / NormalSpringBC / BendingSpringBC / PressureFieldBC on a facet set
function facet_form(ddh::DeviceDofHandler, bc_kind::Integer, param::Real, facets::Vector{Int32}; facet_qpoints = 0)
    form = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:tb_facet_form_create, libtbhip), Cint, (Ptr{Cvoid}, Cint, Cdouble, Cint, Ptr{Int32}, Int64, Cint, Ref{Ptr{Cvoid}}),
        ddh.mesh, bc_kind, param, facet_qpoints, facets, length(facets) ÷ 2, 1, form))
    return form[]
end
facet_set_param!(form::Ptr{Cvoid}, param::Real) = check(ccall((:tb_facet_form_set_param, libtbhip), Cint, (Ptr{Cvoid}, Cdouble), form, param))
facet_set_field!(form::Ptr{Cvoid}, field::Vector{Float64}) = check(ccall((:tb_facet_form_set_field, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64), form, field, length(field)))
facet_assemble!(form::Ptr{Cvoid}, A::HIPSparseMatrixCSR{Float64}, u::HIPVector{Float64}, r::HIPVector{Float64}, t::Real) =
    check(ccall((:tb_facet_assemble, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Cdouble, Ptr{Float64}, Ptr{Float64}), form, A.ddh.pattern, u.ptr, t, A.nzval.ptr, r.ptr))
# sarcomere models (RDQ20-MF …): explicit and implicit pointwise steps (src/modeling/solid/materials.jl:1403-1640 condenses them per quadrature point)
function sarcomere_model_info(model::Integer)
    ns = Ref{Cint}(0); np = Ref{Cint}(0)
    check(ccall((:tb_sarcomere_model_info, libtbhip), Cint, (Cint, Ref{Cint}, Ref{Cint}), model, ns, np))
    return Int(ns[]), Int(np[])
end
function sarcomere_step!(Q::HIPVector{Float64}, model::Integer, p::Vector{Float64}, npoints::Integer, stretch::Real, velocity::Real, calcium::Real, t::Real, Δt::Real; substeps = 1)
    check(ccall((:tb_sarcomere_step, libtbhip), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Ptr{Float64}, Ptr{Float64}),
        Q.dev.handle, model, p, length(p), Q.ptr, npoints, C_NULL, C_NULL, C_NULL, stretch, velocity, calcium, t, Δt, substeps, 0, C_NULL, C_NULL))
end
function sarcomere_implicit_step!(Q::HIPVector{Float64}, Qknown::HIPVector{Float64}, model::Integer, p::Vector{Float64}, npoints::Integer, stretch::Real, velocity::Real, calcium::Real, Δt::Real;
                                  tol = 1e-10, max_iters = 20)
    nf = Ref{Int64}(0)
    check(ccall((:tb_sarcomere_implicit_step, libtbhip), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Ptr{Float64}, Ptr{Float64},
         Ptr{Int32}, Ref{Int64}),
        Q.dev.handle, model, p, length(p), Q.ptr, Qknown.ptr, npoints, C_NULL, C_NULL, C_NULL, stretch, velocity, calcium, Δt, tol, max_iters, C_NULL, C_NULL, C_NULL, nf))
    return Int(nf[])
end
# condensed internal variables and the other setters of a hyperelastic form (src/modeling/solid/elements.jl:411-630)
set_active_tension!(form::Ptr{Cvoid}, tension::Real) = check(ccall((:tb_hyperelastic_set_active_tension, libtbhip), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Float64}, Int64), form, tension, C_NULL, 0))
set_prestress!(form::Ptr{Cvoid}, F0inv::Vector{Float64}) = check(ccall((:tb_hyperelastic_set_prestress, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}), form, F0inv))
set_condensation!(form::Ptr{Cvoid}, sarcomere_model::Integer, p::Vector{Float64}, tmax::Real; local_tol = 1e-10, local_max_iters = 20) =
    check(ccall((:tb_hyperelastic_set_condensation, libtbhip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Cdouble, Cdouble, Cint), form, sarcomere_model, p, length(p), tmax, local_tol, local_max_iters))
function n_quadrature_points(form::Ptr{Cvoid})
    n = Ref{Int64}(0)
    check(ccall((:tb_hyperelastic_n_quadrature_points, libtbhip), Cint, (Ptr{Cvoid}, Ref{Int64}), form, n))
    return Int(n[])
end
set_internal_state!(form::Ptr{Cvoid}, Q::HIPVector{Float64}, Qknown::HIPVector{Float64}, Δt::Real) =
    check(ccall((:tb_hyperelastic_set_internal_state, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cdouble), form, Q.ptr, Qknown.ptr, Δt))
set_previous_solution!(form::Ptr{Cvoid}, uprev::HIPVector{Float64}) = check(ccall((:tb_hyperelastic_set_previous_solution, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}), form, uprev.ptr))
function local_solve_report(form::Ptr{Cvoid})
    nf = Ref{Int64}(0)
    check(ccall((:tb_hyperelastic_local_solve_report, libtbhip), Cint, (Ptr{Cvoid}, Ref{Int64}, Ptr{Int32}, Int64), form, nf, C_NULL, 0))
    return Int(nf[])
end
function comm_rank_size(c::HIPComm)
    r = Ref{Cint}(0); n = Ref{Cint}(0)
    check(ccall((:tb_comm_rank_size, libtbhip), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}), c.handle, r, n))
    return Int(r[]), Int(n[])
end

end # module