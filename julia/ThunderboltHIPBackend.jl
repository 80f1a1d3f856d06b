# ThunderboltHIPBackend.jl — the reference-side binding a Thunderbolt.jl maintainer would add (as a package
# extension next to ext/CuThunderboltExt.jl) to use libtbhip.so as an `AbstractGPUDevice` backend.
#
# NOT executed in the build container (no Julia toolchain there).  It is kept thin on purpose and mirrors
# thunderbolt.jl_amd/api.py call for call; every `ccall` below targets a symbol declared in include/tbhip.h.
module ThunderboltHIPBackend

using Thunderbolt, Ferrite, SparseArrays, SparseMatricesCSR
import Thunderbolt: AbstractGPUDevice, AbstractAssemblyStrategy, setup_operator, update_operator!,
    _pointwise_step_outer_kernel!, create_system_vector, create_system_matrix, value_type, index_type,
    AbstractPointwiseSolverCache, PointwiseODEFunction, num_states, BilinearMassIntegrator,
    BilinearDiffusionIntegrator, LinearIntegrator

const libtbhip = get(ENV, "TBHIP_LIBRARY", "libtbhip.so")

check(rc::Cint) = rc == 0 ? nothing :
    error("libtbhip error $rc: " * unsafe_string(ccall((:tb_last_error_string, libtbhip), Cstring, ())))

# ---------------------------------------------------------------- device (replaces FerriteOperators.CudaDevice)
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
MI355XDevice(id = 0) = MI355XDevice{Float64, Int32}(id)
value_type(::MI355XDevice{Tv}) where {Tv} = Tv
index_type(::MI355XDevice{Tv, Ti}) where {Tv, Ti} = Ti

# device vector: GC-managed wrapper, finaliser calls tb_free
mutable struct HIPVector{T} <: AbstractVector{T}
    dev::MI355XDevice
    ptr::Ptr{T}
    n::Int
    function HIPVector{T}(dev, n) where {T}
        p = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:tb_malloc, libtbhip), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), dev.handle, n * sizeof(T), p))
        v = new{T}(dev, Ptr{T}(p[]), n)
        finalizer(x -> ccall((:tb_free, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), x.dev.handle, x.ptr), v)
        return v
    end
end
Base.size(v::HIPVector) = (v.n,)
function Base.copyto!(v::HIPVector{T}, a::Vector{T}) where {T}
    check(ccall((:tb_memcpy_h2d, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), v.dev.handle, v.ptr, a, sizeof(a)))
    return v
end
function Base.Vector(v::HIPVector{T}) where {T}
    a = Vector{T}(undef, v.n)
    check(ccall((:tb_memcpy_d2h, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), v.dev.handle, a, v.ptr, sizeof(a)))
    return a
end
# ext/CuThunderboltExt.jl:126-127
create_system_vector(::Type{<:HIPVector{T}}, dev::MI355XDevice, n::Integer) where {T} = HIPVector{T}(dev, n)

# ---------------------------------------------------------------- strategies (src/Thunderbolt.jl:22-32)
struct PatchAssemblyStrategy{D <: MI355XDevice} <: AbstractAssemblyStrategy
    device::D
end
strategy_code(::Thunderbolt.PerColorAssemblyStrategy) = Cint(1)
strategy_code(::Thunderbolt.ElementAssemblyStrategy) = Cint(2)
strategy_code(::PatchAssemblyStrategy) = Cint(3)

# ---------------------------------------------------------------- mesh + dof table: Ferrite's own arrays go over as they are
struct DeviceMesh
    handle::Ptr{Cvoid}
end
function DeviceMesh(dev::MI355XDevice, dh::DofHandler)
    grid = Ferrite.get_grid(dh)
    sdh = only(dh.subdofhandlers)                       # one subdomain, one field (SURVEY §8b)
    sdim = Ferrite.getspatialdim(grid)
    xyz = sdim == 3 ? collect(reinterpret(Float64, [n.x for n in grid.nodes])) :        # AoS, 3 per node
          Float64[i <= 2 ? n.x[i] : 0.0 for n in grid.nodes for i in 1:3]               # 2-D meshes travel with z = 0 (TB_QUAD4)
    conn = Int32[v for c in grid.cells for v in c.nodes]          # 1-based
    ndpc = Ferrite.ndofs_per_cell(sdh)
    celldofs = Int32.(dh.cell_dofs)                               # dh.cell_dofs / cell_dofs_offset, src/utils.jl:52-56
    kind = grid.cells[1] isa Quadrilateral ? Cint(2) : grid.cells[1] isa Hexahedron ? Cint(3) : Cint(4)   # TB_QUAD4 / TB_HEX8 / TB_TET4
    order = Ferrite.getorder(Ferrite.getfieldinterpolation(sdh, first(sdh.field_names)))
    fkind = (kind == 3 && order == 2) ? Cint(5) : kind            # TB_HEX27
    ncomp = Ferrite.n_components(Ferrite.getfieldinterpolation(sdh, first(sdh.field_names)))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:tb_mesh_create, libtbhip), Cint,
        (Ptr{Cvoid}, Cint, Int64, Ptr{Float64}, Int64, Ptr{Int32}, Cint, Cint, Ptr{Int32}, Int64, Cint, Ref{Ptr{Cvoid}}),
        dev.handle, kind, length(grid.nodes), xyz, length(grid.cells), conn, fkind, ncomp, celldofs, ndofs(dh), 1, h))
    return DeviceMesh(h[])
end

# CSR pattern of transpose(allocate_matrix(dh)) (src/solver/interface.jl:162-168)
function device_pattern(mesh::DeviceMesh, A::SparseMatrixCSR{1, Tv, Ti}) where {Tv, Ti}
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:tb_pattern_create, libtbhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int32}, Cint, Ref{Ptr{Cvoid}}),
        mesh.handle, size(A, 1), Int64.(A.rowptr), Int32.(A.colval), 1, h))
    return h[]
end

# ---------------------------------------------------------------- coefficients → tb_coef (closures cannot cross the ABI, SURVEY F10)
struct TbCoef
    kind::Int32; wrap::Int32; Cm::Float64; chi::Float64
    p::NTuple{16, Float64}; field::Ptr{Float64}; field_len::Int64
end
pad16(v) = ntuple(i -> i <= length(v) ? Float64(v[i]) : 0.0, 16)
lower(c::ConstantCoefficient{<:Real}) = TbCoef(0, 0, 1.0, 1.0, pad16((c.val,)), C_NULL, 0)
lower(c::ConstantCoefficient{<:Tensors.AbstractTensor{2, 3}}) = TbCoef(1, 0, 1.0, 1.0, pad16(vec(Matrix(c.val)')), C_NULL, 0)
function lower(c::Thunderbolt.ConductivityToDiffusivityCoefficient)       # κ/(Cₘχ), coefficients.jl:152-162
    k = lower(c.conductivity_tensor_coefficient)
    return TbCoef(k.kind, 1, c.capacitance_coefficient.val, c.χ_coefficient.val, k.p, k.field, k.field_len)
end
# SpectralTensorCoefficient of constant / nodal f,s,n: kinds 3, 4, 5 — same lowering as thunderbolt.jl_amd/api.py:_lower_coef

# ---------------------------------------------------------------- operators (src/solver/interface.jl:17-94, euler.jl:172-176)
mutable struct HIPBilinearOperator{Tv}
    form::Ptr{Cvoid}; pattern::Ptr{Cvoid}; strategy::Cint
    A::HIPVector{Tv}                    # nzval of op.A; rowptr/colval via tb_pattern_{rowptr,colidx}_device
end
function setup_operator(strategy::Union{PatchAssemblyStrategy, Thunderbolt.PerColorAssemblyStrategy{<:MI355XDevice}},
        integrator::Union{BilinearMassIntegrator, BilinearDiffusionIntegrator}, dh::DofHandler)
    dev = strategy.device
    mesh = DeviceMesh(dev, dh)
    A = SparseMatrixCSR(transpose(allocate_matrix(dh)))
    pat = device_pattern(mesh, A)
    coef = Ref(lower(integrator isa BilinearMassIntegrator ? integrator.ρ : integrator.D))
    form = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:tb_form_create, libtbhip), Cint, (Ptr{Cvoid}, Cint, Cint, Ref{TbCoef}, Ref{Ptr{Cvoid}}),
        mesh.handle, integrator isa BilinearMassIntegrator ? 0 : 1, 0, coef, form))
    return HIPBilinearOperator{Float64}(form[], pat, strategy_code(strategy), HIPVector{Float64}(dev, nnz(A)))
end
update_operator!(op::HIPBilinearOperator, t) =
    check(ccall((:tb_assemble_matrix, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Ptr{Float64}),
        op.form, op.pattern, op.strategy, t, op.A.ptr))
# linear operators: tb_form_create(kind = 2, TB_SRC_* id or TB_SRC_TABULATED + tb_form_set_table(f.(x_q, t))) and
# update_operator!(op, t) = tb_assemble_vector(form, strategy, t, op.b.ptr); needs_update stays host-side
# (src/discretization/operator.jl:17-26).

# ---------------------------------------------------------------- pointwise reaction step, dispatched on the vector type
# (src/solver/time/partitioned_solver.jl:38-44, ext/CuThunderboltExt.jl:111-117)
model_id(::Thunderbolt.ParametrizedFHNModel) = Cint(0)
model_id(::Thunderbolt.ParametrizedAlievPanfilovModel) = Cint(1)
model_id(::Thunderbolt.ParametrizedPCG2019Model) = Cint(2)
params(m) = Float64[getfield(m, f) for f in fieldnames(typeof(m))]      # struct field order == ABI parameter order

function _pointwise_step_outer_kernel!(f::PointwiseODEFunction, t::Real, Δt::Real,
        cache::AbstractPointwiseSolverCache, u::HIPVector)
    p = params(f.ode)
    npoints = length(f.associated_states) ÷ num_states(f.ode)
    substeps = hasproperty(cache, :substeps) ? cache.substeps : 1
    threshold = hasproperty(cache, :reaction_threshold) ? cache.reaction_threshold : 0.0
    rc = ccall((:tb_reaction_step, libtbhip), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Int64, Cint, Cint, Float64, Float64, Cint, Float64),
        u.dev.handle, model_id(f.ode), p, length(p), cache.uₙ.ptr, cache.du.ptr, npoints, num_states(f.ode),
        0 #= StateBlockedLayout, fem.jl:385-408 =#, t, Δt, substeps, threshold)
    return rc == 0
end

# reaction tangent of the RTC controller: `maximum(@view cache.dumat[:, φₘidx])` (src/solver/time/rtc.jl:64-73) on the device
function reaction_tangent(dev::MI355XDevice, dumat_phi::Ptr{Float64}, npoints::Integer, stride::Integer = 1)
    R = Ref{Float64}()
    check(ccall((:tb_max, libtbhip), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Int64, Ref{Float64}), dev.handle, npoints, dumat_phi, stride, R))
    return R[]
end
# … or fused into the step itself (no dumat needed): tb_reaction_step_rtc(..., rmax::Ref{Float64})


# ---------------------------------------------------------------- quasi-static mechanics (src/solver/nonlinear/newton_raphson.jl:234-238)
struct TbMaterial
    kind::Int32; penalty::Int32; p::NTuple{16, Float64}
    f::NTuple{3, Float64}; s::NTuple{3, Float64}; n::NTuple{3, Float64}
    fsn_field::Ptr{Float64}; fsn_field_len::Int64
end
struct TbHill
    framework::Int32; active_energy::Int32; active_penalty::Int32; adg_kind::Int32; sarcomere_kind::Int32
    active_p::NTuple{12, Float64}; sheetlet_part::Float64; sarcomere_p::NTuple{2, Float64}
end
energy_id(::HolzapfelOgden2009Model) = Int32(0); energy_id(::Thunderbolt.NullEnergyModel) = Int32(1)
energy_id(::Thunderbolt.BioNeoHookean) = Int32(2); energy_id(::Thunderbolt.TransverseIsotopicNeoHookeanModel) = Int32(3)
energy_id(::Thunderbolt.LinYinPassiveModel) = Int32(4); energy_id(::Thunderbolt.LinYinActiveModel) = Int32(5)
energy_id(::Thunderbolt.HumphreyStrumpfYinModel) = Int32(6); energy_id(::Thunderbolt.LinearSpringModel) = Int32(7)
energy_id(::Thunderbolt.Guccione1991PassiveModel) = Int32(8)
penalty_id(::Thunderbolt.SimpleCompressionPenalty) = Int32(0); penalty_id(::Thunderbolt.NullCompressionPenalty) = Int32(1)
penalty_id(::Thunderbolt.HartmannNeffCompressionPenalty1) = Int32(2); penalty_id(::Thunderbolt.HartmannNeffCompressionPenalty2) = Int32(3)
penalty_id(::Thunderbolt.HartmannNeffCompressionPenalty3) = Int32(4)
# energy parameters = the struct fields before mpU, penalty parameters (β, a, b) in p[11:13]; constant frame or nodal f/s/n field

mutable struct HIPNonlinearOperator{Tv}
    form::Ptr{Cvoid}; facet_forms::Vector{Ptr{Cvoid}}; pattern::Ptr{Cvoid}; strategy::Cint
    J::HIPVector{Tv}
    Q::Union{Nothing, HIPVector{Tv}}; Qknown::Union{Nothing, HIPVector{Tv}}     # condensed internal variables, n_states × (n_cells·n_qp)
end
# setup: tb_hyperelastic_create(mesh, 0, Ref(material), form); per weak boundary condition tb_facet_form_create(…);
#   GeneralizedHillModel / ExtendedHillModel          → tb_hyperelastic_set_hill(form, Ref(TbHill(…)))
#   PrestressedMechanicalModel(inner, ConstantCoeff.) → tb_hyperelastic_set_prestress(form, vec(Matrix(F₀⁻¹)'))
#   Dict(cellset => QuasiStaticModel)                  → one form each: tb_form_set_cellset(form, cells, n, 1); tb_form_set_accumulate(form, k > 1)
#   internal sarcomere model (RDQ20MFModel)            → tb_hyperelastic_set_condensation(form, 2, params, 17, Tmax, tol, max_iters)
function Thunderbolt.update_linearization!(op::HIPNonlinearOperator, residual::HIPVector, u::HIPVector, p)
    t = p isa Real ? p : p.t
    if p isa Thunderbolt.FerriteOperators.GenericFirstOrderTimeParameters && op.Q !== nothing
        # GenericFirstOrderTimeParameters(p, t, Δt, uprev) (euler.jl:490-493): Δt and the known state of the local problems; the
        # rate-coupled model additionally reads uprev: tb_hyperelastic_set_previous_solution(op.form, p.uprev.ptr)
        check(ccall((:tb_hyperelastic_set_internal_state, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64),
            op.form, op.Q.ptr, op.Qknown.ptr, p.Δt))
    end
    check(ccall((:tb_linearize, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}),
        op.form, op.pattern, op.strategy, u.ptr, t, op.J.ptr, residual.ptr))
    for h in op.facet_forms
        check(ccall((:tb_facet_assemble, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}),
            h, op.pattern, u.ptr, t, op.J.ptr, residual.ptr))
    end
end
function Thunderbolt.residual!(op::HIPNonlinearOperator, residual::HIPVector, u::HIPVector, p)
    t = p isa Real ? p : p.t
    check(ccall((:tb_residual, libtbhip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Float64, Ptr{Float64}), op.form, op.strategy, u.ptr, t, residual.ptr))
    for h in op.facet_forms
        check(ccall((:tb_facet_assemble, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}),
            h, C_NULL, u.ptr, t, C_NULL, residual.ptr))
    end
end
# check_local_solve_convergence → tb_hyperelastic_local_solve_report(form, n_failed, C_NULL, 0); Newton's linear solve:
# tb_cg_solve (SPD tangents) or tb_gmres_solve (the reference's default KrylovJL_GMRES; required for the rate-coupled model);
# apply_zero!(J, r, ch) → tb_apply_zero_csr; norm(r[free]) → tb_dot.

# sarcomere_rhs! under a pointwise explicit step (StandaloneSarcomereModel): tb_sarcomere_step; the local problem alone:
# tb_sarcomere_implicit_step

# heat-step algebra (src/solver/time/euler.jl:85,90,110-116): tb_heat_matrix, tb_spmv_csr, tb_axpy
# perform_backward_euler_step! (euler.jl:71-101) for device vectors: the initial guess is uₙ₋₁, so b − A·uₙ₋₁ = Δt·K·uₙ₋₁ (+ source) and CG can
# start from that residual — one SpMV with K instead of mul!(b, M, uₙ₋₁) and the solver's own A·x₀:
function backward_euler_heat_step!(pat, Anz::Ptr{Float64}, Knz::Ptr{Float64}, u::Ptr{Float64}, r0::Ptr{Float64}, Δt; rtol = 1e-5, atol = 1e-6, maxiter = 1000)
    check(ccall((:tb_spmv_csr, libtbhip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble, Ptr{Float64}), pat, Knz, u, Δt, 0.0, r0))
    iters = Ref{Cint}(0); res = Ref{Cdouble}(0.0)
    check(ccall((:tb_cg_solve_from_residual, libtbhip), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble, Cint, Cint, Ref{Cint}, Ref{Cdouble}),
                pat, Anz, r0, u, rtol, atol, maxiter, 1, iters, res))
    return iters[] < maxiter || res[] <= atol
end
end # module
