"""Mechanics side of the host mirror — everything above the C ABI that is NOT on the hot path SURVEY §8 scopes (rows a1–a10): the material and
boundary-condition classes of the quasi-static mechanics problem, the nonlinear operator wrapper, Dirichlet constraints, the Newton–Raphson and
load-path (homotopy) solvers with the reference's continuation controllers, and the sarcomere models with internal state.  It was written in round 1
inside api.py; it lives here so that api.py reads as what the path needs (device, mesh, coefficients, integrators / operators, reaction, heat step).
Behaviour and names are unchanged: api.py re-exports this module, `import thunderbolt_jl_amd as tb; tb.NewtonRaphsonSolver` works as before.

Reference files mirrored: src/modeling/solid/{materials,energies,active,contraction,elements}.jl, src/modeling/core/weak_boundary_conditions.jl,
src/solver/nonlinear/{newton_raphson,nlsolve_common}.jl, src/solver/load_stepping/homotopy.jl (class docstrings carry the line numbers)."""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import TBError, check, lib
from .api import (ConstantCoefficient, DeviceVector, ElementAssemblyStrategy, OrthotropicMicrostructure, OrthotropicMicrostructureModel, _ptr,
                  cg_solve, gmres_solve, num_states, pcg_solve, solve_converged)


# --------------------------------------------------------------------------------------- quasi-static mechanics
class HolzapfelOgden2009Model:
    """HolzapfelOgden2009Model(; a, b, aᶠ, bᶠ, aˢ, bˢ, aᶠˢ, bᶠˢ, mpU = SimpleCompressionPenalty(β)) (energies.jl:136-168)."""
    names = ["a", "b", "af", "bf", "as_", "bs", "afs", "bfs", "beta"]

    def __init__(self, a=0.059, b=8.023, af=18.472, bf=16.026, as_=2.581, bs=11.120, afs=0.216, bfs=11.436, beta=1.0, mpU=None):
        if mpU is not None and mpU.pid == 0:
            beta = mpU.u[0]
        self.p = np.array([a, b, af, bf, as_, bs, afs, bfs, beta], dtype=np.float64)
        self.mpU = mpU                       # None / SimpleCompressionPenalty → hand-derived routines; other penalties → device AD


# ---- compression penalties U(I₃) and passive energies of src/modeling/solid/energies.jl (defaults = the reference's @kwdef defaults)
class SimpleCompressionPenalty:
    pid = 0

    def __init__(self, beta=1.0):
        self.u = [beta, 0.0, 0.0]


class NullCompressionPenalty:
    pid = 1
    u = [0.0, 0.0, 0.0]


class HartmannNeffCompressionPenalty1:
    pid = 2

    def __init__(self, a=1, b=2, beta=1.0):
        self.u = [beta, float(a), float(b)]


class HartmannNeffCompressionPenalty2:
    pid = 3

    def __init__(self, a=1.1, beta=1.0):
        self.u = [beta, float(a), 0.0]


class HartmannNeffCompressionPenalty3:
    pid = 4

    def __init__(self, beta=1.0):
        self.u = [beta, 0.0, 0.0]


class _Energy:
    kind = 0

    def __init__(self, p, mpU):
        self.p = np.array(list(p), dtype=np.float64)
        self.mpU = mpU


class NullEnergyModel(_Energy):
    kind = 1

    def __init__(self):
        super().__init__([], NullCompressionPenalty())


class BioNeoHookean(_Energy):
    kind = 2

    def __init__(self, alpha=1.0, mpU=None):
        super().__init__([alpha], mpU or SimpleCompressionPenalty())


class TransverseIsotopicNeoHookeanModel(_Energy):
    kind = 3

    def __init__(self, a1=2.6, a2=2.82, alpha1=30.48, alpha2=7.25, mpU=None):
        super().__init__([a1, a2, alpha1, alpha2], mpU or HartmannNeffCompressionPenalty1())


class LinYinPassiveModel(_Energy):
    kind = 4

    def __init__(self, C1=1.05, C2=9.13, C3=2.32, C4=0.08, mpU=None):
        super().__init__([C1, C2, C3, C4], mpU or SimpleCompressionPenalty())


class LinYinActiveModel(_Energy):
    kind = 5

    def __init__(self, C0=0.0, C1=-13.03, C2=36.65, C3=35.42, C4=15.52, C5=1.62, mpU=None):
        super().__init__([C0, C1, C2, C3, C4, C5], mpU or SimpleCompressionPenalty())


class HumphreyStrumpfYinModel(_Energy):
    kind = 6

    def __init__(self, C1=15.93, C2=55.85, C3=3.59, C4=30.21, mpU=None):
        super().__init__([C1, C2, C3, C4], mpU or SimpleCompressionPenalty())


class LinearSpringModel(_Energy):
    kind = 7

    def __init__(self, eta=10.0, mpU=None):
        super().__init__([eta], mpU or NullCompressionPenalty())


class Guccione1991PassiveModel(_Energy):
    kind = 8

    def __init__(self, C0=0.1, Bff=29.8, Bss=14.9, Bnn=14.9, Bns=9.3, Bfs=19.2, Bfn=14.4, mpU=None):
        super().__init__([C0, Bff, Bss, Bnn, Bns, Bfs, Bfn], mpU or SimpleCompressionPenalty(50.0))


class PK1Model:
    """PK1Model(material, coefficient_field) with a constant OrthotropicMicrostructure (materials.jl:442-453)."""

    def __init__(self, material, microstructure):
        ms = microstructure.val if isinstance(microstructure, ConstantCoefficient) else microstructure
        if not isinstance(ms, (OrthotropicMicrostructure, OrthotropicMicrostructureModel)):
            raise TypeError("PK1Model: OrthotropicMicrostructure (constant) or OrthotropicMicrostructureModel (nodal fields) expected")
        self.material, self.microstructure = material, ms

    def lower(self):
        m = L.tb_material()
        mat = self.material
        if isinstance(mat, _Energy):                  # any energy + any penalty: differentiated on the device
            m.kind, m.reserved = mat.kind, mat.mpU.pid
            for i, v in enumerate(mat.p):
                m.p[i] = v
            for i, v in enumerate(mat.mpU.u):
                m.p[10 + i] = v
        else:                                         # HolzapfelOgden2009Model(…, mpU = SimpleCompressionPenalty(β)): hand-derived fast path
            pen = getattr(mat, "mpU", None)
            m.kind = L.TB_MATERIAL_HOLZAPFEL_OGDEN_2009
            for i, v in enumerate(mat.p):
                m.p[i] = v
            if pen is not None and pen.pid != 0:
                m.reserved = pen.pid
                for i, v in enumerate(pen.u):
                    m.p[10 + i] = v
            else:
                m.p[10] = mat.p[8]
        if isinstance(self.microstructure, OrthotropicMicrostructureModel):
            self._keep = self.microstructure.fsn           # [cell][node][f|s|n][3]
            m.fsn_field = self._keep.ctypes.data_as(L.c_dp)
            m.fsn_field_len = self._keep.size
            m.f[0], m.s[1], m.n[2] = 1.0, 1.0, 1.0
        else:
            for i in range(3):
                m.f[i], m.s[i], m.n[i] = self.microstructure.f[i], self.microstructure.s[i], self.microstructure.n[i]
        return m


class PrestressedMechanicalModel:
    """PrestressedMechanicalModel(inner_model, prestress_field) (materials.jl:781-900): P(F) = Pᵉ(F·F₀⁻¹)·F₀⁻ᵀ; prestress_field:
    ConstantCoefficient of the 3×3 tensor F₀⁻¹ (a numpy array M with M[i, j] = F₀⁻¹_ij — Tensors.jl's Tensor{2,3}((…)) constructor
    lists the entries column by column)."""

    def __init__(self, inner_model, prestress_field):
        self.inner_model = inner_model
        G = prestress_field.val if isinstance(prestress_field, ConstantCoefficient) else prestress_field
        self.F0inv = np.ascontiguousarray(G, dtype=np.float64).reshape(3, 3)
        self.material, self.microstructure = inner_model.material, inner_model.microstructure

    def lower(self, *a):
        return self.inner_model.lower(*a)


class SimpleActiveStress:
    """SimpleActiveStress(; Tmax): Tᵃ = Tmax·[Caᵢ]·(F·f₀)⊗f₀/‖F·f₀‖ (src/modeling/solid/active.jl:100-113)."""

    def __init__(self, Tmax=1.0):
        self.Tmax = float(Tmax)


class PelceSunLangeveld1995Model:
    """PelceSunLangeveld1995Model(; β, λᵃₘₐₓ): steady-state sarcomere model (contraction.jl:302-311); in the active
    *stress* framework 𝓝(state, …) = state for every steady-state model (contraction.jl:103-105)."""

    def __init__(self, beta=3.0, lambda_a_max=0.7):
        self.beta, self.lambda_a_max = beta, lambda_a_max

    def compute_lambda_a(self, Ca):
        f = 0.5 + np.arctan(self.beta * np.log(Ca)) / np.pi if Ca > 0.0 else 0.0
        return 1.0 / (1.0 + f * (1.0 / self.lambda_a_max - 1.0))


class ConstantStretchModel:
    def __init__(self, lam=1.0):
        self.lam = lam


class CaDrivenInternalSarcomereModel:
    """CaDrivenInternalSarcomereModel(model, calcium_field) (contraction.jl:166-175).  calcium_field: a number, a callable
    Ca(t) (spatially uniform transient) or nodal values per cell, shape (n_cells, 8) / callable t ↦ such an array."""

    def __init__(self, model, calcium_field):
        self.model, self.calcium_field = model, calcium_field

    def state(self, t):
        ca = self.calcium_field(t) if callable(self.calcium_field) else self.calcium_field
        return ca


class ActiveStressModel:
    """ActiveStressModel(material_model, active_stress_model, contraction_model, microstructure_model)
    (src/modeling/solid/materials.jl:1200-1266): P = ∂Ψ_passive/∂F + 𝓝·active_stress(F, f₀)."""

    def __init__(self, material_model, active_stress_model, contraction_model, microstructure_model):
        self.passive = PK1Model(material_model, microstructure_model)
        self.active_stress_model, self.contraction_model = active_stress_model, contraction_model
        self.material, self.microstructure = self.passive.material, self.passive.microstructure

    def internal_model(self):
        """the sarcomere model with internal state behind the contraction model (RDQ20MFModel / AsRateIndependent), or None"""
        sm = getattr(self.contraction_model, "model", None)
        return sm if hasattr(sm, "sid") else None

    def tension(self, t):
        """(scale, nodal field or None) of Ta = Tmax·𝓝 at time t — with an internal sarcomere model: of the calcium itself"""
        st = self.contraction_model.state(t) if hasattr(self.contraction_model, "state") else 1.0
        tmax = 1.0 if self.internal_model() is not None else self.active_stress_model.Tmax
        if np.ndim(st) == 0:
            return tmax * float(st), None
        return tmax, np.ascontiguousarray(st, dtype=np.float64)

    def lower(self, t=0.0):
        m = self.passive.lower()
        scale, field = self.tension(t)
        m.p[9] = scale if field is None else 0.0
        return m


class ActiveMaterialAdapter:
    """ActiveMaterialAdapter(mat): Ψᵃ(F, Fᵃ) = Ψ_mat(F·Fᵃ⁻¹) in the frame carried along by Fᵃ (src/modeling/solid/active.jl:8-21)."""

    def __init__(self, mat):
        self.mat = mat


class SimpleActiveSpring:
    """SimpleActiveSpring(; aᶠ = 1): Ψᵃ = aᶠ/2 (f₀·Cᵉ f₀ − 1)², Cᵉ = Fᵉᵀ Fᵉ, Fᵉ = F·Fᵃ⁻¹ (energies.jl:334-347)."""

    def __init__(self, af=1.0):
        self.af = float(af)


class GMKActiveDeformationGradientModel:
    """Fᵃ = I + (λᵃ − 1) f₀⊗f₀ (active.jl:23-39)."""
    adg, sheetlet_part = L.TB_ADG_GMK, 0.0


class GMKIncompressibleActiveDeformationGradientModel:
    """Fᵃ = λᵃ f₀⊗f₀ + λᵃ^(-1/2) (s₀⊗s₀ + n₀⊗n₀) (active.jl:42-62)."""
    adg, sheetlet_part = L.TB_ADG_GMK_INCOMPRESSIBLE, 0.0


class RLRSQActiveDeformationGradientModel:
    """RLRSQActiveDeformationGradientModel(sheetlet_part): Fᵃ = λᵃ f⊗f + (1 + κ(λᵃ−1)) s⊗s + n⊗n / ((1 + κ(λᵃ−1)) λᵃ) (active.jl:65-96)."""
    adg = L.TB_ADG_RLRSQ

    def __init__(self, sheetlet_part):
        self.sheetlet_part = float(sheetlet_part)


class _HillModel:
    framework = L.TB_HILL_NONE

    def __init__(self, passive_spring, active_spring, active_deformation_gradient_model, contraction_model, microstructure_model):
        self.passive = PK1Model(passive_spring, microstructure_model)
        self.active_spring, self.adg_model, self.contraction_model = active_spring, active_deformation_gradient_model, contraction_model
        self.material, self.microstructure = self.passive.material, self.passive.microstructure

    def activation(self, t):
        """(scale, nodal field or None) of the calcium state at time t"""
        st = self.contraction_model.state(t) if hasattr(self.contraction_model, "state") else 1.0
        if np.ndim(st) == 0:
            return float(st), None
        return 1.0, np.ascontiguousarray(st, dtype=np.float64)

    tension = activation

    def lower(self, t=0.0):
        m = self.passive.lower()
        scale, field = self.activation(t)
        m.p[9] = scale if field is None else 0.0
        return m

    def lower_hill(self):
        h = L.tb_hill()
        h.framework = self.framework
        a = self.active_spring
        if isinstance(a, SimpleActiveSpring):
            h.active_energy, h.active_penalty = L.TB_ACTIVE_SIMPLE_SPRING, 0
            h.active_p[0] = a.af
        elif isinstance(a, ActiveMaterialAdapter):
            am = PK1Model(a.mat, self.microstructure).lower()
            h.active_energy, h.active_penalty = am.kind, am.reserved
            for i in range(9):
                h.active_p[i] = am.p[i]
            for i in range(3):
                h.active_p[9 + i] = am.p[10 + i]
        else:
            raise TypeError("active spring: ActiveMaterialAdapter(energy) or SimpleActiveSpring expected")
        h.adg_kind, h.sheetlet_part = self.adg_model.adg, self.adg_model.sheetlet_part
        sm = getattr(self.contraction_model, "model", self.contraction_model)
        if isinstance(sm, PelceSunLangeveld1995Model):
            h.sarcomere_kind = L.TB_SARCOMERE_PELCE_SUN_LANGEVELD_1995
            h.sarcomere_p[0], h.sarcomere_p[1] = sm.beta, sm.lambda_a_max
        elif isinstance(sm, ConstantStretchModel):
            h.sarcomere_kind = L.TB_SARCOMERE_CONSTANT_STRETCH
            h.sarcomere_p[0] = sm.lam
        else:
            raise TypeError("Hill frameworks: a steady-state sarcomere model (PelceSunLangeveld1995Model, ConstantStretchModel) is expected")
        return h


class GeneralizedHillModel(_HillModel):
    """GeneralizedHillModel(passive_spring, active_spring, active_deformation_gradient_model, contraction_model,
    microstructure_model): Ψ = Ψᵖ(F) + Ψᵃ(F, Fᵃ) (src/modeling/solid/materials.jl:1042-1113)."""
    framework = L.TB_HILL_GENERALIZED


class ExtendedHillModel(_HillModel):
    """ExtendedHillModel(…): Ψ = Ψᵖ(F) + 𝓝(state)·Ψᵃ(F, Fᵃ) (materials.jl:1119-1190)."""
    framework = L.TB_HILL_EXTENDED


class RobinBC:
    """RobinBC(α, boundary_name): P·n₀ = −α u (weak_boundary_conditions.jl:23-26; energy α u·u)."""
    kind = L.TB_BC_ROBIN

    def __init__(self, alpha, boundary_name):
        self.param, self.boundary_name = float(alpha), boundary_name


class NormalSpringBC:
    """NormalSpringBC(kₛ, boundary_name): energy ½ kₛ (u·N)² (weak_boundary_conditions.jl:35-38)."""
    kind = L.TB_BC_NORMAL_SPRING

    def __init__(self, ks, boundary_name):
        self.param, self.boundary_name = float(ks), boundary_name


class ConstantPressureBC:
    """ConstantPressureBC(p, boundary_name): follower load p·J·F⁻ᵀ·n₀ (weak_boundary_conditions.jl:59-62,419-515)."""
    kind = L.TB_BC_PRESSURE

    def __init__(self, p, boundary_name):
        self.param, self.boundary_name = float(p), boundary_name


class BendingSpringBC:
    """BendingSpringBC(kᵇ, boundary_name): energy ½ kᵇ |F⁻ᵀN − N|² (weak_boundary_conditions.jl:47-57,301-415)."""
    kind = L.TB_BC_BENDING_SPRING

    def __init__(self, kb, boundary_name):
        self.param, self.boundary_name = float(kb), boundary_name


class PressureFieldBC:
    """PressureFieldBC(pressure_field, boundary_name) (weak_boundary_conditions.jl:71-77,516-632): the follower load of
    ConstantPressureBC with p = evaluate_coefficient(pc, cell, qp, t); pc: ConstantCoefficient(p) or FieldCoefficient of
    first-order nodal data per cell, shape (n_cells, 8)."""
    kind = L.TB_BC_PRESSURE_FIELD

    def __init__(self, pc, boundary_name):
        self.boundary_name = boundary_name
        self.param_of_t = None
        if callable(pc):                       # spatially uniform, time-dependent pressure p(t) (e.g. the reference's TestRampField)
            self.param, self.field, self.param_of_t = float(pc(0.0)), None, pc
        elif isinstance(pc, ConstantCoefficient):
            self.param, self.field = float(pc.val), None
        else:
            self.param, self.field = 1.0, np.ascontiguousarray(pc.data, dtype=np.float64)


class QuasiStaticModel:
    """QuasiStaticModel(:u, constitutive_model, facet_models) (test/test_elements.jl:99-125, fem.jl:597-623)."""

    def __init__(self, sym, constitutive_model, facet_models=()):
        self.sym, self.constitutive_model, self.facet_models = sym, constitutive_model, tuple(facet_models)


def material_routine(model, F, t=0.0):
    """Host evaluation of the device material routine: (Ψ, P, 𝔸) with 𝔸[3i+j, 3k+l] = ∂P_ij/∂F_kl."""
    F = np.ascontiguousarray(F, dtype=np.float64)
    psi = C.c_double()
    P = np.zeros((3, 3))
    A = np.zeros((9, 9))
    if isinstance(model, _HillModel):
        m, h = model.lower(t), model.lower_hill()
        check(lib().tb_host_material_eval_hill(C.byref(m), C.byref(h), m.p[9], F.ctypes.data_as(L.c_dp), C.byref(psi), P.ctypes.data_as(L.c_dp),
                                               A.ctypes.data_as(L.c_dp)))
        return psi.value, P, A
    m = model.lower()
    check(lib().tb_host_material_eval(C.byref(m), F.ctypes.data_as(L.c_dp), C.byref(psi), P.ctypes.data_as(L.c_dp), A.ctypes.data_as(L.c_dp)))
    return psi.value, P, A


class NonlinearOperator:
    """Operator of a quasi-static problem: `.J` (CSR nzval on device), residual vectors supplied by the caller.  `model`: a
    QuasiStaticModel, or a dict cellset-name → QuasiStaticModel (one material per subdomain, test_solid_mechanics.jl:96-140): one form
    per subdomain, the first overwrites J / residual, the others accumulate."""

    def __init__(self, strategy, model, dh, pattern, qorder=0, local_solver=None):
        self.strategy, self.dh, self.model = strategy, dh, model
        self.dmesh = dh.device_mesh(strategy.device)
        self.pattern = self.dmesh.pattern(pattern)
        self.J = DeviceVector(strategy.device, pattern.nnz)
        self.internal = None
        self.forms, self.facet_forms, self._keep, self._facet_bcs = [], [], [], []
        domains = list(model.items()) if isinstance(model, dict) else [(None, model)]
        if len(domains) > 1 and isinstance(strategy, ElementAssemblyStrategy):
            raise ValueError("multi-domain operators accumulate: use PerColorAssemblyStrategy or AtomicAssemblyStrategy")
        for k, (name, qm) in enumerate(domains):
            cm = qm.constitutive_model
            mat = cm.lower()
            form = C.c_void_p()
            check(lib().tb_hyperelastic_create(self.dmesh.h, qorder, C.byref(mat), C.byref(form)))
            self.forms.append((form, cm))
            self._keep.append(mat)
            if name is not None:
                cells = np.ascontiguousarray(dh.grid.getcellset(name), dtype=np.int32)
                check(lib().tb_form_set_cellset(form, cells.ctypes.data_as(L.c_i32p), len(cells), 0))
            if k > 0:
                check(lib().tb_form_set_accumulate(form, 1))
            if isinstance(cm, PrestressedMechanicalModel):
                check(lib().tb_hyperelastic_set_prestress(form, cm.F0inv.ctypes.data_as(L.c_dp)))
                cm = cm.inner_model
                self.forms[-1] = (form, cm)
            if isinstance(cm, _HillModel):
                h = cm.lower_hill()
                self._keep.append(h)
                check(lib().tb_hyperelastic_set_hill(form, C.byref(h)))
            # condensed internal variable (QuasiStaticCondensedElementCache): states per quadrature point on the device
            sm = cm.internal_model() if isinstance(cm, ActiveStressModel) else None
            if sm is not None:
                if self.internal is not None:
                    raise NotImplementedError("one subdomain with condensed internal variables per operator")
                self.internal_form = form
                ls = local_solver or GenericLocalNonlinearSolver()
                pp = sm.params()
                check(lib().tb_hyperelastic_set_condensation(form, sm.sid, pp.ctypes.data_as(L.c_dp), len(pp), cm.active_stress_model.Tmax, ls.tol, ls.max_iters))
                npts = C.c_int64()
                check(lib().tb_hyperelastic_n_quadrature_points(form, C.byref(npts)))
                self.internal = SarcomereState(strategy.device, sm, npts.value)        # Q: current iterate / solution
                self.internal_known = SarcomereState(strategy.device, sm, npts.value)  # Q_known: accepted state of the previous step
                self.dt = None
                self.u_prev = None
                if not sm.rate_independent:      # rate-coupled local problem (QuasiStaticCondensedDAEElementCache): Ḟ from the previous displacement
                    self.u_prev = strategy.device.zeros(dh.ndofs)
                    check(lib().tb_hyperelastic_set_previous_solution(form, self.u_prev.ptr))
            # surface terms: one facet form per weak boundary condition (setup_boundary_cache, weak_boundary_conditions.jl:1-7)
            for bc in getattr(qm, "facet_models", ()):
                fs = bc.boundary_name if not isinstance(bc.boundary_name, str) else dh.grid.facetset(bc.boundary_name)
                fs = np.ascontiguousarray(fs, dtype=np.int32).reshape(-1, 2)
                if name is not None:                       # the surface terms of a subdomain act on its own cells' facets
                    fs = fs[np.isin(fs[:, 0], dh.grid.getcellset(name))]
                h = C.c_void_p()
                check(lib().tb_facet_form_create(self.dmesh.h, bc.kind, bc.param, 0, fs.ctypes.data_as(L.c_i32p), len(fs), 0, C.byref(h)))
                if getattr(bc, "field", None) is not None:
                    check(lib().tb_facet_form_set_field(h, bc.field.ctypes.data_as(L.c_dp), bc.field.size))
                self.facet_forms.append(h)
                self._facet_bcs.append(bc)
        self.form = self.forms[0][0]

    def __del__(self):
        try:
            for h in self.facet_forms:
                lib().tb_form_destroy(h)
            for form, _ in self.forms:
                lib().tb_form_destroy(form)
        except Exception:
            pass


def set_timestep(op, dt):
    """Δt of the internal variable's backward Euler step (GenericFirstOrderTimeParameters.Δt, euler.jl:490-493)."""
    op.dt = float(dt)
    check(lib().tb_hyperelastic_set_internal_state(op.internal_form, op.internal.u.ptr, op.internal_known.u.ptr, op.dt))


def accept_internal_state(op):
    """the time step was accepted: Q_known ← Q"""
    check(lib().tb_memcpy_d2d(op.strategy.device.h, op.internal_known.u.ptr, op.internal.u.ptr, op.internal.u.nbytes))


def reject_internal_state(op):
    """the time step was rejected: Q ← Q_known (initial guess of the retry)"""
    check(lib().tb_memcpy_d2d(op.strategy.device.h, op.internal.u.ptr, op.internal_known.u.ptr, op.internal.u.nbytes))


def local_solve_failures(op):
    """number of quadrature points whose local solve failed in the last assembly (check_local_solve_convergence)"""
    nf = C.c_int64()
    check(lib().tb_hyperelastic_local_solve_report(op.internal_form, C.byref(nf), None, 0))
    return nf.value


def _g_deuflhard(x):
    return np.sqrt(1.0 + 4.0 * x) - 1.0


class Deuflhard2004DiscreteContinuationController:
    """Deuflhard2004DiscreteContinuationController(; Θmin, p, Θreject = 0.95, Θbar = 0.5, γ = 0.95, qmin = 1/5, qmax = 5)
    (src/solver/time/homotopy.jl:204-281): step-size control of the load path from Newton's contraction history Θₖ — accept when every
    Θₖ ≤ Θreject (only finiteness without monotonicity enforcement); on rejection dt ← clamp(γ (g(Θ̄)/g(Θₖ))^(1/p)) dt for the first
    offending Θₖ; after an accepted step dt ← clamp(γ (g(Θ̄)/(2Θ₀))^(1/p)) dt with Θ₀ = max(Θ₁, Θmin); g(x) = √(1+4x) − 1."""
    variant = "A"

    def __init__(self, theta_min=1.0 / 8.0, p=1, theta_reject=0.95, theta_bar=0.5, gamma=0.95, qmin=0.2, qmax=5.0):
        self.theta_min, self.p, self.theta_reject, self.theta_bar = theta_min, p, theta_reject, theta_bar
        self.gamma, self.qmin, self.qmax = gamma, qmin, qmax

    def _clamp(self, q):
        return min(max(q, self.qmin), self.qmax)

    def should_accept_step(self, thetas, enforce_monotonic_convergence=True):
        th = np.asarray(thetas, dtype=np.float64)
        return bool(np.all(th <= self.theta_reject)) if enforce_monotonic_convergence else bool(np.all(np.isfinite(th)))

    def reject_step(self, dt, thetas):
        for th in thetas:
            if th > self.theta_reject:
                return self._clamp(self.gamma * (_g_deuflhard(self.theta_bar) / _g_deuflhard(th)) ** (1.0 / self.p)) * dt
        return dt

    def _theta0(self, thetas):
        return max(thetas[0], self.theta_min) if len(thetas) else self.theta_min

    def _denominator(self, theta0):
        return 2.0 * theta0

    def adapt_dt(self, dt, thetas):
        return self._clamp(self.gamma * (_g_deuflhard(self.theta_bar) / self._denominator(self._theta0(thetas))) ** (1.0 / self.p)) * dt


class Deuflhard2004_B_DiscreteContinuationControllerVariant(Deuflhard2004DiscreteContinuationController):
    """the variant with g(Θ₀) in the denominator of the predictor (homotopy.jl:283-342) — the default controller of HomotopyPathSolver (:406-408)"""
    variant = "B"

    def _denominator(self, theta0):
        return _g_deuflhard(theta0)


class ExperimentalDiscreteContinuationController(Deuflhard2004DiscreteContinuationController):
    """ExperimentalDiscreteContinuationController(; Θmin, p, Θreject = 0.9, Θbar = 0.75, …) (homotopy.jl:344-400): the mean contraction rate
    drives the predictor, the largest one the rejection."""
    variant = "experimental"

    def __init__(self, theta_min=1.0 / 8.0, p=1, theta_reject=0.9, theta_bar=0.75, gamma=0.95, qmin=0.2, qmax=5.0):
        super().__init__(theta_min, p, theta_reject, theta_bar, gamma, qmin, qmax)

    def reject_step(self, dt, thetas):
        return self._clamp(self.gamma * (_g_deuflhard(self.theta_bar) / _g_deuflhard(max(thetas))) ** (1.0 / self.p)) * dt

    def _theta0(self, thetas):
        return max(float(np.mean(thetas)), self.theta_min) if len(thetas) else self.theta_min


class HomotopyPathSolver:
    """HomotopyPathSolver(inner_solver) (src/solver/time/homotopy.jl): solve F(u, t) = 0 along the pseudo-time t with a Newton solve per
    load step; `solve` mirrors the adaptive integrator around it: the controller (default Deuflhard2004_B…Variant(Θmin = 1/8, p = 1))
    judges Newton's contraction history, a rejected or failed step is rolled back (u restored) and retried with a shorter increment
    (controller law for a poor contraction rate; failfactor 1/2 for a failed solve), an accepted one adapts the next increment."""

    def __init__(self, inner_solver, controller=None):
        self.inner_solver = inner_solver
        self.controller = controller or Deuflhard2004_B_DiscreteContinuationControllerVariant()
        self.steps = []                      # (t, dt, newton iterations, accepted)

    def solve(self, u, op, ch, tspan, dt, adaptive=True, dtmin=1e-6, maxiters=200):
        if getattr(op, "internal", None) is not None:
            # check_internal_variables_are_rate_free (homotopy.jl:22-58): continuation has neither a previous solution nor a timestep
            raise ValueError("the material carries an internal variable with a time derivative, which HomotopyPathSolver cannot integrate: "
                             "continuation supplies neither a previous solution nor a timestep. Use perform_mechanics_step (backward Euler) instead")
        t, t_end = float(tspan[0]), float(tspan[1])
        self.steps = []
        ns = self.inner_solver
        while t < t_end - 1e-12 * max(1.0, abs(t_end)):
            if len(self.steps) >= maxiters:
                return False
            h = min(dt, t_end - t)
            u0 = u.to_host()
            solved = nlsolve(u, op, ch, ns, t=t + h)
            thetas = [th for th in ns.theta if np.isfinite(th)] if solved else list(ns.theta)
            ok = solved and (not adaptive or self.controller.should_accept_step(thetas, ns.enforce_monotonic_convergence))
            self.steps.append((t + h, h, ns.iter, bool(ok)))
            if ok:
                t += h
                if adaptive:
                    dt = self.controller.adapt_dt(h, thetas)
            else:
                u.copy_from_host(u0)                           # rollback_state!
                if not adaptive:
                    return False
                dt = 0.5 * h if not solved else self.controller.reject_step(h, thetas)
                if dt < dtmin or dt >= h:
                    if dt >= h:
                        dt = 0.5 * h
                    if dt < dtmin:
                        return False
        return True


def perform_mechanics_step(u, op, ch, solver, t, dt):
    """One backward-Euler step t → t + Δt of a quasi-static problem with condensed internal variables: the multi-level Newton of the
    reference (BackwardEulerSolver(inner_solver = MultiLevelNewtonRaphsonSolver), euler.jl / multilevel_newton_raphson.jl) — global
    Newton on u, the local problems re-solved inside every assembly from the last iterate, Q_known = the accepted state.  A step with a
    failed local solve or a diverged global Newton is rejected: u and Q are restored and False is returned."""
    u0 = u.to_host()
    set_timestep(op, dt)
    if getattr(op, "u_prev", None) is not None:
        op.u_prev.copy_from_host(u0)                    # backward-Euler rate: Ḟ = (∇u − ∇u_prev)/Δt (AffineVelocity(inv(Δt), uₑprev))
    ok = nlsolve(u, op, ch, solver, t=t + dt)
    if ok and local_solve_failures(op) == 0:
        accept_internal_state(op)
        return True
    u.copy_from_host(u0)
    reject_internal_state(op)
    return False


def _sync_active_tension(op, t):
    for h, bc in zip(op.facet_forms, op._facet_bcs):
        if getattr(bc, "param_of_t", None) is not None:
            check(lib().tb_facet_form_set_param(h, float(bc.param_of_t(t))))
    for form, cm in op.forms:
        if isinstance(cm, (ActiveStressModel, _HillModel)):
            scale, field = cm.tension(t)
            check(lib().tb_hyperelastic_set_active_tension(form, float(scale), None if field is None else field.ctypes.data_as(L.c_dp),
                                                           0 if field is None else field.size))


def update_linearization(op, u, t=0.0, residual=None):
    """update_linearization!(op, residual, u, p) / update_linearization!(op, u, p) (newton_raphson.jl:238): volume terms of every
    subdomain, then the surface terms, all accumulated into the same J / residual."""
    _sync_active_tension(op, t)
    for form, _ in op.forms:
        check(lib().tb_linearize(form, op.pattern.h, op.strategy.code, _ptr(u), float(t), op.J.ptr, _ptr(residual)))
    for h in op.facet_forms:
        check(lib().tb_facet_assemble(h, op.pattern.h, _ptr(u), float(t), op.J.ptr, _ptr(residual)))
    return op


def residual(op, residual, u, t=0.0):
    """residual!(op, residual, u, p) (newton_raphson.jl:234)."""
    _sync_active_tension(op, t)
    for form, _ in op.forms:
        check(lib().tb_residual(form, op.strategy.code, _ptr(u), float(t), _ptr(residual)))
    for h in op.facet_forms:
        check(lib().tb_facet_assemble(h, None, _ptr(u), float(t), None, _ptr(residual)))
    return residual


# --------------------------------------------------------------------------------------- constraints + Newton–Raphson
class ConstraintHandler:
    """ConstraintHandler(dh) with Dirichlet conditions on whole dofs (Ferrite, third party; used through apply_zero! /
    apply!).  `prescribed_dofs`: dof ids; `values`: their prescribed values (inhomogeneities; default 0)."""

    def __init__(self, dh, prescribed_dofs, values=None):
        self.dh = dh
        self.prescribed_dofs = np.unique(np.asarray(prescribed_dofs, dtype=np.int64))
        self.values = np.zeros(len(self.prescribed_dofs)) if values is None else np.asarray(values, dtype=np.float64)
        self._flags = {}

    def flags(self, device):
        key = id(device)
        if key not in self._flags:
            f = np.zeros(self.dh.ndofs, dtype=np.uint8)
            f[self.prescribed_dofs] = 1
            self._flags[key] = device.to_device(f)
        return self._flags[key]

    def free_dofs(self):
        return np.setdiff1d(np.arange(self.dh.ndofs), self.prescribed_dofs)


def apply(u, ch):
    """apply!(u, ch): write the prescribed values into a solution vector (host round trip: setup-time operation)."""
    h = u.to_host()
    h[ch.prescribed_dofs] = ch.values
    u.copy_from_host(h)
    return u


def meandiag(op_or_pattern, nz):
    pat = getattr(op_or_pattern, "pattern", op_or_pattern)
    out = C.c_double()
    check(lib().tb_meandiag(pat.h, _ptr(nz), C.byref(out)))
    return out.value


def apply_zero(K, f, ch, pattern=None, diag=None):
    """apply_zero!(K, f, ch) / apply_zero!(f, ch) on the device (src/utils.jl:263-278, nlsolve_common.jl:12-26).
    K: CSR nzval DeviceVector (or None), f: DeviceVector (or None)."""
    dev = (K if K is not None else f).dev
    if K is not None and diag is None:
        diag = meandiag(pattern, K)
    check(lib().tb_apply_zero_csr(pattern.h, _ptr(K), _ptr(f), ch.flags(dev).ptr, float(diag if diag is not None else 1.0)))


def dot(x, y):
    out = C.c_double()
    check(lib().tb_dot(x.dev.h, x.n, x.ptr, y.ptr, C.byref(out)))
    return out.value


def norm(x):
    return float(np.sqrt(dot(x, x)))


class EisenstatWalkerForcing:
    """EisenstatWalkerForcing(; η₀ = 0.5, ηₘₐₓ = 0.9, γ = 0.9, α = 2, safeguard = true, safeguard_threshold = 0.1)
    (newton_raphson.jl:1-41,158-178): ηₖ = γ (‖rₖ‖/‖rₖ₋₁‖)^α becomes the relative tolerance of the k-th inner Krylov solve."""

    def __init__(self, eta0=0.5, eta_max=0.9, gamma=0.9, alpha=2.0, safeguard=True, safeguard_threshold=0.1):
        self.eta0, self.eta_max, self.gamma, self.alpha = eta0, eta_max, gamma, alpha
        self.safeguard, self.safeguard_threshold = safeguard, safeguard_threshold
        self.eta, self.rnorm = eta0, 0.0

    def prestep(self, residualnorm, it):
        if it == 0:
            self.eta = min(self.eta0, self.eta_max)
        else:
            eta = self.gamma * (residualnorm / self.rnorm) ** self.alpha
            if self.safeguard:
                sg = self.gamma * self.eta ** self.alpha
                if sg > self.safeguard_threshold and sg > eta:
                    eta = sg
            self.eta = min(max(eta, 0.0), self.eta_max)
        self.rnorm = residualnorm
        return self.eta


class NewtonRaphsonSolver:
    """NewtonRaphsonSolver(; max_iter, tol, inner_solver, forcing, simplified_newton) (src/solver/nonlinear/newton_raphson.jl:1-60); nlsolve!
    follows :215-320 — update_linearization!, eliminate constraints, residual norm over the free dofs, linear solve,
    eliminate the increment, u .-= Δu, Θₖ contraction monitor, early exits.  inner_solver: "cg" (Jacobi-PCG; symmetric positive
    definite tangents) or "gmres" (restarted, right-Jacobi; the reference's default KrylovJL_GMRES — for indefinite or
    non-symmetric tangents), or — LinearSolve.jl's pluggability — any callable (pattern, J, residual, Δu) → iterations that leaves the
    solution of J Δu = residual in Δu (device vectors; the CSR structure is `pattern.sp.rowptr/colidx` on the host, J its values on the device)."""

    def __init__(self, max_iter=100, tol=1e-4, inner_rtol=1e-8, inner_atol=1e-14, inner_maxiter=5000, enforce_monotonic_convergence=True,
                 inner_solver="cg", gmres_restart=50, inner_precond=None, simplified_newton=False, forcing=None, strict_inner_solve=True):
        if inner_solver not in ("cg", "gmres") and not callable(inner_solver):
            raise ValueError("inner_solver: 'cg', 'gmres' or a callable (pattern, J, residual, Δu) -> linear iterations")
        self.inner_solver, self.gmres_restart = inner_solver, gmres_restart
        self.inner_precond = inner_precond          # None (Jacobi, device-scalar CG), L1GSPrecBuilder(partsize) or ChebyshevPrecBuilder(degree)
        # simplified_newton: the tangent of the first iteration is reused, later iterations assemble the residual only (residual!);
        # forcing: EisenstatWalkerForcing() adapts the inner Krylov tolerance (ignored by callable inner solvers, as by direct ones)
        self.simplified_newton, self.forcing = bool(simplified_newton), forcing
        self.max_iter, self.tol = max_iter, tol
        self.inner_rtol, self.inner_atol, self.inner_maxiter = inner_rtol, inner_atol, inner_maxiter
        self.enforce_monotonic_convergence = enforce_monotonic_convergence
        self.iter, self.theta, self.residual_norms, self.linear_iters = -1, [], [], []
        self.linear_failure = None
        # strict_inner_solve (default, the reference's behaviour — `solve_succeeded || return false`, newton_raphson.jl:266-269): an inner Krylov
        # solve that stops above its tolerance fails the nonlinear solve.  False: the increment is applied as an inexact Newton step and the
        # event is recorded in `linear_failure` (restarted GMRES on an indefinite tangent may stagnate above a tight tolerance while the outer
        # iteration still contracts); callers opt out explicitly
        self.strict_inner_solve = bool(strict_inner_solve)


def nlsolve(u, op, ch, solver, t=0.0):
    """nlsolve!(u, stage, cache, t) → Bool.  `u` must already satisfy the Dirichlet values (apply!(u, ch))."""
    dev = u.dev
    res = DeviceVector(dev, u.n)
    du = DeviceVector(dev, u.n)
    solver.iter, solver.theta, solver.residual_norms, solver.linear_iters = -1, [], [], []
    rprev = iprev = 0.0
    eps = np.finfo(np.float64).eps
    while True:
        solver.iter += 1
        if solver.simplified_newton and solver.iter > 0:
            residual(op, res, u, t)                            # the eliminated tangent of iteration 0 stays in op.J
            apply_zero(None, res, ch, pattern=op.pattern)
        else:
            update_linearization(op, u, t, residual=res)
            apply_zero(op.J, res, ch, pattern=op.pattern)
        solver.jacobian_is_fresh = not (solver.simplified_newton and solver.iter > 0)
        rnorm = norm(res)                                     # prescribed entries are zero: this is the norm over the free dofs
        solver.residual_norms.append(rnorm)
        if rnorm < solver.tol and solver.iter > 0:
            solver.theta.append(0.0)
            break
        if solver.iter > solver.max_iter or not np.isfinite(rnorm):
            solver.theta.append(np.inf)
            return False
        du.fill_zero()
        inner_rtol = solver.inner_rtol if solver.forcing is None else solver.forcing.prestep(rnorm, solver.iter)
        try:
            if callable(solver.inner_solver):
                its = solver.inner_solver(op.pattern, op.J, res, du)
            elif solver.inner_solver == "cg" and solver.inner_precond is not None:
                its, lres = pcg_solve(op.pattern, op.J, res, du, inner_rtol, solver.inner_atol, solver.inner_maxiter, solver.inner_precond)
            elif solver.inner_solver == "gmres":
                its, lres = gmres_solve(op.pattern, op.J, res, du, inner_rtol, solver.inner_atol, solver.inner_maxiter, solver.gmres_restart, True)
            else:
                its, lres = cg_solve(op.pattern, op.J, res, du, inner_rtol, solver.inner_atol, solver.inner_maxiter, True)
            if not callable(solver.inner_solver) and not solve_converged(op.pattern, lres):
                # newton_raphson.jl:266-269 `solve_succeeded || return false`: an inner solve that ran into its iteration limit fails the step
                # unless the solver was built with strict_inner_solve=False (inexact Newton step, event recorded)
                solver.linear_failure = "inner linear solve stopped at %d iterations with residual %.3e above its tolerance" % (its, lres)
                if solver.strict_inner_solve:
                    solver.linear_iters.append(its)
                    solver.theta.append(np.inf)
                    return False
        except TBError as e:
            # a failed inner linear solve fails the nonlinear solve (newton_raphson.jl:262-270: `solve_inner_linear_system!` → false) — e.g. CG
            # meeting an indefinite tangent after too large a load step; the caller retries with a smaller step or another inner solver
            solver.linear_failure = str(e)
            solver.theta.append(np.inf)
            return False
        solver.linear_iters.append(its)
        apply_zero(None, du, ch, pattern=op.pattern)          # eliminate_constraints_from_increment!
        check(lib().tb_axpy(dev.h, u.n, -1.0, du.ptr, u.ptr))  # u .-= Δu
        inorm = norm(du)
        if solver.iter > 0:
            theta = min(rnorm / rprev, inorm / iprev) if rprev > 0.0 and iprev > 0.0 else 0.0
            solver.theta.append(theta)
            if rnorm < eps or inorm < eps:
                break
            if solver.enforce_monotonic_convergence and theta >= 1.0:
                return False
        rprev, iprev = rnorm, inorm
    return True


# --------------------------------------------------------------------------------------- sarcomere models with internal state
class RDQ20MFModel:
    """RDQ20MFModel(; …) — mean-field Regazzoni–Dedè–Quarteroni 2020 sarcomere model, 20 states (contraction.jl:337-376); keyword
    names as the reference's fields (ASCII: SL0, Kd0, alphaKd, mu, gamma, r0, alpha, mu0_fP, mu1_fP, eps_v)."""
    sid = L.TB_SARCOMERE_RDQ20MF
    _fields = ("LA", "LM", "LB", "SL0", "Q", "Kd0", "alphaKd", "mu", "gamma", "Koff", "Kbasic", "r0", "alpha", "mu0_fP", "mu1_fP", "a_XB", "eps_v")
    _defaults = (1.25, 1.65, 0.18, 2.2, 2.0, 0.381, -0.571, 10.0, 12.0, 0.1, 0.013, 0.13431, 25.184, 0.032653, 0.000778, 22.894e3, 1.0e-6)

    def __init__(self, **kw):
        for n, v in zip(self._fields, self._defaults):
            setattr(self, n, float(kw.pop(n, v)))
        if kw:
            raise TypeError("RDQ20MFModel: unknown parameter(s) %s" % sorted(kw))

    def params(self):
        return np.array([getattr(self, n) for n in self._fields], dtype=np.float64)

    rate_independent = False


class AsRateIndependent:
    """AsRateIndependent(model): the model evaluated at zero shortening velocity (contraction.jl:107-148)."""
    rate_independent = True

    def __init__(self, model):
        self.model = model
        self.sid = model.sid

    def params(self):
        return self.model.params()


def default_sarcomere_state(model, n_points=1):
    """default_initial_state!(Q, model): Q[1] = 1, the rest 0 (contraction.jl:371-375); shape (n_states, n_points)."""
    u = np.zeros((num_states(model), n_points))
    u[0] = 1.0
    return u


def sarcomere_rhs(model, u, stretch, velocity, calcium):
    """sarcomere_rhs!(du, u, λ, dλdt, Ca, t, model) evaluated on the host by the code the kernel runs → (du, Ta, As)."""
    p = model.params()
    u = np.ascontiguousarray(u, dtype=np.float64)
    du = np.zeros_like(u)
    Ta, As = C.c_double(), C.c_double()
    check(lib().tb_host_sarcomere_eval(model.sid, p.ctypes.data_as(L.c_dp), len(p), u.ctypes.data_as(L.c_dp), float(stretch),
                                       0.0 if model.rate_independent else float(velocity), float(calcium), du.ctypes.data_as(L.c_dp),
                                       C.byref(Ta), C.byref(As)))
    return du, Ta.value, As.value


def compute_active_tension(model, state, sarcomere_stretch):
    return sarcomere_rhs(model, state, sarcomere_stretch, 0.0, 0.0)[1]


def compute_active_stiffness(model, state, sarcomere_stretch):
    return sarcomere_rhs(model, state, sarcomere_stretch, 0.0, 0.0)[2]


def internal_state_in_bounds(model, Q):
    """internal_state_in_bounds(::RDQ20MFModel, Q) = all(≥(0), Q[1:16]) (contraction.jl:596)."""
    return bool(np.all(np.asarray(Q)[:16] >= 0))


class SarcomereState:
    """Internal states of a sarcomere model at n_points points on the device, (n_states, n_points), point-fastest."""

    def __init__(self, device, model, n_points, initial=None):
        self.device, self.model, self.n_points = device, model, int(n_points)
        self.n_states = num_states(model)
        u = default_sarcomere_state(model, self.n_points) if initial is None else np.ascontiguousarray(initial, dtype=np.float64)
        if u.shape != (self.n_states, self.n_points):
            raise ValueError("SarcomereState: initial state must have shape (n_states, n_points)")
        self.u = device.to_device(u.ravel()) if self.n_points else DeviceVector(device, 0)

    def to_host(self):
        return self.u.to_host().reshape(self.n_states, self.n_points)


def _dev_or_scalar(x):
    if isinstance(x, DeviceVector):
        return x.ptr, 0.0
    return None, float(x)


def sarcomere_step(state, t, dt, stretch, velocity, calcium, substeps=1, tension=None, stiffness=None):
    """One forward-Euler step (or `substeps` of them with held inputs) of du = sarcomere_rhs!(u, λ, dλdt, Ca) at every point — the
    StandaloneSarcomereModel protocol (contraction.jl:150-163).  Inputs: numbers or DeviceVectors of per-point values."""
    m = state.model
    p = m.params()
    ps, s = _dev_or_scalar(stretch)
    pv, v = _dev_or_scalar(velocity)
    pc, c = _dev_or_scalar(calcium)
    check(lib().tb_sarcomere_step(state.device.h, m.sid, p.ctypes.data_as(L.c_dp), len(p), state.u.ptr, state.n_points, ps, pv, pc, s, v, c,
                                  float(t), float(dt), int(substeps), int(m.rate_independent), _ptr(tension), _ptr(stiffness)))


def sarcomere_stepper(state, dt, tension=None, stiffness=None):
    """A bound single-step call with scalar inputs, step(λ, dλdt, Ca), for launch-rate-bound host loops."""
    m = state.model
    p = m.params()
    fn = lib().tb_sarcomere_step
    args = (state.device.h, m.sid, p.ctypes.data_as(L.c_dp), len(p), state.u.ptr, state.n_points, None, None, None)
    tail = (0.0, float(dt), 1, int(m.rate_independent), _ptr(tension), _ptr(stiffness))

    def step(lam, vel, ca):
        rc = fn(*args, lam, vel, ca, *tail)
        if rc:
            check(rc)
    step._keep = p
    return step


def sarcomere_derivatives(model, u, stretch, velocity, calcium, analytic=True):
    """(∂rhs/∂u [20×20], ∂rhs/∂λ, ∂rhs/∂λ̇, rhs) at one point: the kernels' hand-derived linearisation, or forward mode (analytic=False)"""
    p = model.params()
    u = np.ascontiguousarray(u, dtype=np.float64)
    D, gl, gv, f = np.zeros((20, 20)), np.zeros(20), np.zeros(20), np.zeros(20)
    check(lib().tb_host_sarcomere_derivatives(model.sid, p.ctypes.data_as(L.c_dp), len(p), u.ctypes.data_as(L.c_dp), float(stretch), float(velocity),
                                              float(calcium), int(analytic), D.ctypes.data_as(L.c_dp), gl.ctypes.data_as(L.c_dp), gv.ctypes.data_as(L.c_dp),
                                              f.ctypes.data_as(L.c_dp)))
    return D, gl, gv, f


class GenericLocalNonlinearSolver:
    """GenericLocalNonlinearSolver(; max_iters = 10, tol = 1e-4) (multilevel_newton_raphson.jl:1-4)."""

    def __init__(self, max_iters=10, tol=1e-4):
        self.max_iters, self.tol = int(max_iters), float(tol)


def sarcomere_local_solve(model, Qguess, Qknown, stretch, calcium, dt, local_solver=None, velocity=None):
    """Host evaluation of the local problem (solve_internal_timestep + corrector(s), materials.jl:1403-1568):
    → (status, Q, dQ/dλ, iterations, last residual norm[, dQ/d(dλ/dt) when a velocity is given: the rate-coupled form])."""
    ls = local_solver or GenericLocalNonlinearSolver()
    p = model.params()
    Q = np.ascontiguousarray(Qguess, dtype=np.float64).copy()
    Qk = np.ascontiguousarray(Qknown, dtype=np.float64)
    dQ, dQv = np.zeros(20), np.zeros(20)
    st, it, rn = C.c_int(), C.c_int(), C.c_double()
    check(lib().tb_host_sarcomere_local_solve(model.sid, p.ctypes.data_as(L.c_dp), len(p), Q.ctypes.data_as(L.c_dp), Qk.ctypes.data_as(L.c_dp),
                                              float(stretch), 0.0 if velocity is None else float(velocity), float(calcium), float(dt), ls.tol, ls.max_iters,
                                              dQ.ctypes.data_as(L.c_dp), None if velocity is None else dQv.ctypes.data_as(L.c_dp),
                                              C.byref(st), C.byref(it), C.byref(rn)))
    if velocity is None:
        return st.value, Q, dQ, it.value, rn.value
    return st.value, Q, dQ, it.value, rn.value, dQv


def sarcomere_implicit_step(state, known, dt, stretch, calcium, local_solver=None, dstate_dstretch=None, status=None, count_failures=True,
                            velocity=None, dstate_dvelocity=None):
    """Backward-Euler step of the internal states at every point with the stretch and calcium frozen (the local problem of the
    condensed mechanics); `state` holds the initial guess and receives the solution, `known` is Q_known (SarcomereState).  Returns the
    number of failed points (or None with count_failures=False: no synchronisation)."""
    ls = local_solver or GenericLocalNonlinearSolver()
    m = state.model
    p = m.params()
    ps, s = _dev_or_scalar(stretch)
    pc, c = _dev_or_scalar(calcium)
    pv, v = _dev_or_scalar(0.0 if velocity is None else velocity)
    nf = C.c_int64()
    check(lib().tb_sarcomere_implicit_step(state.device.h, m.sid, p.ctypes.data_as(L.c_dp), len(p), state.u.ptr, known.u.ptr, state.n_points, ps, pv, pc, s, v, c,
                                           float(dt), ls.tol, ls.max_iters, _ptr(dstate_dstretch), _ptr(dstate_dvelocity), _ptr(status),
                                           C.byref(nf) if count_failures else None))
    return nf.value if count_failures else None


class StandaloneSarcomereModel:
    """StandaloneSarcomereModel(model, calcium, fiber_stretch, fiber_velocity): inputs as functions of t (contraction.jl:150-163)."""

    def __init__(self, model, calcium, fiber_stretch, fiber_velocity):
        self.model, self.calcium, self.fiber_stretch, self.fiber_velocity = model, calcium, fiber_stretch, fiber_velocity

    def step(self, state, t, dt, **kw):
        sarcomere_step(state, t, dt, self.fiber_stretch(t), self.fiber_velocity(t), self.calcium(t), **kw)
