"""CPU checks for the quasi-static hyperelastic path: the oracle's hyper-dual differentiation of Ψ (what
Tensors.hessian does, materials.jl:1025-1040) against the reference's own assertions (Ψ(I)=0, P(I)=0,
test/test_type_stability.jl:29-63; the three assemble_element! variants agree, test/test_elements.jl:99-150)
and the shipped hand-derived device routine (compiled for the host) against that oracle."""
import numpy as np
import pytest

FSN = np.stack([np.array([1, 1, 0.0]) / np.sqrt(2), np.array([-1, 1, 0.0]) / np.sqrt(2), np.array([0, 0, 1.0])])


def test_energy_identities(oracle):
    psi, P, A = oracle.ho_energy(np.eye(3))
    assert psi == 0.0 and np.abs(P).max() == 0.0                 # Ψ(I) = 0, P(I) = 0
    np.testing.assert_allclose(A, A.T, atol=1e-13)               # major symmetry of ∂²Ψ/∂F²
    rng = np.random.default_rng(0)
    F = np.eye(3) + 0.1 * rng.normal(size=(3, 3))
    psi, P, A = oracle.ho_energy(F, fsn=FSN)
    h = 1e-6
    for i in range(3):
        for j in range(3):
            Fp, Fm = F.copy(), F.copy()
            Fp[i, j] += h; Fm[i, j] -= h
            assert abs((oracle.ho_energy(Fp, fsn=FSN)[0] - oracle.ho_energy(Fm, fsn=FSN)[0]) / (2 * h) - P[i, j]) < 1e-8 * np.abs(P).max()
            dP = (oracle.ho_energy(Fp, fsn=FSN)[1] - oracle.ho_energy(Fm, fsn=FSN)[1]) / (2 * h)
            np.testing.assert_allclose(A[:, 3 * i + j], dP.ravel(), atol=1e-7 * np.abs(A).max())
    # frame indifference of Ψ: Ψ(QF) = Ψ(F)
    th = 0.7
    Q = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    np.testing.assert_allclose(oracle.ho_energy(Q @ F, fsn=FSN)[0], psi, rtol=1e-13)


@pytest.mark.parametrize("stretch", [0.8, 1.0, 1.25])
def test_device_material_routine_matches_ad_oracle(tb, oracle, stretch):
    """I₄ ≥ 1 switches (energies.jl:160-165): compression along f (off), identity (boundary), tension (on)."""
    fsn = np.eye(3) if stretch == 1.0 else FSN   # exact unit frame for the I₄ == 1 boundary case
    model = tb.PK1Model(tb.HolzapfelOgden2009Model(), tb.ConstantCoefficient(tb.OrthotropicMicrostructure(*fsn)))
    rng = np.random.default_rng(3)
    R = np.stack(fsn).T                      # columns f, s, n
    F = R @ np.diag([stretch, 1.0 / np.sqrt(stretch), 1.0 / np.sqrt(stretch)]) @ R.T
    F = np.eye(3) if stretch == 1.0 else F + 0.02 * rng.normal(size=(3, 3))   # I₄ == 1 exactly: branch is taken (>=)
    psi, P, A = tb.material_routine(model, F)
    psi0, P0, A0 = oracle.ho_energy(F, fsn=fsn)
    I4f = np.linalg.norm(F @ fsn[0]) ** 2
    assert (I4f >= 1.0) == (stretch >= 1.0)
    np.testing.assert_allclose(psi, psi0, rtol=1e-12, atol=1e-16)
    np.testing.assert_allclose(P, P0, rtol=0, atol=1e-12 * max(np.abs(P0).max(), 1e-3))
    np.testing.assert_allclose(A, A0, rtol=0, atol=1e-12 * np.abs(A0).max())


def test_element_variants_agree_and_composite(oracle):
    """test/test_elements.jl:99-150: K+r, r-only and K-only calls agree; uₑ = ±1e-4 pattern of :52-78."""
    xyz, conn = oracle.generate_grid_hex(1, 1, 1)
    for kind, order, q in ((oracle.HEX8, 1, 2), (oracle.HEX27, 2, 3)):
        cd, nd = oracle.close_dofs(kind, 3, conn, len(xyz))
        m = oracle.Mesh(kind, q, xyz, conn, cd)
        nb = nd // 3
        rng = np.random.default_rng(order)
        ue = 1e-4 * np.sign(rng.normal(size=nd))
        K1, r1 = oracle.element_hyperelastic(m, 0, ue)
        K2, _ = oracle.element_hyperelastic(m, 0, ue, want_r=False)
        _, r2 = oracle.element_hyperelastic(m, 0, ue, want_K=False)
        assert np.abs(K1).max() > 0 and np.abs(r1).max() > 0
        np.testing.assert_array_equal(K1, K2)
        np.testing.assert_array_equal(r1, r2)
        np.testing.assert_allclose(K1, K1.T, atol=1e-12 * np.abs(K1).max())
        # rigid translation is in the kernel of the tangent; the residual of u = 0 vanishes
        t = np.tile([0.3, -0.2, 0.5], nb)
        np.testing.assert_allclose(K1 @ t, 0, atol=1e-11 * np.abs(K1).max())
        _, r0 = oracle.element_hyperelastic(m, 0, np.zeros(nd), want_K=False)
        np.testing.assert_allclose(r0, 0, atol=1e-16)
        # tangent = derivative of the residual
        h = 1e-7
        for i in (0, nd // 2, nd - 1):
            up, um = ue.copy(), ue.copy()
            up[i] += h; um[i] -= h
            fd = (oracle.element_hyperelastic(m, 0, up, want_K=False)[1] - oracle.element_hyperelastic(m, 0, um, want_K=False)[1]) / (2 * h)
            np.testing.assert_allclose(K1[:, i], fd, atol=1e-6 * np.abs(K1).max())


def test_threaded_hyperelastic_assembly(oracle):
    xyz, conn = oracle.generate_grid_hex(3, 2, 2, (0, 0, 0), (1, 1, 1))
    cd, nd = oracle.close_dofs(oracle.HEX8, 3, conn, len(xyz))
    rp, ci = oracle.build_pattern(cd, nd)
    m = oracle.Mesh(oracle.HEX8, 2, xyz, conn, cd)
    u = 1e-2 * np.random.default_rng(0).uniform(-1, 1, nd)
    K, r = oracle.assemble_hyperelastic(m, u, rp, ci)
    col, nc = oracle.color_cells(cd, nd)
    K2, r2 = oracle.assemble_hyperelastic(m, u, rp, ci, nthreads=4, color=col, ncolors=nc)
    np.testing.assert_allclose(K2, K, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(r2, r, rtol=1e-12, atol=1e-16)


# ------------------------------------------------------------------------------------------- weak boundary conditions
def _one_hex(oracle, order, distort=0.0, seed=0):
    rng = np.random.default_rng(seed)
    xyz, conn = oracle.generate_grid_hex(1, 1, 1, (0, 0, 0), (1.0, 1.0, 1.0))
    xyz = xyz + distort * rng.uniform(-1, 1, xyz.shape)
    kind, q = (oracle.HEX8, 2) if order == 1 else (oracle.HEX27, 3)
    cd, nd = oracle.close_dofs(kind, 3, conn, len(xyz))
    return oracle.Mesh(kind, q, xyz, conn, cd), nd


@pytest.mark.parametrize("order", [1, 2])
def test_facet_terms_closed_forms(oracle, order):
    """Closed forms on the unit cube (weak_boundary_conditions.jl): Robin r = 2α·M_Γ·u and K = 2α·M_Γ (Σ = 2α·area per
    component); pressure at u = 0 pushes with p·area·n₀ in total; on a closed surface the follower load has no resultant."""
    m, nd = _one_hex(oracle, order)
    fq = order + 1
    normals = np.array([[0, 0, -1], [0, -1, 0], [1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, 0, 1]], dtype=float)
    u0 = np.zeros(nd)
    for lf in range(6):
        Ke, re = oracle.element_facet(m, 0, lf, oracle.BC_ROBIN, 0.7, fq, np.tile([1.0, -2.0, 0.5], nd // 3))
        assert Ke.sum() == pytest.approx(3 * 2 * 0.7 * 1.0, rel=1e-13)
        np.testing.assert_allclose(re.reshape(-1, 3).sum(axis=0), 2 * 0.7 * np.array([1.0, -2.0, 0.5]), rtol=1e-13)
        np.testing.assert_allclose(Ke, Ke.T, atol=1e-15)
        _, rp = oracle.element_facet(m, 0, lf, oracle.BC_PRESSURE, 3.0, fq, u0)
        np.testing.assert_allclose(rp.reshape(-1, 3).sum(axis=0), 3.0 * normals[lf], atol=1e-14)
        Kn, rn = oracle.element_facet(m, 0, lf, oracle.BC_NORMAL_SPRING, 2.0, fq, np.tile([1.0, -2.0, 0.5], nd // 3))
        un = np.array([1.0, -2.0, 0.5]) @ normals[lf]
        np.testing.assert_allclose(rn.reshape(-1, 3).sum(axis=0), 2.0 * un * normals[lf], atol=1e-13)
    rng = np.random.default_rng(1)
    md, nd = _one_hex(oracle, order, distort=0.1, seed=3)
    u = rng.uniform(-0.05, 0.05, nd)
    tot = np.zeros(3)
    for lf in range(6):
        _, rp = oracle.element_facet(md, 0, lf, oracle.BC_PRESSURE, 1.3, fq + 1 if fq < 3 else 3, u)
        tot += rp.reshape(-1, 3).sum(axis=0)
    if order == 1:  # exact only when the rule integrates the (tri-quadratic) integrand: Q1 with 3 points
        assert np.abs(tot).max() < 1e-13


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("kind", ["BC_ROBIN", "BC_NORMAL_SPRING", "BC_PRESSURE", "BC_BENDING_SPRING", "BC_PRESSURE_FIELD"])
def test_facet_tangent_is_derivative_of_residual(oracle, order, kind):
    m, nd = _one_hex(oracle, order, distort=0.12, seed=5)
    rng = np.random.default_rng(2)
    u = rng.uniform(-0.05, 0.05, nd)
    k = getattr(oracle, kind)
    oracle.set_facet_pressure_field(rng.uniform(0.5, 1.5, (1, 8)) if kind == "BC_PRESSURE_FIELD" else None)
    for lf in (0, 3, 4):
        Ke, _ = oracle.element_facet(m, 0, lf, k, 1.7, order + 1, u)
        fd = np.zeros_like(Ke)
        h = 1e-6
        for j in range(nd):
            e = np.zeros(nd); e[j] = h
            _, rp = oracle.element_facet(m, 0, lf, k, 1.7, order + 1, u + e, want_K=False)
            _, rm = oracle.element_facet(m, 0, lf, k, 1.7, order + 1, u - e, want_K=False)
            fd[:, j] = (rp - rm) / (2 * h)
        assert np.abs(Ke - fd).max() < 1e-8 * max(1.0, np.abs(Ke).max())
    oracle.set_facet_pressure_field(None)


def test_active_stress_material_routine_matches_ad_oracle(tb, oracle):
    """ActiveStressModel + SimpleActiveStress (materials.jl:1200-1266, active.jl:100-113): the device routine's P and 𝔸
    (evaluated on the host through the same inline code) against the hyper-dual derivative of Ψ_passive + Ta·‖F f₀‖, and the
    stress increment against its closed form Ta·(F f₀)⊗f₀/‖F f₀‖."""
    rng = np.random.default_rng(8)
    f, s, n = np.linalg.qr(rng.normal(size=(3, 3)))[0].T
    for Ta in (0.0, 0.7, 25.0):
        for _ in range(4):
            F = np.eye(3) + 0.15 * rng.normal(size=(3, 3))
            model = tb.ActiveStressModel(tb.HolzapfelOgden2009Model(), tb.SimpleActiveStress(Tmax=Ta), tb.CaDrivenInternalSarcomereModel(tb.PelceSunLangeveld1995Model(), 1.0),
                                         tb.ConstantCoefficient(tb.OrthotropicMicrostructure(f, s, n)))
            psi, P, A = tb.material_routine(model, F)
            oracle.set_active_tension(0.0)
            psi0, P0, A0 = oracle.ho_energy(F, fsn=np.stack([f, s, n]))
            Ff = F @ f
            lam = np.linalg.norm(Ff)
            np.testing.assert_allclose(P - P0, Ta * np.outer(Ff, f) / lam, rtol=1e-11, atol=1e-12 * np.abs(P).max())
            # second derivative of Ta·‖F f₀‖ by its closed form
            dA = Ta * (np.einsum("ik,j,l->ijkl", np.eye(3), f, f) / lam - np.einsum("i,j,k,l->ijkl", Ff, f, Ff, f) / lam ** 3)
            np.testing.assert_allclose((A - A0).reshape(3, 3, 3, 3), dA, rtol=1e-10, atol=1e-12 * np.abs(A).max())
            assert psi - psi0 == pytest.approx(Ta * lam, rel=1e-11, abs=1e-12 * max(1.0, abs(psi)))
    # PelceSunLangeveld1995: compute_λᵃ (contraction.jl:307-311) — used by the active *strain* models, 𝓝 ignores it
    m = tb.PelceSunLangeveld1995Model()
    assert m.compute_lambda_a(0.0) == 1.0 and m.compute_lambda_a(1.0) == pytest.approx(1.0 / (1.0 + 0.5 * (1 / 0.7 - 1)))


def test_active_stress_element_through_the_oracle(oracle):
    """The oracle's element routine with an active tension: K = ∂r/∂u by central differences, nodal calcium interpolated."""
    m, nd = _one_hex(oracle, 1, distort=0.1, seed=2)
    rng = np.random.default_rng(3)
    u = rng.uniform(-0.03, 0.03, nd)
    ca = rng.uniform(0.2, 1.0, (1, 8))
    oracle.set_active_tension(3.0, ca)
    try:
        Ke, re = oracle.element_hyperelastic(m, 0, u)
        fd = np.zeros_like(Ke)
        for j in range(nd):
            e = np.zeros(nd); e[j] = 1e-6
            fd[:, j] = (oracle.element_hyperelastic(m, 0, u + e, want_K=False)[1] - oracle.element_hyperelastic(m, 0, u - e, want_K=False)[1]) / 2e-6
        assert np.abs(Ke - fd).max() < 1e-7 * np.abs(Ke).max()
        oracle.set_active_tension(0.0)
        K0, r0 = oracle.element_hyperelastic(m, 0, u)
        assert np.abs(re - r0).max() > 1e-3
    finally:
        oracle.set_active_tension(0.0)


ENERGY_CASES = [
    ("NullEnergyModel", {}, "EN_NULL"),
    ("BioNeoHookean", {}, "EN_BIO_NEOHOOKEAN"),
    ("TransverseIsotopicNeoHookeanModel", {}, "EN_TI_NEOHOOKEAN"),
    ("LinYinPassiveModel", {}, "EN_LIN_YIN_PASSIVE"),
    ("LinYinActiveModel", {}, "EN_LIN_YIN_ACTIVE"),
    ("HumphreyStrumpfYinModel", {}, "EN_HSY"),
    ("LinearSpringModel", {}, "EN_LINEAR_SPRING"),
    ("Guccione1991PassiveModel", {}, "EN_GUCCIONE"),
]
PENALTIES = ["SimpleCompressionPenalty", "NullCompressionPenalty", "HartmannNeffCompressionPenalty1", "HartmannNeffCompressionPenalty2",
             "HartmannNeffCompressionPenalty3"]


@pytest.mark.parametrize("cls,kw,en", ENERGY_CASES)
def test_every_reference_energy_matches_the_ad_oracle(tb, oracle, cls, kw, en):
    """All passive energies of src/modeling/solid/energies.jl, with the reference's default parameters and default penalty: the
    device routine (hyper-dual evaluation per component pair, run on the host through the same inline code) against the oracle's
    hyper-dual Hessian; Ψ(I) = 0 (or the constant the formula gives) and P(I) = 0 as test/test_type_stability.jl:29-63 checks."""
    rng = np.random.default_rng(11)
    f, s, n = np.linalg.qr(rng.normal(size=(3, 3)))[0].T
    mat = getattr(tb, cls)(**kw)
    model = tb.PK1Model(mat, tb.ConstantCoefficient(tb.OrthotropicMicrostructure(f, s, n)))
    eid = getattr(oracle, en)
    for trial in range(4):
        F = np.eye(3) + 0.12 * rng.normal(size=(3, 3))
        psi, P, A = tb.material_routine(model, F)
        rpsi, rP, rA = oracle.energy(eid, mat.mpU.pid, mat.p, mat.mpU.u, F, fsn=np.stack([f, s, n]))
        sc = max(1.0, np.abs(rA).max())
        assert psi == pytest.approx(rpsi, rel=1e-12, abs=1e-13)
        np.testing.assert_allclose(P, rP, rtol=1e-11, atol=1e-12 * sc)
        np.testing.assert_allclose(A, rA, rtol=1e-10, atol=1e-12 * sc)
        np.testing.assert_allclose(A, A.T, atol=1e-12 * sc)
    psiI, PI, _ = tb.material_routine(model, np.eye(3))
    if cls != "Guccione1991PassiveModel":          # Ψ(I) = C₀/2 there: the formula has no −1
        assert abs(psiI) < 1e-14
    if cls != "LinYinActiveModel":                 # an *active* energy: its linear terms C₄(I₁−3) + C₅(I₄−1) pre-stress the reference state
        assert np.abs(PI).max() < 1e-12


@pytest.mark.parametrize("pen", PENALTIES)
def test_compression_penalties_with_holzapfel_ogden(tb, oracle, pen):
    """HolzapfelOgden2009Model(mpU = each penalty of energies.jl:13-87): the Simple penalty takes the hand-derived routines, the
    others the device AD path; both against the hyper-dual oracle."""
    rng = np.random.default_rng(12)
    f, s, n = np.linalg.qr(rng.normal(size=(3, 3)))[0].T
    mpU = getattr(tb, pen)()
    if pen == "HartmannNeffCompressionPenalty2":
        mpU = tb.HartmannNeffCompressionPenalty2(a=2.0)   # the default a = 1.1 has an infinite second derivative at I₃ = 1 and NaNs under compression
    model = tb.PK1Model(tb.HolzapfelOgden2009Model(mpU=mpU), tb.ConstantCoefficient(tb.OrthotropicMicrostructure(f, s, n)))
    for trial in range(3):
        F = np.eye(3) + 0.1 * rng.normal(size=(3, 3))
        F *= (1.2 / np.linalg.det(F)) ** (1 / 3)                 # J = 1.2 > 1 so that (√I₃ − 1)ᵃ is defined
        psi, P, A = tb.material_routine(model, F)
        rpsi, rP, rA = oracle.energy(oracle.EN_HO, mpU.pid, oracle.HO_DEFAULTS[:8], mpU.u, F, fsn=np.stack([f, s, n]))
        np.testing.assert_allclose(P, rP, rtol=1e-11, atol=1e-12 * np.abs(rA).max())
        np.testing.assert_allclose(A, rA, rtol=1e-10, atol=1e-12 * np.abs(rA).max())
        assert psi == pytest.approx(rpsi, rel=1e-12)


def _hill_cases(tb):
    """The Hill-type material set-ups of the reference's contractile-cuboid integration test
    (test/integration/test_solid_mechanics.jl:300-360), plus the remaining spring / deformation-gradient / sarcomere variants."""
    psl = lambda ca: tb.CaDrivenInternalSarcomereModel(tb.PelceSunLangeveld1995Model(), ca)
    return [
        ("ExtendedHillModel", tb.HolzapfelOgden2009Model(), tb.ActiveMaterialAdapter(tb.LinearSpringModel()), tb.GMKActiveDeformationGradientModel(), psl),
        ("GeneralizedHillModel", tb.LinYinPassiveModel(), tb.ActiveMaterialAdapter(tb.LinYinActiveModel()), tb.GMKIncompressibleActiveDeformationGradientModel(), psl),
        ("GeneralizedHillModel", tb.HumphreyStrumpfYinModel(), tb.ActiveMaterialAdapter(tb.LinearSpringModel()), tb.RLRSQActiveDeformationGradientModel(0.75), psl),
        ("ExtendedHillModel", tb.HolzapfelOgden2009Model(), tb.SimpleActiveSpring(), tb.RLRSQActiveDeformationGradientModel(0.25), psl),
        ("GeneralizedHillModel", tb.BioNeoHookean(), tb.SimpleActiveSpring(2.0), tb.GMKActiveDeformationGradientModel(),
         lambda ca: tb.CaDrivenInternalSarcomereModel(tb.ConstantStretchModel(0.9), ca)),
    ]


def _oracle_hill(oracle, tb, model, ca):
    h = model.lower_hill()
    oracle.set_hill(h.framework, h.active_energy, h.active_penalty, list(h.active_p), h.adg_kind, h.sheetlet_part, h.sarcomere_kind, list(h.sarcomere_p))
    oracle.set_point_activation(ca)


@pytest.mark.parametrize("case", range(5))
@pytest.mark.parametrize("ca", [0.0, 0.35, 1.0])
def test_hill_frameworks_match_the_ad_oracle(tb, oracle, case, ca):
    """GeneralizedHillModel / ExtendedHillModel (materials.jl:1042-1190) over every active deformation gradient model (active.jl:23-96)
    and both steady-state sarcomere models: device routine (run on the host) against the oracle, which forms Fᵃ, inverts it numerically
    and rotates the frame as ActiveMaterialAdapter does instead of using the closed forms of the kernels."""
    rng = np.random.default_rng(21 + case)
    f, s, n = np.linalg.qr(rng.normal(size=(3, 3)))[0].T
    fw, passive, spring, adg, sarc = _hill_cases(tb)[case]
    model = getattr(tb, fw)(passive, spring, adg, sarc(ca), tb.ConstantCoefficient(tb.OrthotropicMicrostructure(f, s, n)))
    pm = tb.PK1Model(passive, tb.ConstantCoefficient(tb.OrthotropicMicrostructure(f, s, n))).lower()
    try:
        _oracle_hill(oracle, tb, model, ca)
        for trial in range(3):
            F = np.eye(3) + 0.1 * rng.normal(size=(3, 3))
            psi, P, A = tb.material_routine(model, F)
            rpsi, rP, rA = oracle.energy(pm.kind, pm.reserved, list(pm.p)[:9], list(pm.p)[10:13], F, fsn=np.stack([f, s, n]))
            sc = max(1.0, np.abs(rA).max())
            assert psi == pytest.approx(rpsi, rel=1e-12, abs=1e-13)
            np.testing.assert_allclose(P, rP, rtol=1e-11, atol=1e-12 * sc)
            np.testing.assert_allclose(A, rA, rtol=1e-10, atol=1e-12 * sc)
    finally:
        oracle.set_hill(); oracle.set_point_activation(0.0)


def test_hill_without_calcium_reduces_to_the_springs_at_rest(tb, oracle):
    """Ca = 0 ⇒ λᵃ = 1 ⇒ Fᵃ = I (contraction.jl:302-311): the generalized model is Ψᵖ(F) + Ψᵃ(F); the extended one (𝓝 = Ca = 0) is
    the passive spring alone.  With Ca > 0 and the PSL1995 defaults the fibre shortens: P(I)·f₀·f₀ > 0 pulls along f₀."""
    f, s, n = np.eye(3)
    ms = tb.ConstantCoefficient(tb.OrthotropicMicrostructure(f, s, n))
    rng = np.random.default_rng(5)
    F = np.eye(3) + 0.1 * rng.normal(size=(3, 3))
    sarc = tb.CaDrivenInternalSarcomereModel(tb.PelceSunLangeveld1995Model(), 0.0)
    ext = tb.ExtendedHillModel(tb.HolzapfelOgden2009Model(), tb.ActiveMaterialAdapter(tb.LinearSpringModel()), tb.GMKActiveDeformationGradientModel(), sarc, ms)
    gen = tb.GeneralizedHillModel(tb.HolzapfelOgden2009Model(), tb.ActiveMaterialAdapter(tb.LinearSpringModel()), tb.GMKActiveDeformationGradientModel(), sarc, ms)
    p0 = tb.material_routine(tb.PK1Model(tb.HolzapfelOgden2009Model(), ms), F)
    pa = tb.material_routine(tb.PK1Model(tb.LinearSpringModel(), ms), F)
    pe, pg = tb.material_routine(ext, F), tb.material_routine(gen, F)
    np.testing.assert_allclose(pe[1], p0[1], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(pg[1], p0[1] + pa[1], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(pg[2], p0[2] + pa[2], rtol=1e-12, atol=1e-13)
    active = tb.GeneralizedHillModel(tb.HolzapfelOgden2009Model(), tb.ActiveMaterialAdapter(tb.LinearSpringModel()), tb.GMKActiveDeformationGradientModel(),
                                     tb.CaDrivenInternalSarcomereModel(tb.PelceSunLangeveld1995Model(), 1.0), ms)
    PI = tb.material_routine(active, np.eye(3))[1]
    assert PI[0, 0] > 1e-3 and abs(PI[1, 1]) < 1e-12 and abs(PI[0, 1]) < 1e-12
