"""RDQ20-MF sarcomere model: the oracle against the reference's golden trajectory (test/test_sarcomere.jl:7-115,
test/data/trajectories/RDQ20-MF/transient-test.csv — copied as a fixture), the host evaluation of the device code against the
oracle, and (gpu) the device kernel against both."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSV = os.path.join(HERE, "golden", "trajectories", "RDQ20-MF", "transient-test.csv")


def protocol():
    """Inputs of the reference test: calcium and sarcomere-length transients, dt = 1e-3 ms, the backward-difference velocity."""
    with open(CSV) as f:
        header = f.readline().strip().split(",")
    data = np.loadtxt(CSV, delimiter=",", skiprows=1)
    col = {n: i for i, n in enumerate(header)}
    ts_data = 1000.0 * data[:, col["t"]]                       # s → ms
    dt = 1e-3
    n_steps = int(np.floor(ts_data[-1] / dt + 1e-9)) + 1       # τ = 0:dt:Tmax
    t = np.arange(n_steps) * dt
    c0, cmax, tau1, tau2, t0 = 0.1, 0.9, 20.0, 50.0, 10.0
    beta = (tau1 / tau2) ** (-1 / (tau1 / tau2 - 1)) - (tau1 / tau2) ** (-1 / (1 - tau2 / tau1))
    ca = np.where(t < t0, c0, c0 + (cmax - c0) / beta * (np.exp(-(t - t0) / tau1) - np.exp(-(t - t0) / tau2)))
    SL0, SLt0, SLt1, SLtau0, SLtau1 = 2.2, 50.0, 350.0, 50.0, 20.0
    SL1 = SL0 * 0.97

    def stretch(x):
        return (SL0 + (SL1 - SL0) * (np.maximum(0.0, 1.0 - np.exp((SLt0 - x) / SLtau0)) - np.maximum(0.0, 1.0 - np.exp((SLt1 - x) / SLtau1)))) / SL0
    lam = stretch(t)
    vel = (stretch(t) - stretch(t - dt)) / dt
    # sample i ↔ the data row with t_i − dt/2 ≤ t_ref < t_i + dt/2 (findfirst in the reference)
    sample = np.zeros(n_steps, dtype=np.uint8)
    rows = []
    for r, tr in enumerate(ts_data):
        i = int(np.floor((tr + dt / 2) / dt))
        if 0 <= i < n_steps and t[i] - dt / 2 <= tr < t[i] + dt / 2:
            sample[i] = 1
            rows.append(r)
    return dict(dt=dt, t=t, ca=ca, lam=lam, vel=vel, sample=sample, rows=np.array(rows), data=data, col=col)


def isapprox(x, y, rtol):
    """Base.isapprox(x, y; rtol) with atol = 0: norm(x − y) ≤ rtol·max(norm(x), norm(y))"""
    x, y = np.asarray(x, dtype=float), np.asarray(y, dtype=float)
    return np.linalg.norm((x - y).ravel()) <= rtol * max(np.linalg.norm(x.ravel()), np.linalg.norm(y.ravel()))


def check_against_golden(states, pr, tension, stiffness, SL0=2.2):
    """The assertions of test/test_sarcomere.jl:75-111 (rtol 1e-3 on states, Ta, As; 1e-2 on the derived inputs)."""
    data, col, rows = pr["data"], pr["col"], pr["rows"]
    idx = np.flatnonzero(pr["sample"])
    S0 = col["S0"]
    assert len(rows) == len(idx) == states.shape[0] and len(rows) >= 60
    for k, (r, i) in enumerate(zip(rows, idx)):
        assert isapprox(pr["ca"][i], data[r, col["Ca"]], 1e-3)
        uref = data[r, S0:S0 + 20]
        # urefRU = permutedims(reshape(uref[1:16], (2,2,2,2)), (4,3,2,1)) in column-major = a plain C-order transpose of axes
        urefRU = np.transpose(uref[:16].reshape((2, 2, 2, 2), order="F"), (3, 2, 1, 0))
        uRU = states[k, :16].reshape((2, 2, 2, 2), order="F")
        assert isapprox(uRU, urefRU, 1e-3), (r, uRU, urefRU)              # Julia's ≈ on arrays: ‖x − y‖₂ ≤ rtol·max(‖x‖₂, ‖y‖₂)
        for j in range(16, 20):
            assert isapprox(states[k, j], uref[j], 1e-3), (r, j, states[k, j], uref[j])
        assert isapprox(1000.0 * pr["vel"][i] * SL0, data[r, col["dSL_dt"]], 1e-2)
        assert isapprox(pr["lam"][i] * SL0, data[r, col["SL"]], 1e-2)
        assert isapprox(tension[k], data[r, col["Ta"]], 1e-3), (r, tension[k], data[r, col["Ta"]])
        assert isapprox(stiffness[k], data[r, col["As"]], 1e-3), (r, stiffness[k], data[r, col["As"]])


def test_oracle_reproduces_the_reference_golden_trajectory(oracle):
    """PINNED: 600 ms of forward Euler at dt = 1 µs against all 61 rows of the original solution."""
    pr = protocol()
    u0 = np.zeros(20); u0[0] = 1.0
    _, states = oracle.rdq20mf_trajectory(u0, pr["dt"], pr["lam"], pr["vel"], pr["ca"], pr["sample"])
    idx = np.flatnonzero(pr["sample"])
    Ta = [oracle.rdq20mf_tension(s, pr["lam"][i]) for s, i in zip(states, idx)]
    As = [oracle.rdq20mf_stiffness(s, pr["lam"][i]) for s, i in zip(states, idx)]
    check_against_golden(states, pr, Ta, As)
    # the chain is conservative: Σ occupancies stays 1 (contraction.jl:585-596)
    np.testing.assert_allclose(states[:, :16].sum(axis=1), 1.0, rtol=0, atol=1e-10)
    assert states[:, :16].min() >= 0.0


def test_host_evaluation_of_the_device_code_matches_the_oracle(tb, oracle):
    rng = np.random.default_rng(0)
    p = oracle.RDQ20MF_DEFAULTS
    for trial in range(50):
        u = np.concatenate([rng.dirichlet(np.ones(16)), rng.uniform(0, 0.2, 4)])
        if trial == 0:
            u = np.zeros(20); u[0] = 1.0                       # default initial state: several zero-probability guards are taken
        lam, vel, ca = rng.uniform(0.5, 1.4), rng.normal() * 5e-3, rng.uniform(0.05, 1.2)
        if trial % 7 == 0:
            vel = 0.0                                          # smooth_abs at 0
        du, Ta, As = tb.sarcomere_rhs(tb.RDQ20MFModel(), u, lam, vel, ca)
        np.testing.assert_allclose(du, oracle.rdq20mf_rhs(u, lam, vel, ca), rtol=1e-13, atol=1e-16)
        assert Ta == pytest.approx(oracle.rdq20mf_tension(u, lam), rel=1e-14, abs=0)
        assert As == pytest.approx(oracle.rdq20mf_stiffness(u, lam), rel=1e-14, abs=0)
        assert abs(du[:16].sum()) < 1e-15                      # Σ dQ[1:16] = 0 to machine precision
    # every branch of fraction_single_overlap, parameters other than the defaults
    m = tb.RDQ20MFModel(LA=1.2, a_XB=1.0e3, gamma=9.0, mu=7.0, Q=3.0)
    u = np.concatenate([rng.dirichlet(np.ones(16)), rng.uniform(0, 0.2, 4)])
    for SL in (1.0, 1.3, 1.9, 2.4, 2.7, 4.5):
        lam = SL / m.SL0
        du, Ta, As = tb.sarcomere_rhs(m, u, lam, 1e-3, 0.4)
        np.testing.assert_allclose(du, oracle.rdq20mf_rhs(u, lam, 1e-3, 0.4, p=m.params()), rtol=1e-13, atol=1e-16)
        assert Ta == pytest.approx(oracle.rdq20mf_tension(u, lam, p=m.params()), rel=1e-14, abs=0)


def test_as_rate_independent(tb):
    """test/test_sarcomere.jl:117-147: the wrapper is the model at zero shortening velocity, and that is not a no-op."""
    inner = tb.RDQ20MFModel()
    wrapped = tb.AsRateIndependent(inner)
    u = np.zeros(20); u[0] = 1.0; u[16:] = 0.1
    lam, ca = 0.97, 0.5
    duref = tb.sarcomere_rhs(inner, u, lam, 0.0, ca)[0]
    for v in (0.0, -1.0e-3, 5.0e-3):
        np.testing.assert_allclose(tb.sarcomere_rhs(wrapped, u, lam, v, ca)[0], duref, rtol=0, atol=0)
    assert not np.allclose(tb.sarcomere_rhs(inner, u, lam, -1.0e-3, ca)[0], duref)
    assert tb.num_states(wrapped) == tb.num_states(inner) == 20
    assert tb.compute_active_tension(wrapped, u, lam) == tb.compute_active_tension(inner, u, lam)
    assert tb.compute_active_stiffness(wrapped, u, lam) == tb.compute_active_stiffness(inner, u, lam)
    bad = u.copy(); bad[2] = -1.0e-3
    assert tb.internal_state_in_bounds(inner, u) and not tb.internal_state_in_bounds(inner, bad) and not tb.internal_state_in_bounds(wrapped, bad)


@pytest.mark.gpu
def test_device_trajectory_matches_oracle_and_golden(tb, oracle, device):
    """The reference protocol on the device, step for step (600 000 launches with scalar inputs, 64 identical points): the sampled
    states agree with the oracle to 1e-11 and satisfy the reference's own assertions against the golden data."""
    pr = protocol()
    u0 = np.zeros(20); u0[0] = 1.0
    _, ref_states = oracle.rdq20mf_trajectory(u0, pr["dt"], pr["lam"], pr["vel"], pr["ca"], pr["sample"])
    model = tb.RDQ20MFModel()
    npts = 64
    st = tb.SarcomereState(device, model, npts)
    Ta, As = device.zeros(npts), device.zeros(npts)
    step = tb.sarcomere_stepper(st, pr["dt"], tension=Ta, stiffness=As)          # bound ctypes call: the loop below is launch-rate bound
    got, gTa, gAs = [], [], []
    lam, vel, ca, sample = pr["lam"].tolist(), pr["vel"].tolist(), pr["ca"].tolist(), pr["sample"].tolist()
    for i in range(len(lam)):
        step(lam[i], vel[i], ca[i])
        if sample[i]:
            s = st.to_host()
            assert np.abs(s - s[:, :1]).max() == 0.0                             # every point saw the same inputs
            got.append(s[:, 0]); gTa.append(Ta.to_host()[0]); gAs.append(As.to_host()[0])
    got = np.array(got)
    np.testing.assert_allclose(got, ref_states, rtol=1e-11, atol=1e-15)
    check_against_golden(got, pr, gTa, gAs)


@pytest.mark.gpu
def test_device_step_with_per_point_inputs(tb, oracle, device):
    """Per-point stretch / velocity / calcium arrays, ragged point counts, outputs; one and several substeps."""
    rng = np.random.default_rng(1)
    model = tb.RDQ20MFModel()
    p = model.params()
    for npts in (1, 63, 1000, 4099):
        u = np.concatenate([rng.dirichlet(np.ones(16), npts).T, rng.uniform(0, 0.2, (4, npts))])      # [state][point]
        lam, vel, ca = rng.uniform(0.8, 1.15, npts), rng.normal(size=npts) * 3e-3, rng.uniform(0.05, 1.0, npts)   # Kd(SL) has a pole at SL ≈ 2.82 µm
        for sub in (1, 5):
            st = tb.SarcomereState(device, model, npts, initial=u)
            Ta, As = device.zeros(npts), device.zeros(npts)
            tb.sarcomere_step(st, 0.0, 0.01, stretch=device.to_device(lam), velocity=device.to_device(vel), calcium=device.to_device(ca),
                              substeps=sub, tension=Ta, stiffness=As)
            ref = u.copy()
            for i in range(npts):
                x = ref[:, i].copy()
                for s in range(sub):
                    x = x + 0.01 * oracle.rdq20mf_rhs(x, lam[i], vel[i], ca[i])
                ref[:, i] = x
            got = st.to_host()
            np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-13)   # cancellation in the flux differences: absolute scale of the occupancies is 1
            np.testing.assert_allclose(Ta.to_host(), [oracle.rdq20mf_tension(ref[:, i], lam[i]) for i in range(npts)], rtol=1e-12)
            np.testing.assert_allclose(As.to_host(), [oracle.rdq20mf_stiffness(ref[:, i], lam[i]) for i in range(npts)], rtol=1e-12)
    # zero points: a no-op
    st = tb.SarcomereState(device, model, 0)
    tb.sarcomere_step(st, 0.0, 0.01, stretch=1.0, velocity=0.0, calcium=0.1)


# ------------------------------------------------------------------------------------------- the local problem of the condensed mechanics
def random_states(rng, n):
    return np.concatenate([rng.dirichlet(np.ones(16), n).T, rng.uniform(0, 0.2, (4, n))])


def test_local_solve_host_matches_oracle_and_is_a_backward_euler_step(tb, oracle):
    """solve_internal_timestep (materials.jl:1403-1497) + corrector (:1556-1568): the host version of the device algebra against the
    oracle's 21-partial forward mode + LU; the solution satisfies the backward Euler equation; the corrector is dQ/dλ (central
    difference of the solve itself); the reference's defaults tol = 1e-4, max_iters = 10 and the default initial state."""
    rng = np.random.default_rng(2)
    model = tb.RDQ20MFModel()
    tight = tb.GenericLocalNonlinearSolver(max_iters=30, tol=1e-13)
    for trial in range(30):
        Qk = random_states(rng, 1)[:, 0]
        if trial == 0:
            Qk = np.zeros(20); Qk[0] = 1.0
        lam, ca, dt = rng.uniform(0.8, 1.15), rng.uniform(0.05, 1.0), rng.choice([0.01, 0.25, 1.0, 5.0])
        for ls in (None, tight):
            st, Q, dQ, it, rn = tb.sarcomere_local_solve(model, Qk, Qk, lam, ca, dt, ls)
            ost, oQ, odQ, oit, orn = oracle.rdq20mf_local_solve(Qk, Qk, lam, ca, dt, *( (1e-4, 10) if ls is None else (ls.tol, ls.max_iters)))
            assert st == ost == 0 and it == oit, (trial, st, ost, it, oit)
            np.testing.assert_allclose(Q, oQ, rtol=1e-12, atol=1e-15)
            np.testing.assert_allclose(dQ, odQ, rtol=1e-9, atol=1e-13)
        # backward Euler: (Q − Qk)/dt = rhs(Q) at the tight solution
        res = (Q - Qk) / dt - tb.sarcomere_rhs(model, Q, lam, 0.0, ca)[0]
        assert np.abs(res).max() < 1e-12
        assert abs(Q[:16].sum() - 1.0) < 1e-12                              # backward Euler inherits the conservation
        h = 1e-6
        Qp = tb.sarcomere_local_solve(model, Qk, Qk, lam + h, ca, dt, tight)[1]
        Qm = tb.sarcomere_local_solve(model, Qk, Qk, lam - h, ca, dt, tight)[1]
        np.testing.assert_allclose(dQ, (Qp - Qm) / (2 * h), rtol=2e-6, atol=1e-9)


def test_local_solve_failure_reports(tb, oracle):
    """LocalSolveReport retcodes (multilevel_newton_raphson.jl, materials.jl:1453-1497): MaxIters when the iteration limit ends the
    loop, Infeasible for a converged state with a negative occupancy (a step far too long for the chain's dynamics)."""
    model = tb.RDQ20MFModel()
    Qk = np.zeros(20); Qk[0] = 1.0
    st, *_ = tb.sarcomere_local_solve(model, Qk, Qk, 1.0, 0.8, 1.0, tb.GenericLocalNonlinearSolver(max_iters=1, tol=1e-14))
    assert st == 2 == oracle.rdq20mf_local_solve(Qk, Qk, 1.0, 0.8, 1.0, 1e-14, 1)[0]
    rng = np.random.default_rng(3)
    found = False
    for trial in range(200):                                                 # negative occupancies appear for Δt ≳ 5 (contraction.jl:592-594)
        Qg = random_states(rng, 1)[:, 0]
        st, Q, *_ = tb.sarcomere_local_solve(model, Qg, Qg, rng.uniform(0.8, 1.15), rng.uniform(0.05, 1.0), 200.0, tb.GenericLocalNonlinearSolver(30, 1e-10))
        ost = oracle.rdq20mf_local_solve(Qg, Qg, 1.0, 0.5, 200.0, 1e-10, 30)[0]
        if st == 4:
            assert Q[:16].min() < 0
            found = True
            break
    assert found or ost in (0, 4)


@pytest.mark.gpu
def test_device_local_solve_matches_oracle(tb, oracle, device):
    """The 16-lanes-per-point Newton + Gauss–Jordan kernel against the oracle: states, corrector, status, over ragged point counts
    (groups of 4 points per wave, 16 per block), with per-point and scalar inputs, reference defaults and a tight tolerance."""
    rng = np.random.default_rng(4)
    model = tb.RDQ20MFModel()
    for npts in (1, 5, 16, 67, 1031):
        Qk = random_states(rng, npts)
        Qk[:, 0] = 0.0; Qk[0, 0] = 1.0                                       # default initial state at point 0
        lam, ca = rng.uniform(0.8, 1.15, npts), rng.uniform(0.05, 1.0, npts)
        for ls, dt in ((None, 0.25), (tb.GenericLocalNonlinearSolver(30, 1e-13), 1.0)):
            known = tb.SarcomereState(device, model, npts, initial=Qk)
            st = tb.SarcomereState(device, model, npts, initial=Qk)
            dQ = device.zeros(20 * npts)
            status = device.to_device(np.full(npts, -1, dtype=np.int32))
            nf = tb.sarcomere_implicit_step(st, known, dt, device.to_device(lam), device.to_device(ca), ls, dstate_dstretch=dQ, status=status)
            got, gdQ = st.to_host(), dQ.to_host().reshape(20, npts)
            tol, mi = (1e-4, 10) if ls is None else (ls.tol, ls.max_iters)
            codes = []
            for i in range(npts):
                code, oQ, odQ, _, _ = oracle.rdq20mf_local_solve(Qk[:, i], Qk[:, i], lam[i], ca[i], dt, tol, mi)
                codes.append(code)
                np.testing.assert_allclose(got[:, i], oQ, rtol=1e-11, atol=1e-14)
                np.testing.assert_allclose(gdQ[:, i], odQ, rtol=1e-8, atol=1e-12)
            assert nf == sum(c != 0 for c in codes) == 0
            assert np.array_equal(status.to_host(), np.array(codes, dtype=np.int32))
    # scalar inputs; no sensitivities requested
    known = tb.SarcomereState(device, model, 40)
    st = tb.SarcomereState(device, model, 40)
    assert tb.sarcomere_implicit_step(st, known, 0.5, 1.02, 0.6) == 0
    code, oQ, *_ = oracle.rdq20mf_local_solve(known.to_host()[:, 0], known.to_host()[:, 0], 1.02, 0.6, 0.5)
    np.testing.assert_allclose(st.to_host()[:, 7], oQ, rtol=1e-11, atol=1e-14)
    # the iteration limit is reported per point and counted
    st = tb.SarcomereState(device, model, 40)
    status = device.to_device(np.zeros(40, dtype=np.int32))
    assert tb.sarcomere_implicit_step(st, known, 1.0, 1.0, 0.8, tb.GenericLocalNonlinearSolver(1, 1e-14), status=status) == 40
    assert (status.to_host() == 2).all()


def test_rate_coupled_local_solve_host_matches_oracle(tb, oracle):
    """The rate-coupled local problem dₜQ = L(F, dₜF, Q) (materials.jl:1664-1750): dλ/dt enters the cross-bridge block and a second
    corrector dQ/d(dλ/dt) appears; host version of the device algebra against the oracle's 22-partial forward mode, and both
    correctors against central differences of the solve."""
    rng = np.random.default_rng(5)
    model = tb.RDQ20MFModel()
    tight = tb.GenericLocalNonlinearSolver(max_iters=30, tol=1e-13)
    for trial in range(20):
        Qk = random_states(rng, 1)[:, 0]
        lam, ca, dt = rng.uniform(0.8, 1.15), rng.uniform(0.05, 1.0), rng.choice([0.25, 1.0])
        vel = rng.normal() * 5e-3 if trial else 0.0
        st, Q, dQ, it, rn, dQv = tb.sarcomere_local_solve(model, Qk, Qk, lam, ca, dt, tight, velocity=vel)
        ost, oQ, odQ, oit, orn, odQv = oracle.rdq20mf_local_solve(Qk, Qk, lam, ca, dt, tight.tol, tight.max_iters, dlam=vel, rate=True)
        assert st == ost == 0 and it == oit
        np.testing.assert_allclose(Q, oQ, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(dQ, odQ, rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(dQv, odQv, rtol=1e-9, atol=1e-13)
        assert np.abs(dQv[:16]).max() == 0.0                      # the chain does not see the velocity
        res = (Q - Qk) / dt - tb.sarcomere_rhs(model, Q, lam, vel, ca)[0]
        assert np.abs(res).max() < 1e-12
        if abs(vel) > 1e-4:                                        # away from the regularised kink of |v|
            h = 1e-7
            Qp = tb.sarcomere_local_solve(model, Qk, Qk, lam, ca, dt, tight, velocity=vel + h)[1]
            Qm = tb.sarcomere_local_solve(model, Qk, Qk, lam, ca, dt, tight, velocity=vel - h)[1]
            np.testing.assert_allclose(dQv, (Qp - Qm) / (2 * h), rtol=1e-5, atol=1e-8)


@pytest.mark.gpu
def test_device_rate_coupled_local_solve_matches_oracle(tb, oracle, device):
    rng = np.random.default_rng(6)
    model = tb.RDQ20MFModel()
    tight = tb.GenericLocalNonlinearSolver(30, 1e-13)
    for npts in (3, 130):
        Qk = random_states(rng, npts)
        lam, ca, vel = rng.uniform(0.8, 1.15, npts), rng.uniform(0.05, 1.0, npts), rng.normal(size=npts) * 5e-3
        known = tb.SarcomereState(device, model, npts, initial=Qk)
        st = tb.SarcomereState(device, model, npts, initial=Qk)
        dQ, dQv = device.zeros(20 * npts), device.zeros(20 * npts)
        nf = tb.sarcomere_implicit_step(st, known, 0.5, device.to_device(lam), device.to_device(ca), tight, dstate_dstretch=dQ,
                                        velocity=device.to_device(vel), dstate_dvelocity=dQv)
        assert nf == 0
        got, gdQ, gdQv = st.to_host(), dQ.to_host().reshape(20, npts), dQv.to_host().reshape(20, npts)
        for i in range(npts):
            code, oQ, odQ, _, _, odQv = oracle.rdq20mf_local_solve(Qk[:, i], Qk[:, i], lam[i], ca[i], 0.5, tight.tol, tight.max_iters, dlam=vel[i], rate=True)
            assert code == 0
            np.testing.assert_allclose(got[:, i], oQ, rtol=1e-11, atol=1e-14)
            np.testing.assert_allclose(gdQ[:, i], odQ, rtol=1e-8, atol=1e-12)
            np.testing.assert_allclose(gdQv[:, i], odQv, rtol=1e-8, atol=1e-12)
