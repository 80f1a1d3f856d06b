"""Prototype (numpy) of the sum-factorised Q1-hexahedron element matrices used by the fused M+K patch kernel.
Checks the factorised contraction against the direct 8-point sums  K[i,j] = Σ_q dN_j·G_q·dN_i,  M[i,j] = Σ_q d_q N_i N_j."""
import numpy as np

g = 0.5773502691896258
S = np.array([[-1, 1, 1, -1, -1, 1, 1, -1], [-1, -1, 1, 1, -1, -1, 1, 1], [-1, -1, -1, -1, 1, 1, 1, 1]])  # S[d][a]
bit = (S > 0).astype(int)  # node bit per direction


def xi(q, d):
    return g if (q >> d) & 1 else -g


def direct(x, D):
    K = np.zeros((8, 8)); M = np.zeros((8, 8))
    for q in range(8):
        N = np.array([0.125 * np.prod([1 + S[d][a] * xi(q, d) for d in range(3)]) for a in range(8)])
        dN = np.zeros((8, 3))
        for a in range(8):
            for d in range(3):
                dN[a, d] = 0.125 * np.prod([S[e][a] if e == d else 1 + S[e][a] * xi(q, e) for e in range(3)])
        J = x.T @ dN  # J[i][k] = Σ_a x[a][i] dN[a][k]
        det = np.linalg.det(J)
        Jinv = np.linalg.inv(J)
        G = -det * Jinv @ D @ Jinv.T
        K += dN @ G @ dN.T   # K[i,j] = Σ dN_i[n] G[m][n] dN_j[m] (G symmetric)
        M += det * np.outer(N, N)
    return K, M


def factorised(x, D):
    p, m = 0.5 * (1 + g), 0.5 * (1 - g)
    # modal geometry
    c = np.zeros((8, 3))
    sig = [np.ones(8), S[0], S[1], S[2], S[0] * S[1], S[1] * S[2], S[2] * S[0], S[0] * S[1] * S[2]]
    for k in range(8):
        c[k] = 0.125 * (sig[k][:, None] * x).sum(0)
    G = np.zeros((8, 6)); dq = np.zeros(8)
    for q in range(8):
        X, E, Z = xi(q, 0), xi(q, 1), xi(q, 2)
        J = np.zeros((3, 3))
        J[:, 0] = c[1] + c[4] * E + c[6] * Z + c[7] * E * Z
        J[:, 1] = c[2] + c[4] * X + c[5] * Z + c[7] * Z * X
        J[:, 2] = c[3] + c[5] * E + c[6] * X + c[7] * X * E
        A = np.array([[J[1, 1] * J[2, 2] - J[1, 2] * J[2, 1], J[0, 2] * J[2, 1] - J[0, 1] * J[2, 2], J[0, 1] * J[1, 2] - J[0, 2] * J[1, 1]],
                      [J[1, 2] * J[2, 0] - J[1, 0] * J[2, 2], J[0, 0] * J[2, 2] - J[0, 2] * J[2, 0], J[0, 2] * J[1, 0] - J[0, 0] * J[1, 2]],
                      [J[1, 0] * J[2, 1] - J[1, 1] * J[2, 0], J[0, 1] * J[2, 0] - J[0, 0] * J[2, 1], J[0, 0] * J[1, 1] - J[0, 1] * J[1, 0]]])
        det = J[0, 0] * A[0, 0] + J[0, 1] * A[1, 0] + J[0, 2] * A[2, 0]
        s = -0.25 / det           # the two e-factors (±1/2 each) are folded in here
        H = A @ D
        Gf = s * H @ A.T
        G[q] = [Gf[0, 0], Gf[0, 1], Gf[0, 2], Gf[1, 1], Gf[1, 2], Gf[2, 2]]
        dq[q] = det
    # 1-D factors: f(bit, qbit) = p if equal else m ; w(type, qbit): type 0 = (0,0), 1 = (0,1), 2 = (1,1)
    f = lambda b, qb: p if b == qb else m
    w = lambda t, qb: (f(0, qb) ** 2, p * m, f(1, qb) ** 2)[t]
    ty = lambda a, b: a + b  # pair type of two bits
    Gq = lambda comp: G[:, comp].reshape(2, 2, 2)  # [q3][q2][q1]
    # diagonal terms Ydd[c_e][c_f] over the two other directions (e < f)
    def Ydiag(comp, d):
        g3 = Gq(comp)
        Sx = g3.sum(axis=2 - d)          # sum over q_d → remaining two q bits, order [q_hi][q_lo]
        Y = np.zeros((3, 3))             # [type_lo][type_hi]
        for tl in range(3):
            for th in range(3):
                Y[tl, th] = sum(Sx[qh, ql] * w(tl, ql) * w(th, qh) for qh in range(2) for ql in range(2))
        return Y
    Y11, Y22, Y33 = Ydiag(0, 0), Ydiag(3, 1), Ydiag(5, 2)
    # cross terms X_de[a_d][b_e][type_f]: Σ_q G_de f_d(a, q_d) f_e(b, q_e) w_f(type, q_f)
    def Xcross(comp, d, e, fdir):
        g3 = Gq(comp)
        X = np.zeros((2, 2, 3))
        for a in range(2):
            for b in range(2):
                for t in range(3):
                    acc = 0.0
                    for q in range(8):
                        qb = [(q >> k) & 1 for k in range(3)]
                        acc += g3[qb[2], qb[1], qb[0]] * f(a, qb[d]) * f(b, qb[e]) * w(t, qb[fdir])
                    X[a, b, t] = acc
        return X
    X12, X13, X23 = Xcross(1, 0, 1, 2), Xcross(2, 0, 2, 1), Xcross(4, 1, 2, 0)
    K = np.zeros((8, 8))
    for i in range(8):
        for j in range(8):
            bi, bj = bit[:, i], bit[:, j]
            si, sj = S[:, i], S[:, j]
            t = [ty(bi[d], bj[d]) for d in range(3)]
            v = si[0] * sj[0] * Y11[t[1], t[2]] + si[1] * sj[1] * Y22[t[0], t[2]] + si[2] * sj[2] * Y33[t[0], t[1]]
            # G12 [∂1Ni ∂2Nj + ∂2Ni ∂1Nj]
            v += si[0] * sj[1] * X12[bj[0], bi[1], t[2]] + sj[0] * si[1] * X12[bi[0], bj[1], t[2]]
            v += si[0] * sj[2] * X13[bj[0], bi[2], t[1]] + sj[0] * si[2] * X13[bi[0], bj[2], t[1]]
            v += si[1] * sj[2] * X23[bj[1], bi[2], t[0]] + sj[1] * si[2] * X23[bi[1], bj[2], t[0]]
            K[i, j] = v
    # mass
    d3 = dq.reshape(2, 2, 2)
    Z = np.zeros((3, 3, 3))
    for t1 in range(3):
        for t2 in range(3):
            for t3 in range(3):
                Z[t1, t2, t3] = sum(d3[q3, q2, q1] * w(t1, q1) * w(t2, q2) * w(t3, q3) for q1 in range(2) for q2 in range(2) for q3 in range(2))
    M = np.zeros((8, 8))
    for i in range(8):
        for j in range(8):
            t = [ty(bit[d, i], bit[d, j]) for d in range(3)]
            M[i, j] = Z[t[0], t[1], t[2]]
    return K, M


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    ref = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], float)
    for trial in range(5):
        x = ref * [0.7, 1.1, 0.9] + rng.uniform(-0.25, 0.25, (8, 3))
        B = rng.uniform(-1, 1, (3, 3)); D = B @ B.T + 0.3 * np.eye(3)
        K0, M0 = direct(x, D); K1, M1 = factorised(x, D)
        print(trial, np.abs(K1 - K0).max() / np.abs(K0).max(), np.abs(M1 - M0).max() / np.abs(M0).max())
