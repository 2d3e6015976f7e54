"""Pure-Python big-int models of the Hades permutation: the reference schedule and the algebraically
equivalent schedules of poseidon252_amd/csrc/tables.hpp — sparse partial rounds, ARMA recurrence, integer MDS
(schedule 4) and integer MDS + integer ARMA (schedule 5, what the HIP kernels execute) — each derived here
independently of the C++.  Test infrastructure.

Reference schedule: src/hades/permutation.rs:105-123 with scalar.rs:39-64
  round r:  x <- M * S_r(x + C_r)      (S on all 5 lanes in full rounds, on lane 4 in partial rounds)

Optimised schedule (derived here; must give the same field elements, hence bit-exact limbs):
  (1) partial-round constants are pushed forward through the linear layer so that each partial round
      adds ONE constant (to lane 4, before the S-box); the accumulated vector lands in the ARC of the
      first closing full round.
  (2) M = M'' * M' with M' = diag-block(A,1) commuting with the lane-4 S-box and M'' sparse
      (identity except row 4 and column 4).  Chaining from the last partial round backwards leaves
      60 sparse matrices (9 multiplications each instead of 25) and one dense pre-matrix merged into
      the MDS of full round 3.
"""
import os

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
WIDTH, FULL, PARTIAL = 5, 8, 60
ROUNDS = FULL + PARTIAL
_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "poseidon252_amd", "assets")


def load_constants():
    """field VALUES (not Montgomery): from_raw semantics = the LE integer in the file, mod p"""
    arc = open(os.path.join(_ASSETS, "arc.bin"), "rb").read()
    mds = open(os.path.join(_ASSETS, "mds.bin"), "rb").read()
    C = [[int.from_bytes(arc[(r * 5 + i) * 32:(r * 5 + i + 1) * 32], "little") % P for i in range(5)] for r in range(ROUNDS)]
    M = [[int.from_bytes(mds[(i * 5 + j) * 32:(i * 5 + j + 1) * 32], "little") % P for j in range(5)] for i in range(5)]
    return C, M


def matvec(M, x):
    return [sum(M[k][j] * x[j] for j in range(len(x))) % P for k in range(len(M))]


def matmul(A, B):
    n, m, q = len(A), len(B), len(B[0])
    return [[sum(A[i][k] * B[k][j] for k in range(m)) % P for j in range(q)] for i in range(n)]


def matinv(A):
    n = len(A)
    a = [row[:] + [1 if i == j else 0 for j in range(n)] for i, row in enumerate(A)]
    for c in range(n):
        piv = next(r for r in range(c, n) if a[r][c] % P)
        a[c], a[piv] = a[piv], a[c]
        inv = pow(a[c][c], -1, P)
        a[c] = [v * inv % P for v in a[c]]
        for r in range(n):
            if r != c and a[r][c]:
                f = a[r][c]
                a[r] = [(v - f * w) % P for v, w in zip(a[r], a[c])]
    return [row[n:] for row in a]


def perm_reference(x, C=None, M=None):
    if C is None:
        C, M = load_constants()
    x = list(x)
    for r in range(ROUNDS):
        x = [(x[i] + C[r][i]) % P for i in range(5)]
        if r < FULL // 2 or r >= FULL // 2 + PARTIAL:
            x = [pow(v, 5, P) for v in x]
        else:
            x[4] = pow(x[4], 5, P)
        x = matvec(M, x)
    return x


def derive_optimised(C, M):
    """Returns dict with
         c_first[5]            ARC of round 0
         full_add[r][5]        vector added AFTER the matrix of full round r (r = 0..2, 64..66), = C[r+1]
         m_pre[5][5]           matrix used by full round 3 (= M'_1 * M)
         pre_add[5]            vector added after m_pre: k_1 on lane 4 only
         sparse[q] (q=0..59)   dict(w[4], d, b[4], add[5]) :
                                 v  = (x4)^5                      (x4 already holds its constant)
                                 y4 = sum_j w[j]*x[j] + d*v + add[4]
                                 y_i = x_i + b[i]*v + add[i]      (add[i]=0 except after the last sparse layer)
    """
    Rf = FULL // 2
    # (1) forward-push partial constants
    k = []
    delta = [0] * 5
    for q in range(PARTIAL):
        a = [(delta[i] + C[Rf + q][i]) % P for i in range(5)]
        k.append(a[4])
        rest = a[:4] + [0]
        delta = matvec(M, rest)
    closing_first = [(C[Rf + PARTIAL][i] + delta[i]) % P for i in range(5)]
    # (2) sparse factorisation, last partial round first
    sparse = [None] * PARTIAL
    Mk = [row[:] for row in M]
    for q in range(PARTIAL - 1, -1, -1):
        A = [row[:4] for row in Mk[:4]]
        b = [Mk[i][4] for i in range(4)]
        c = Mk[4][:4]
        d = Mk[4][4]
        Ainv = matinv(A)
        w = [sum(c[i] * Ainv[i][j] for i in range(4)) % P for j in range(4)]  # c^T A^{-1}
        sparse[q] = dict(w=w, d=d, b=b, add=[0] * 5)
        Mprime = [A[i] + [0] for i in range(4)] + [[0, 0, 0, 0, 1]]
        # check M'' * M' == Mk
        Mpp = [[1 if i == j else 0 for j in range(4)] + [b[i]] for i in range(4)] + [w + [d]]
        assert matmul(Mpp, Mprime) == Mk
        Mk = matmul(Mprime, M)
    m_pre = Mk
    # constants folded into layer outputs: lane-4 S-box input constant of the NEXT partial round
    for q in range(PARTIAL - 1):
        sparse[q]["add"][4] = k[q + 1]
    sparse[PARTIAL - 1]["add"] = closing_first
    pre_add = [0, 0, 0, 0, k[0]]
    full_add = {r: C[r + 1] for r in list(range(0, Rf - 1)) + list(range(Rf + PARTIAL, ROUNDS - 1))}
    return dict(c_first=C[0], full_add=full_add, m_pre=m_pre, pre_add=pre_add, sparse=sparse)


def perm_optimised(x, C=None, M=None, T=None):
    if C is None:
        C, M = load_constants()
    if T is None:
        T = derive_optimised(C, M)
    Rf = FULL // 2
    x = [(x[i] + T["c_first"][i]) % P for i in range(5)]
    for r in range(Rf):
        x = [pow(v, 5, P) for v in x]
        if r < Rf - 1:
            x = [(a + b) % P for a, b in zip(matvec(M, x), T["full_add"][r])]
        else:
            x = [(a + b) % P for a, b in zip(matvec(T["m_pre"], x), T["pre_add"])]
    for q in range(PARTIAL):
        s = T["sparse"][q]
        v = pow(x[4], 5, P)
        y4 = (sum(s["w"][j] * x[j] for j in range(4)) + s["d"] * v + s["add"][4]) % P
        y = [(x[i] + s["b"][i] * v + s["add"][i]) % P for i in range(4)] + [y4]
        x = y
    for r in range(Rf + PARTIAL, ROUNDS):
        x = [pow(v, 5, P) for v in x]
        x = matvec(M, x)
        if r < ROUNDS - 1:
            x = [(a + b) % P for a, b in zip(x, T["full_add"][r])]
    return x


def sponge(tag, inputs, out_len, perm=perm_reference):
    """SAFE sponge as driven by Hash::finalize (values, not Montgomery)"""
    st = [tag, 0, 0, 0, 0]
    pos = 0
    for e in inputs:
        if pos == 4:
            st = perm(st)
            pos = 0
        st[1 + pos] = (st[1 + pos] + e) % P
        pos += 1
    out, ps = [], 4
    for _ in range(out_len):
        if ps == 4:
            st = perm(st)
            ps = 0
        out.append(st[1 + ps])
        ps += 1
    return out


if __name__ == "__main__":
    import random
    C, M = load_constants()
    T = derive_optimised(C, M)
    rng = random.Random(1)
    for _ in range(5):
        x = [rng.randrange(P) for _ in range(5)]
        assert perm_reference(x, C, M) == perm_optimised(x, C, M, T)
    print("optimised schedule == reference schedule")
    print(["%064x" % v for v in perm_reference([0, 1, 2, 3, 4], C, M)])


# ------------------------------------------------------------------------------------------------
# "ARMA" form of the partial rounds.  With M = [[A, b], [c^T, d]] (lane 4 = the S-box lane), the
# partial rounds are a 4th-order linear time-invariant system driven by the S-box outputs v_q:
#     L_{q+1} = A L_q + b v_q ,   u_{q+1} = c^T L_q + d v_q + k_{q+1}      (u = S-box input)
# Cayley-Hamilton on A (chi(x) = x^4 - a1 x^3 - a2 x^2 - a3 x - a4) eliminates L:
#     u_{q+1} = sum_{m=1..4} a_m u_{q+1-m} + sum_{n=0..4} beta_n v_{q-n} + kappa_{q+1}      (q >= 5)
# i.e. 9 multiplications and ONE reduction per partial round.  The first 4 partial rounds run in the
# sparse state-space form (they also create the history), and after the last one the state lanes
# 0..3 are recovered from the last four u and v through the observability matrix.
# ------------------------------------------------------------------------------------------------
def charpoly4(A):
    """returns [a1,a2,a3,a4] with A^4 = a1 A^3 + a2 A^2 + a3 A + a4 I  (Faddeev-LeVerrier)"""
    n = 4
    I = [[1 if i == j else 0 for j in range(n)] for i in range(n)]
    Mk = [row[:] for row in I]
    coeffs = []
    for k in range(1, n + 1):
        AM = matmul(A, Mk)
        ck = (-sum(AM[i][i] for i in range(n)) * pow(k, -1, P)) % P
        coeffs.append(ck)  # coefficient of x^(n-k)
        Mk = [[(AM[i][j] + (ck if i == j else 0)) % P for j in range(n)] for i in range(n)]
    return [(-c) % P for c in coeffs]


def derive_arma(C, M):
    T = derive_optimised(C, M)
    Rf = FULL // 2
    A = [row[:4] for row in M[:4]]
    b = [M[i][4] for i in range(4)]
    c = M[4][:4]
    d = M[4][4]
    a = charpoly4(A)  # a[0]=a1..a[3]=a4
    # Markov parameters g_0 = d, g_i = c^T A^(i-1) b
    g = [d]
    vec = b[:]
    for _ in range(4):
        g.append(sum(c[i] * vec[i] for i in range(4)) % P)
        vec = matvec(A, vec)
    beta = [(g[n] - sum(a[m - 1] * g[n - m] for m in range(1, min(4, n) + 1))) % P for n in range(5)]
    # constants: k_q (q = 1..60) as in derive_optimised step (1); k_61 := closing constant of lane 4
    k = [None] * 62
    delta = [0] * 5
    for q in range(1, PARTIAL + 1):
        aq = [(delta[i] + C[Rf + q - 1][i]) % P for i in range(5)]
        k[q] = aq[4]
        delta = matvec(M, aq[:4] + [0])
    closing = [(C[Rf + PARTIAL][i] + delta[i]) % P for i in range(5)]
    k[61] = closing[4]
    kappa = {q: (k[q] - sum(a[m - 1] * k[q - m] for m in range(1, 5))) % P for q in range(6, 62)}
    # exit: L_61 = Gy * (y_58..y_61) + Gv * (v_57..v_60),  y = u - k
    powA = [[[1 if i == j else 0 for j in range(4)] for i in range(4)]]
    for _ in range(4):
        powA.append(matmul(powA[-1], A))
    O = [[sum(c[i] * powA[r][i][j] for i in range(4)) % P for j in range(4)] for r in range(4)]  # rows c^T A^r
    Oinv = matinv(O)
    Toep = [[(g[r - s] if s <= r else 0) for s in range(4)] for r in range(4)]  # y_{q+1+r} -= sum_s g_{r-s} v_{q+s}
    A4Oinv = matmul(powA[4], Oinv)
    Gy = A4Oinv
    Kb = [[matvec(powA[3 - s], b)[i] for s in range(4)] for i in range(4)]  # columns A^(3-s) b
    A4OinvT = matmul(A4Oinv, Toep)
    Gv = [[(Kb[i][s] - A4OinvT[i][s]) % P for s in range(4)] for i in range(4)]
    exit_add = [(closing[i] - sum(Gy[i][r] * k[58 + r] for r in range(4))) % P for i in range(4)]
    return dict(T=T, a=a, beta=beta, kappa=kappa, Gy=Gy, Gv=Gv, exit_add=exit_add)


def perm_arma(x, C=None, M=None, AR=None):
    if C is None:
        C, M = load_constants()
    if AR is None:
        AR = derive_arma(C, M)
    T = AR["T"]
    Rf = FULL // 2
    x = [(x[i] + T["c_first"][i]) % P for i in range(5)]
    for r in range(Rf):
        x = [pow(v, 5, P) for v in x]
        if r < Rf - 1:
            x = [(p + q) % P for p, q in zip(matvec(M, x), T["full_add"][r])]
        else:
            x = [(p + q) % P for p, q in zip(matvec(T["m_pre"], x), T["pre_add"])]
    # partial rounds 1..4 in sparse form (lane-4 value IS u_q: constants already folded in)
    u, v = {1: x[4]}, {}
    for q in range(1, 5):
        s = T["sparse"][q - 1]
        v[q] = pow(x[4], 5, P)
        y4 = (sum(s["w"][j] * x[j] for j in range(4)) + s["d"] * v[q] + s["add"][4]) % P
        x = [(x[i] + s["b"][i] * v[q]) % P for i in range(4)] + [y4]
        u[q + 1] = y4
    # ARMA rounds: u_6 .. u_61
    for q in range(5, PARTIAL + 1):
        v[q] = pow(u[q], 5, P)
        u[q + 1] = (sum(AR["a"][m - 1] * u[q + 1 - m] for m in range(1, 5))
                    + sum(AR["beta"][n] * v[q - n] for n in range(5)) + AR["kappa"][q + 1]) % P
    # exit: lanes 0..3 from the history; lane 4 = u_61 (holds the closing constant)
    L = [(sum(AR["Gy"][i][r] * u[58 + r] for r in range(4)) + sum(AR["Gv"][i][s] * v[57 + s] for s in range(4))
          + AR["exit_add"][i]) % P for i in range(4)]
    x = L + [u[61]]
    for r in range(Rf + PARTIAL, ROUNDS):
        x = [pow(t, 5, P) for t in x]
        x = matvec(M, x)
        if r < ROUNDS - 1:
            x = [(p + q) % P for p, q in zip(x, T["full_add"][r])]
    return x


# ------------------------------------------------------------------------------------------------
# Schedule 4 — "integer MDS": the schedule the kernels execute (tables.hpp step 5, hades29.hpp)
#
# The MDS matrix is a Cauchy matrix in disguise: mds.bin holds M[i][j] = R/(i+j+5) mod p with R = 2^256
# (mds_matrix.rs:21-36 reads Montgomery words with from_raw).  So M = (R/L) * N with the INTEGER matrix
# N[i][j] = L/(i+j+5), L = lcm(5..13) = 360360, every entry < 2^17: a product by N[i][j] is 9 MACs
# (one digit) instead of 81, and the constant field factor R/L is carried in the *scale* of the stored
# state, which costs nothing because x -> x^5 is homogeneous:
#     stored Z = s_k * y   (y = true S-box input of round k, one scale s_k for all five lanes)
#     full round:     X_j = Z_j^5 / R'^4                       (3 Montgomery products, R' = 2^261)
#     partial round:  X_j = Z_j (j < 4),  X_4 = Z_4^5 / R'^4 * G_k / R'   with G_k = s_k^-4 R'^5, so
#                     that all five X_j carry the same scale (the one generic product of the round)
#     linear layer:   Z'_i = (sum_j N[i][j] X_j + kappa_i) / 2^29    — ONE Montgomery digit step —
#                     with s_{k+1} = e_k L / (R 2^29),  kappa_i = 2^29 s_{k+1} C_{k+1}[i]
#     after round 67: out_i * R = Z'_i * F / R',  F = R R' / s_68
# Everything below works on residues mod p exactly as the device does (digit-level effects — lazy
# residues, signed top digits — are covered by the C++ host build in test_host_arith.py).
# ------------------------------------------------------------------------------------------------
L_INT = 360360
N_INT = [[L_INT // (i + j + 5) for j in range(5)] for i in range(5)]
RP = pow(2, 261, P)   # R'
RM = pow(2, 256, P)   # R (Montgomery radix of BlsScalar)


def derive_int(C, M):
    assert all(M[i][j] == RM * pow(i + j + 5, -1, P) % P for i in range(5) for j in range(5))
    inv = lambda v: pow(v, -1, P)
    Rf = FULL // 2
    s = RM  # Z = in*R + C_0*R
    out = dict(c_first=[c * RM % P for c in C[0]], kappa=[], G={}, scales=[s])
    for k in range(ROUNDS):
        full = k < Rf or k >= Rf + PARTIAL
        if full:
            e = pow(s, 5, P) * inv(pow(RP, 4, P)) % P
        else:
            out["G"][k] = inv(pow(s, 4, P)) * pow(RP, 5, P) % P
            e = s
        s = e * L_INT % P * inv(RM) % P * inv(pow(2, 29, P)) % P
        out["scales"].append(s)
        out["kappa"].append([pow(2, 29, P) * s % P * C[k + 1][i] % P for i in range(5)] if k + 1 < ROUNDS else [0] * 5)
    out["F"] = RM * RP % P * inv(s) % P
    return out


def perm_int(x_mont, C=None, M=None, T=None):
    """x_mont: the five input residues in BlsScalar Montgomery form (x*R); returns out*R residues"""
    if C is None:
        C, M = load_constants()
    if T is None:
        T = derive_int(C, M)
    inv = lambda v: pow(v, -1, P)
    iRP, i29 = inv(RP), inv(pow(2, 29, P))
    mm = lambda a, b: a * b % P * iRP % P  # one Montgomery product (redc)
    sbox = lambda z: mm(mm(mm(z, z), mm(z, z)), z)
    Rf = FULL // 2
    Z = [(x_mont[i] + T["c_first"][i]) % P for i in range(5)]
    for k in range(ROUNDS):
        if k < Rf or k >= Rf + PARTIAL:
            X = [sbox(z) for z in Z]
        else:
            X = Z[:4] + [mm(sbox(Z[4]), T["G"][k])]
        Z = [(sum(N_INT[i][j] * X[j] for j in range(5)) + T["kappa"][k][i]) * i29 % P for i in range(5)]
    return [mm(z, T["F"]) for z in Z]


# ------------------------------------------------------------------------------------------------
# Schedule 5 — integer MDS in the full rounds + integer ARMA recurrence in the partial rounds
#
# The 60 partial rounds are the 4th-order linear system of derive_arma (u = S-box input, v = u^5):
#     u_{q+1} = sum_m a_m u_{q+1-m} + sum_n beta_n v_{q-n} + kappa_{q+1}
# With M = R * Cauchy the coefficients are a_m = R^m a~_m, beta_n = R^(n+1) b~_n with RATIONAL a~, b~ whose
# scaled forms A_m = a~_m D^m (D = 27720) and B_n = b~_n D^n K (K = 12870) are ONE-DIGIT INTEGERS (< 2^25).
# Stored values carry geometric scales  U_q = sigma_q u_q, sigma_{q+1} = sigma_q D / (R 2^29), and
# W_q = (sigma_q D / K) v_q  (= sbox(U_q) times one generic constant G_q, the only generic product of a round),
# so that a term of age j sits j digits lower in the accumulator:
#     U_{q+1} = ( sum_m A_m U_{q+1-m} 2^(29(5-m)) + sum_n B_n W_{q-n} 2^(29(4-n)) ) / 2^(29*5) + K_{q+1}
# = 81 one-digit MACs + 5 Montgomery digit steps, instead of five 5-term rows.
# Entry: the linear layer of full round 3 produces U_1 (lane 4, integer row) and a VIRTUAL history
# (U_0, U_-1, U_-2, W_0; older v's are zero) — four generic rows — chosen so that the recurrence already holds for
# q = 1..4.  Exit: lanes 0..3 of the state are recovered from (U_58..U_61, W_57..W_60) by four generic rows.
# All additive constants are obtained from the zero-input trajectory of the affine system (they do not depend
# on the trajectory), so no closed form for pushed constants is needed.
# ------------------------------------------------------------------------------------------------
D_INT, K_INT = 27720, 12870


def _rational_arma():
    from fractions import Fraction as Fr
    C5 = [[Fr(1, i + j + 5) for j in range(5)] for i in range(5)]
    A = [r[:4] for r in C5[:4]]
    b = [C5[i][4] for i in range(4)]
    c = C5[4][:4]
    d = C5[4][4]
    mm = lambda X, Y: [[sum(X[i][k] * Y[k][j] for k in range(len(Y))) for j in range(len(Y[0]))] for i in range(len(X))]
    I4 = [[Fr(int(i == j)) for j in range(4)] for i in range(4)]
    Mk, ck, cs = [[Fr(0)] * 4 for _ in range(4)], Fr(1), []
    for k in range(1, 5):  # Faddeev-LeVerrier
        AMk = mm(A, Mk) if k > 1 else [[Fr(0)] * 4 for _ in range(4)]
        Mk = [[AMk[i][j] + ck * I4[i][j] for j in range(4)] for i in range(4)]
        AM = mm(A, Mk)
        ck = -sum(AM[i][i] for i in range(4)) / k
        cs.append(ck)
    a = [-x for x in cs]
    h, row = [d], c[:]
    for _ in range(4):
        h.append(sum(row[i] * b[i] for i in range(4)))
        row = [sum(row[i] * A[i][j] for i in range(4)) for j in range(4)]
    beta = [h[n] - sum(a[m - 1] * h[n - m] for m in range(1, n + 1)) for n in range(5)]
    Aint = [a[m - 1] * D_INT ** m for m in range(1, 5)]
    Bint = [beta[n] * D_INT ** n * K_INT for n in range(5)]
    assert all(x.denominator == 1 for x in Aint + Bint)
    # exit rows (observability form, as derive_arma): lanes 0..3 after round 60 = Gy (u_58..u_61) + Gv (v_57..v_60) + const.
    # In the scaled variables the coefficients are D^(3-r) Gy~[i][r] and D^(3-s) K Gv~[i][s] (times powers of 2^-29):
    # rationals whose row-wise common denominator den_i leaves integers below 2^47.
    def inv4(Mx):
        n = 4
        aug = [row[:] + [Fr(int(i == j)) for j in range(n)] for i, row in enumerate(Mx)]
        for col in range(n):
            piv = next(r for r in range(col, n) if aug[r][col] != 0)
            aug[col], aug[piv] = aug[piv], aug[col]
            f = aug[col][col]
            aug[col] = [v / f for v in aug[col]]
            for r in range(n):
                if r != col and aug[r][col] != 0:
                    f = aug[r][col]
                    aug[r] = [v - f * w for v, w in zip(aug[r], aug[col])]
        return [row[n:] for row in aug]
    powA = [I4]
    for _ in range(4):
        powA.append(mm(powA[-1], A))
    O = [[sum(c[i] * powA[r][i][j] for i in range(4)) for j in range(4)] for r in range(4)]
    Gy = mm(powA[4], inv4(O))
    Toep = [[(h[r - t] if t <= r else Fr(0)) for t in range(4)] for r in range(4)]
    GyT = mm(Gy, Toep)
    Gv = [[sum(powA[3 - t][i][j] * b[j] for j in range(4)) - GyT[i][t] for t in range(4)] for i in range(4)]
    from math import lcm
    ex = []
    for i in range(4):
        cy = [Gy[i][r] * D_INT ** (3 - r) for r in range(4)]
        cv = [Gv[i][t] * D_INT ** (3 - t) * K_INT for t in range(4)]
        den = 1
        for x in cy + cv:
            den = lcm(den, x.denominator)
        ex.append((den, [int(x * den) for x in cy], [int(x * den) for x in cv]))
    # entry rows: the virtual history (u_0, u_-1, u_-2, v_0) as a linear function of the S-box outputs of full round 3
    # (constants aside).  Triangular solve of the q = 1..4 recurrence equations on the zero-input trajectory, all in
    # exact fractions; v_0 turns out to be the lane-4 S-box output itself and the u rows are small integers over 495.
    def resid(y1):
        x, u = {1: list(y1[:4])}, {1: y1[4]}
        for q in range(1, 5):
            x[q + 1] = [sum(A[i][j] * x[q][j] for j in range(4)) for i in range(4)]
            u[q + 1] = sum(c[j] * x[q][j] for j in range(4))
        return [u[q + 1] - sum(a[m - 1] * u[q + 1 - m] for m in range(1, 5) if q + 1 - m >= 1) for q in range(1, 5)]
    def solve(r):
        v0 = r[3] / beta[4]
        um0 = (r[2] - beta[3] * v0) / a[3]
        um1 = (r[1] - beta[2] * v0 - a[2] * um0) / a[3]
        um2 = (r[0] - beta[1] * v0 - a[1] * um0 - a[2] * um1) / a[3]
        return [um0, um1, um2, v0]
    Hm = [[None] * 5 for _ in range(4)]
    for j in range(5):
        e = [Fr(int(t == j)) for t in range(5)]
        col = solve(resid(e))
        for i in range(4):
            Hm[i][j] = col[i]
    HC = mm(Hm, C5)
    assert HC[3] == [0, 0, 0, 0, 1]
    from math import gcd
    ent = []
    for i in range(3):
        den = 1
        for x in HC[i]:
            den = lcm(den, x.denominator)
        nums = [int(x * den) for x in HC[i]]
        g = 0
        for x in nums:
            g = gcd(g, abs(x))
        ent.append([x // g for x in nums])
    return [int(x) for x in Aint], [int(x) for x in Bint], ex, ent


A_INT, B_INT, EXIT_INT, ENTRY_INT = _rational_arma()   # [15104, -4729406, 18244864, -419265], [990, -1555121, 23296324, -2924911, 1694]


def derive_armaint(C, M):
    inv = lambda v: pow(v, -1, P)
    Rf = FULL // 2
    AR = derive_arma(C, M)
    a, beta, Gy, Gv = AR["a"], AR["beta"], AR["Gy"], AR["Gv"]
    assert all(a[m - 1] == pow(RM, m, P) * A_INT[m - 1] % P * inv(pow(D_INT, m, P)) % P for m in range(1, 5))
    assert all(beta[n] == pow(RM, n + 1, P) * B_INT[n] % P * inv(pow(D_INT, n, P) * K_INT) % P for n in range(5))
    A4 = [row[:4] for row in M[:4]]
    cvec = M[4][:4]
    # ---- zero-input trajectory of the affine system (v = 0 everywhere), from an arbitrary first state y1 ----
    def traj(y1):
        x, u = {1: list(y1[:4])}, {1: y1[4]}
        for q in range(1, PARTIAL + 1):
            k = Rf + q  # constants added after partial round q are C[k] (k <= 64)
            x[q + 1] = [(sum(A4[i][j] * x[q][j] for j in range(4)) + C[k][i]) % P for i in range(4)]
            u[q + 1] = (sum(cvec[j] * x[q][j] for j in range(4)) + C[k][4]) % P
        return x, u
    x0, u0 = traj([0] * 5)
    kappa = {q + 1: (u0[q + 1] - sum(a[m - 1] * u0[q + 1 - m] for m in range(1, 5))) % P for q in range(5, PARTIAL + 1)}
    for q in range(1, 5):
        kappa[q + 1] = 0
    exit_add = [(x0[61][i] - sum(Gy[i][r] * u0[58 + r] for r in range(4))) % P for i in range(4)]
    # ---- virtual history theta = (u_0, u_-1, u_-2, v_0) = H y1 + h0 from the residuals r_q (q = 1..4) ----
    def residuals(y1):
        _, u = traj(y1)
        return [(u[q + 1] - sum(a[m - 1] * u[q + 1 - m] for m in range(1, 5) if q + 1 - m >= 1)) % P for q in range(1, 5)]
    def solve(r):
        v0 = r[3] * inv(beta[4]) % P
        um0 = (r[2] - beta[3] * v0) * inv(a[3]) % P
        um1 = (r[1] - beta[2] * v0 - a[2] * um0) * inv(a[3]) % P
        um2 = (r[0] - beta[1] * v0 - a[1] * um0 - a[2] * um1) * inv(a[3]) % P
        return [um0, um1, um2, v0]
    th0 = solve(residuals([0] * 5))
    H = [[0] * 5 for _ in range(4)]
    for j in range(5):
        e = [0] * 5
        e[j] = 1
        r0, rj = residuals([0] * 5), residuals(e)
        col = solve([(rj[i] - r0[i]) % P for i in range(4)])  # linear part: offsets cancel (solve is linear)
        for i in range(4):
            H[i][j] = col[i]
    # ---- scales ----
    out = dict(c_first=[c * RM % P for c in C[0]], fr_kappa={}, ent_fix=None, ent_add=None, K={}, G={}, ex_fix=None, ex_add=None)
    i29 = inv(pow(2, 29, P))
    step = L_INT * inv(RM) % P * i29 % P
    s = RM
    for k in range(Rf):  # opening full rounds; the layer of round 3 is the entry
        e = pow(s, 5, P) * inv(pow(RP, 4, P)) % P
        s_next = e * step % P
        if k < Rf - 1:
            out["fr_kappa"][k] = [pow(2, 29, P) * s_next % P * C[k + 1][i] % P for i in range(5)]
        else:
            sigma1 = s_next * pow(2, 29, P) % P   # U_1 = lane 4's integer row WITHOUT its digit step: scale e L / R
            mu = D_INT * inv(RM) % P              # every term of the recurrence at the same weight (round 2: D / (R 2^29))
            sig = lambda q: sigma1 * pow(mu, q - 1, P) % P          # also for q <= 0
            omg = lambda q: sig(q) * D_INT % P * inv(K_INT) % P
            # theta_i = sum_j (H M)_ij v_j + (H C_4 + h0)_i with v_j = X_j / e
            HM = [[sum(H[i][t] * M[t][j] for t in range(5)) % P for j in range(5)] for i in range(4)]
            hc = [(sum(H[i][t] * C[Rf][t] for t in range(5)) + th0[i]) % P for i in range(4)]
            tscale = [sig(0), sig(-1), sig(-2), omg(0)]
            gen = [[tscale[i] * HM[i][j] % P * inv(e) % P * RP % P for j in range(5)] for i in range(4)]   # generic form: coefficient * R'
            out["ent_add"] = [tscale[i] * hc[i] % P for i in range(4)]
            # integer form: theta_i = (sum_j n_ij X_j) / 2^(29 steps_i) * fix_i / R' + add_i with steps = digits of the n_ij;
            # W_0 = 28 X_4 + add_3 needs no generic product at all
            out["ent_fix"] = []
            for i in range(3):
                steps = 1 if max(abs(v) for v in ENTRY_INT[i]) < 1 << 28 else 2
                j0 = next(j for j in range(5) if ENTRY_INT[i][j])
                fix = gen[i][j0] * pow(2, 29 * steps, P) % P * inv(ENTRY_INT[i][j0]) % P
                assert all(gen[i][j] == fix * ENTRY_INT[i][j] % P * pow(i29, steps, P) % P for j in range(5))
                out["ent_fix"].append(fix)
            assert gen[3] == [0, 0, 0, 0, 28 * RP % P]
            out["fr_kappa"][k] = [0, 0, 0, 0, pow(2, 29, P) * s_next % P * C[Rf][4] % P]
        s = s_next
    for q in range(1, PARTIAL + 1):
        out["G"][q] = pow(RP, 5, P) * D_INT % P * inv(K_INT) % P * inv(pow(sig(q), 4, P)) % P
        out["K"][q + 1] = sig(q + 1) * kappa[q + 1] % P
    s = sig(61)
    # exit rows in integer form: Z_i = ( sum_r ny U_{58+r} + sum_t nv W_{57+t} ) / 2^58 * fix_i / R' + add_i
    for i in range(4):
        den, ny, nv = EXIT_INT[i]
        assert all(s * Gy[i][r] % P * inv(sig(58 + r)) % P == ny[r] * inv(den) % P for r in range(4))
        assert all(s * Gv[i][t] % P * inv(omg(57 + t)) % P == nv[t] * inv(den) % P for t in range(4))
    out["ex_fix"] = [pow(2, 58, P) * inv(EXIT_INT[i][0]) % P * RP % P for i in range(4)]
    out["ex_add"] = [s * exit_add[i] % P for i in range(4)]
    for k in range(Rf + PARTIAL, ROUNDS):
        e = pow(s, 5, P) * inv(pow(RP, 4, P)) % P
        s = e * step % P
        out["fr_kappa"][k] = [pow(2, 29, P) * s % P * C[k + 1][i] % P for i in range(5)] if k + 1 < ROUNDS else [0] * 5
    out["F"] = RM * RP % P * inv(s) % P
    return out


def perm_armaint(x_mont, C=None, M=None, T=None):
    if C is None:
        C, M = load_constants()
    if T is None:
        T = derive_armaint(C, M)
    inv = lambda v: pow(v, -1, P)
    iRP, i29 = inv(RP), inv(pow(2, 29, P))
    mm = lambda a, b: a * b % P * iRP % P
    sbox = lambda z: mm(mm(mm(z, z), mm(z, z)), z)
    irow = lambda X, i, kap: (sum(N_INT[i][j] * X[j] for j in range(5)) + kap) * i29 % P
    grow = lambda terms, add: (sum(x * n for x, n in terms) % P * iRP + add) % P   # generic row: redc(sum x*n) + add
    Rf = FULL // 2
    Z = [(x_mont[i] + T["c_first"][i]) % P for i in range(5)]
    for k in range(Rf - 1):
        X = [sbox(z) for z in Z]
        Z = [irow(X, i, T["fr_kappa"][k][i]) for i in range(5)]
    X = [sbox(z) for z in Z]
    th = []
    for i in range(3):
        steps = 1 if max(abs(v) for v in ENTRY_INT[i]) < 1 << 28 else 2
        acc = sum(ENTRY_INT[i][j] * X[j] for j in range(5)) * pow(i29, steps, P) % P
        th.append(grow([(acc, T["ent_fix"][i])], T["ent_add"][i]))
    th.append((28 * X[4] + T["ent_add"][3]) % P)
    U = {1: (sum(N_INT[4][j] * X[j] for j in range(5)) + T["fr_kappa"][Rf - 1][4]) % P, 0: th[0], -1: th[1], -2: th[2]}
    W = {0: th[3], -1: 0, -2: 0, -3: 0}
    for q in range(1, PARTIAL + 1):
        W[q] = mm(sbox(U[q]), T["G"][q])
        acc = sum(A_INT[m - 1] * U[q + 1 - m] for m in range(1, 5)) + sum(B_INT[n] * W[q - n] for n in range(5))
        U[q + 1] = (acc + T["K"][q + 1]) % P
    i58 = pow(i29, 2, P)
    Z = []
    for i in range(4):
        _, ny, nv = EXIT_INT[i]
        acc = sum(ny[r] * U[58 + r] for r in range(4)) + sum(nv[t] * W[57 + t] for t in range(4))
        Z.append(grow([(acc * i58 % P, T["ex_fix"][i])], T["ex_add"][i]))
    Z.append(U[61])
    for k in range(Rf + PARTIAL, ROUNDS):
        X = [sbox(z) for z in Z]
        Z = [irow(X, i, T["fr_kappa"][k][i]) for i in range(5)]
    return [mm(z, T["F"]) for z in Z]
