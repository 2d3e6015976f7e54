"""Pure-Python big-int model of the Hades permutation: reference schedule and the algebraically
equivalent "sparse partial round" schedule the HIP kernels execute.  Test infrastructure.

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
