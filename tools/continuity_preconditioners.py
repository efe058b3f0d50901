"""CG iteration counts of the continuity system (M + strength I) x = strength c on the reference's benchmark tree (sphere, strength 8:
83 080 unknowns) under three preconditioners -- Jacobi (what oracle, host and device run), block Jacobi with one dense block per
leaf, symmetric Gauss-Seidel (an IC(0)-class one) -- and how serial the latter's triangular solves are.  CPU only (scipy).
usage: python tools/continuity_preconditioners.py"""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
import hpsdf_loader, oracle as O
H = hpsdf_loader.load()
cfg = O.default_config(1e-8)
t0 = time.time()
tree = O.Tree.create(cfg, O.sphere_field((0.25, 0, 0), 0.5), 1024, threads=8) if "threads" in O.Tree.create.__code__.co_varnames else O.Tree.create(cfg, O.sphere_field((0.25, 0, 0), 0.5), 1024)
blk = bytearray(tree.to_block()); print("tree built in %.1f s, %d bytes" % (time.time() - t0, len(blk)))
blk[-80 + 16] = 1  # continuity.enforce
rp, col, val, st = H.continuity_matrix(bytes(blk))
n = len(rp) - 1
A = sp.csr_matrix((val, col, rp), shape=(n, n))
print("n", n, "nnz", A.nnz, st)
a = O.parse_block(bytes(blk))
# right-hand side: use M x0 style: b = lambda * c (the system is (M + lambda I) x = lambda c): take random b
lam = 8.0
S = (A + lam * sp.identity(n)).tocsr()
print('symmetric:', abs(A - A.T).max())
b = lam * a['coeffs'][:n]
def run(M, name):
    it = [0]
    def cb(x): it[0] += 1
    x, info = spl.cg(S, b, x0=b.copy(), rtol=1e-6, maxiter=2000, M=M, callback=cb)
    print("%-28s iterations %d info %d" % (name, it[0], info))
d = S.diagonal()
run(spl.LinearOperator((n, n), lambda v: v / d), "Jacobi")
# block Jacobi by leaf: leaf coefficient ranges from coeffs_start and degree
deg, cs = a["degree"], a["coeffsStart"] if "coeffsStart" in a else a["coeffs_start"]
NC = [1, 4, 10, 20, 35, 56, 83, 120, 165, 220, 286, 364, 455]
leaves = [(int(cs[i]), NC[int(deg[i])]) for i in range(len(deg)) if deg[i] != 13]
Sd = S.tocsr()
blocks = []
for s0, k in leaves:
    B = Sd[s0:s0 + k, s0:s0 + k].toarray()
    blocks.append((s0, k, np.linalg.inv(B)))
def bj(v):
    out = np.empty_like(v)
    for s0, k, Bi in blocks: out[s0:s0 + k] = Bi @ v[s0:s0 + k]
    return out
run(spl.LinearOperator((n, n), bj), "block Jacobi (per leaf)")
L = sp.tril(S, format="csr"); U = sp.triu(S, format="csr")
def ssor(v):
    y = spl.spsolve_triangular(L, v, lower=True)
    y = d * y
    return spl.spsolve_triangular(U, y, lower=False)
run(spl.LinearOperator((n, n), ssor), "symmetric Gauss-Seidel")
# dependency levels of the lower triangular solve (how serial it is)
lev = np.zeros(n, np.int32)
indptr, indices = L.indptr, L.indices
for r in range(n):
    cols = indices[indptr[r]:indptr[r + 1]]
    cols = cols[cols < r]
    if len(cols): lev[r] = lev[cols].max() + 1
print("levels of the triangular solve:", int(lev.max()) + 1, "rows per level (median): %d" % np.median(np.bincount(lev)))

# ---- round 4: SpMV-only preconditioners (no triangular solves): Chebyshev polynomials of the Jacobi-scaled operator D^-1/2 S D^-1/2
# on [lmin, lmax] (lmax by 30 power iterations + 5 %, lmin = lmax / kappa for a few guesses), and the truncated Neumann series.
# What counts is SpMVs in all: a polynomial of degree m costs m SpMVs per application on top of the iteration's own.
dis = 1.0 / np.sqrt(d)
Ah = sp.diags(dis) @ S @ sp.diags(dis)
v = np.random.default_rng(0).standard_normal(n)
for _ in range(30):
    v = Ah @ v
    v /= np.linalg.norm(v)
lmax = float(v @ (Ah @ v)) * 1.05
print("largest eigenvalue of the Jacobi-scaled matrix ~ %.3f" % lmax)
def cheb(m, lmin):
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    def apply(r):  # m steps of the Chebyshev iteration for S z = r from z = 0, in the scaled variables
        rh = dis * r
        z = np.zeros(n); p = np.zeros(n); res = rh.copy()
        alpha = 0.0
        for k in range(m + 1):
            if k == 0:
                p = res / theta; alpha = 1.0 / theta
            else:
                beta = (0.5 * delta * alpha) ** 2 if k > 1 else 0.5 * (delta * alpha) ** 2
                alpha = 1.0 / (theta - beta / alpha)
                p = alpha * res + beta * p  # (standard three-term form)
            z = z + p
            if k < m: res = res - Ah @ p
        return dis * z
    return apply
for m in (1, 2, 3, 4):
    for kappa in (10.0, 30.0, 100.0):
        it = [0]
        def cb(x): it[0] += 1
        x, info = spl.cg(S, b, x0=b.copy(), rtol=1e-6, maxiter=2000, M=spl.LinearOperator((n, n), cheb(m, lmax / kappa)), callback=cb)
        print("Chebyshev degree %d on [lmax/%g, lmax]: iterations %3d, SpMVs in all %4d (Jacobi: 107)" % (m, kappa, it[0], it[0] * (m + 1)))
