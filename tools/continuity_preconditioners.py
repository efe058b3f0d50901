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
