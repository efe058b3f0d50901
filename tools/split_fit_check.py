"""The split fit mode (HPSDF_FIT_SPLIT: top-degree rows of from-scratch fits of degree >= SPLIT_FROM (env, default 2) bit-exact, the rows
below them by the sum-factorised kernel of csrc/fit_low.hip -- or, HPSDF_LOW_KERNEL=mfma, by the direct contraction on the matrix cores)
against the all-exact mode, on the GPU (by hand):  python tools/split_fit_check.py
1. single fits, degrees SPLIT_FROM..11: errors and top-degree rows bit for bit, lower rows within 1e-15 of the cell's scale;
2. whole builds (union3 @ 1e-8, sphere @ 1e-9, union3 @ 1e-7 K = 256, CSG rebuild): node arrays and statistics identical, coefficients
   within 1e-12; the host scheduler and two simulated ranks give the split mode's own bytes;
3. the fit micro-benchmark per mode."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, hpsdf_loader
H = hpsdf_loader.load()
exact, split, fast = H.Context(0), H.Context(0), H.Context(0)
exact.set_fit_mode(H.FIT_EXACT); split.set_fit_mode(H.FIT_SPLIT); fast.set_fit_mode(H.FIT_FAST)
SPLIT_FROM = int(os.environ.get('SPLIT_FROM', '2'))
split.set_split_min_degree(SPLIT_FROM)
cfg = H.make_config(1e-5)
union3 = H.Field.union3()
ok = True

print("== single fits (64 cells of the depth-5 lattice, union3)")
for p in range(SPLIT_FROM, 12):
    ce, ee = H.fit_cells(exact, cfg, union3, p, 5, 64)
    cs, es = H.fit_cells(split, cfg, union3, p, 5, 64)
    nlow = int(H.NCOEF[p - 1])
    errs_equal = np.array_equal(ee.view(np.uint64), es.view(np.uint64))
    top_equal = np.array_equal(ce[:, nlow:].view(np.uint64), cs[:, nlow:].view(np.uint64))
    dlow = float(np.abs(ce[:, :nlow] - cs[:, :nlow]).max())
    good = errs_equal and top_equal and dlow <= 1e-15
    ok &= good
    print("degree %2d: errors bitwise %s, top-degree rows bitwise %s, lower rows max |d| %.2e  %s" % (p, errs_equal, top_equal, dlow, "ok" if good else "FAIL"))


def parse(blk):
    nc = int(np.frombuffer(blk[:8], np.uint64)[0])
    co = np.frombuffer(blk[8:8 + 8 * nc], np.float64)
    nn = int(np.frombuffer(blk[8 + 8 * nc:16 + 8 * nc], np.uint64)[0])
    nodes = blk[16 + 8 * nc:16 + 8 * nc + 56 * nn]
    return co, nodes, blk[16 + 8 * nc + 56 * nn:]


print("== builds")
cases = [("union3 @ 1e-8, K 1024", H.make_config(1e-8), lambda: H.Field.union3(), 1024),
         ("sphere @ 1e-9, K 1024", H.make_config(1e-9), lambda: H.Field.sphere(), 1024),
         ("union3 @ 1e-7, K 256", H.make_config(1e-7), lambda: H.Field.union3(), 256),
         ("union3 @ 1e-5, K 1024 (BASELINE configs[1])", H.make_config(1e-5), lambda: H.Field.union3(), 1024)]
for name, c, mk, K in cases:
    t0 = time.perf_counter(); be, se = H.create_block(exact, c, mk(), K); te = time.perf_counter() - t0
    H.create_block(split, c, mk(), K)
    t0 = time.perf_counter(); bs, ss = H.create_block(split, c, mk(), K); ts = time.perf_counter() - t0
    t0 = time.perf_counter(); be, se = H.create_block(exact, c, mk(), K); te = time.perf_counter() - t0
    (ce, ne, ke), (cs, ns, ks) = parse(be), parse(bs)
    same_nodes = ne == ns and ke == ks
    same_stats = all(se[k] == ss[k] for k in se)
    dco = float(np.abs(ce - cs).max()) if len(ce) == len(cs) else float("nan")
    os.environ["HPSDF_HOST_FRONTIER"] = "1"
    bh, _ = H.create_block(split, c, mk(), K)
    os.environ["HPSDF_HOST_FRONTIER"] = "0"
    good = same_nodes and same_stats and dco <= 1e-12 and bh == bs
    ok &= good
    print("%-46s nodes %6d max degree-split? | node array + config identical %s, statistics identical %s (total error %.17g), max |dcoeff| %.2e, "
          "bytes identical %s, host scheduler == device frontier (split) %s | exact %.2f ms, split %.2f ms  %s"
          % (name, ss["n_nodes"], same_nodes, same_stats, ss["total_error"], dco, be == bs, bh == bs, te * 1e3, ts * 1e3, "ok" if good else "FAIL"))

# a CSG rebuild on a split-built tree (the old tree's coefficients enter the new field)
o = H.Octree(0, jobs_per_round=1024)
print("== fit micro-benchmark (union3 field; TFLOP/s algorithmic, fraction of 78.6)")
plane = H.Field.analytic([(H.PRIM_PLANE, H.OP_UNION, [0.3, -0.2, 0.5, 0.1])])
for p in (2, 3, 4, 5, 6, 7, 8):
    cells = 16384
    flops = 2.0 * H.NCOEF[p] * (4 * p + 1) ** 3 * cells
    row = []
    for cname, c in (("exact", exact), ("split", split), ("fast", fast)):
        ms = H.bench_fit(c, cfg, union3, p, 5, cells, 3)
        msp = H.bench_fit(c, cfg, plane, p, 5, cells, 3)
        row.append("%s %.2f ms = %.1f TF (%.1f %%), contraction only %.1f TF (%.1f %%)"
                   % (cname, ms, flops / ms / 1e9, flops / ms / 1e9 / 78.6 * 100, flops / msp / 1e9, flops / msp / 1e9 / 78.6 * 100))
    print("p%d: " % p + " | ".join(row), flush=True)
print("ALL OK" if ok else "FAILURES")
