#!/usr/bin/env python3
"""The opt-in matrix-core fit (hpsdf_ctx_set_fast_fit, csrc/fit_mfma.hip) against the default bit-exact path and the
oracle: topology, coefficients, Query values, and how many refinement decisions of the default build sit inside the
guard band |pImp - hImp| <= 1e-9 max(|pImp|, |hImp|) -- the decisions a last-bit change could flip.
Then the fit micro-benchmark per degree with both kernels.   usage: python tools/fast_fit_check.py [--write]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hpsdf_loader
import oracle as O
H = hpsdf_loader.load()
ctx = H.Context(0)
fast = H.Context(0)
fast.set_fast_fit(True)
NCOEF = H.NCOEF
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def guard_band(cfg, field, K):
    """near-ties of the default build, from the stepwise API (Octree.cpp:814-825, :846-854 restated)"""
    b = H.Build(cfg, K)
    near = total = 0
    while True:
        n = b.select()
        if n == 0:
            break
        jobs = b.jobs(n)
        b.compute(ctx, field)
        hdr = b.results_host(ctx).reshape(n, 9)
        for j in range(n):
            jb = jobs[j]
            if jb.coarse or jb.degree < 3:  # only decisions that involve a fit of degree >= 4 can change
                continue
            p, d, err = jb.degree, jb.depth, jb.err
            himp = (1.0 / (7.0 * NCOEF[p])) * (err - 8.0 * hdr[j, 1:].max()) if d < 10 else 0.0
            pimp = (1.0 / (NCOEF[p + 1] - NCOEF[p])) * (err - 8.0 * hdr[j, 0]) if p < 11 else 0.0
            total += 1
            if abs(pimp - himp) <= 1e-9 * max(abs(pimp), abs(himp)):
                near += 1
        b.apply(hdr)
    b.close()
    return near, total


say("fast fit (matrix cores, degrees >= 4) vs default path; 100 000 query points")
pts = O.splitmix64_points(100000, seed=5)
ok = True
for name, field, ofield, target, K in (("A1 union3 1e-7", H.Field.union3(), O.union3_field(), 1e-7, 1024),
                                       ("A2 sphere 1e-8", H.Field.sphere(), O.sphere_field(), 1e-8, 1024),
                                       ("union3 1e-8", H.Field.union3(), O.union3_field(), 1e-8, 1024),
                                       ("sphere 1e-9", H.Field.sphere(), O.sphere_field(), 1e-9, 1024)):
    cfg = H.make_config(target)
    a_blk, a_st = H.create_block(ctx, cfg, field, K)
    b_blk, b_st = H.create_block(fast, cfg, field, K)
    a, b = O.parse_block(a_blk), O.parse_block(b_blk)
    same = len(a["degree"]) == len(b["degree"]) and np.array_equal(a["degree"], b["degree"]) and np.array_equal(a["childIdx"], b["childIdx"])
    qa, qb = H.DeviceTree(ctx, a_blk).query(pts), H.DeviceTree(ctx, b_blk).query(pts)
    true = ofield.eval(pts)
    near, tot = guard_band(cfg, field, K)
    leaf = a["degree"][a["degree"] != 13]
    say("%-16s %6d nodes, leaf degrees %s: topology %s; max|dcoef| %s; max|dQuery| %.3e; max|Query-F| default %.3e fast %.3e; "
        "guard-band decisions %d of %d" % (name, len(a["degree"]), np.bincount(leaf).tolist(), "identical" if same else "DIFFERS",
                                           "%.3e" % np.abs(a["coeffs"] - b["coeffs"]).max() if same else "n/a", np.abs(qa - qb).max(),
                                           np.abs(qa - true).max(), np.abs(qb - true).max(), near, tot))
    if same:
        ok &= np.abs(a["coeffs"] - b["coeffs"]).max() <= 1e-6
    ok &= np.abs(qb - true).max() <= max(1.5 * np.abs(qa - true).max(), 1e-6)

say("")
say("fit micro-benchmark, algorithmic TFLOP/s = 2 ncoef (4p+1)^3 cells / time (78.6 TFLOP/s FP64 peak)")
say("%-4s %8s | %-30s | %-30s" % ("p", "cells", "union3 field: default -> fast", "plane field (contraction only): default -> fast"))
cfg = H.make_config(1e-5)
plane = H.Field.analytic([(H.PRIM_PLANE, H.OP_UNION, [0.3, -0.2, 0.5, 0.1])])
u3 = H.Field.union3()
for p in (4, 5, 6, 7, 8):
    cells = 16384 if p <= 5 else 4096
    flops = 2.0 * NCOEF[p] * (4 * p + 1) ** 3 * cells
    r = []
    for f in (u3, plane):
        for c in (ctx, fast):
            ms = H.bench_fit(c, cfg, f, p, 5, cells, 3)
            r.append(flops / ms / 1e9)
    say("p%-3d %8d | %6.2f -> %6.2f TF (%4.1f %% -> %4.1f %%) | %6.2f -> %6.2f TF (%4.1f %% -> %4.1f %%)" % (
        p, cells, r[0], r[1], 100 * r[0] / 78.6, 100 * r[1] / 78.6, r[2], r[3], 100 * r[2] / 78.6, 100 * r[3] / 78.6))
say("OK" if ok else "FAILED")
if "--write" in sys.argv:
    open(os.path.join(ROOT, "profiles", "r02_fast_fit.txt"), "w").write("\n".join(lines) + "\n")
sys.exit(0 if ok else 1)
