"""Randomised parity sweep (run by hand on a GPU box, not part of the test suite): random analytic CSG fields, root
boxes, thresholds, weightings and round sizes -- the product's block against the oracle's byte for byte, then Query,
QueryWithGradient and QueryRay bit for bit.  Usage: python tools/fuzz_parity.py [cases] [first seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hpsdf_loader
import oracle as O
from helpers import edge_points
H = hpsdf_loader.load()
ctx = H.Context(0)
if os.environ.get("HPSDF_REDUCTION_ORDER", "")[:1] in ("l", "1"):  # the product reads the variable itself at load; the oracle follows
    O.set_reduction_order(1)
    print("reduction order: (a . b) . c on both sides (product %d, oracle %d)" % (H.reduction_order(), O.reduction_order()))
bits = lambda a: np.ascontiguousarray(a, np.float64).view(np.uint64)
cases, first = (int(sys.argv[1]) if len(sys.argv) > 1 else 24), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for seed in range(first, first + cases):
    rng = np.random.default_rng(seed)
    spec = []
    for k in range(int(rng.integers(1, 4))):
        kind = int(rng.integers(0, 4))
        c = rng.uniform(-0.3, 0.3, 3)
        if kind == H.PRIM_SPHERE:
            par = list(c) + [float(rng.uniform(0.08, 0.35))]
        elif kind == H.PRIM_BOX:
            par = list(c) + list(rng.uniform(0.05, 0.25, 3))
        elif kind == H.PRIM_TORUS_Y:
            par = list(c) + [float(rng.uniform(0.1, 0.25)), float(rng.uniform(0.03, 0.08))]
        else:
            nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
            par = list(nrm) + [float(rng.uniform(-0.2, 0.2))]
        spec.append((kind, H.OP_UNION if k == 0 else int(rng.integers(0, 3)), [float(x) for x in par]))
    lo = tuple(float(x) for x in (-0.5 + rng.uniform(-0.2, 0.2, 3)).astype(np.float32))
    hi = tuple(float(x) for x in (0.5 + rng.uniform(-0.2, 0.3, 3)).astype(np.float32))
    target = float(rng.choice([1e-4, 1e-5, 1e-6, 3e-7, 1e-7, 3e-8]))
    K = int(rng.choice([256, 1024, 4096]))
    wtype = int(rng.choice([0, 0, 1, 2]))
    cfg, ocfg = H.make_config(target, lo, hi), O.default_config(target, lo, hi)
    if wtype:
        cfg.nearnessWeighting_type, cfg.nearnessWeighting_strength = wtype, 3.0
        ocfg.weighting_type, ocfg.weighting_strength = wtype, 3.0
    t0 = time.time()
    try:
        blk, st = H.create_block(ctx, cfg, H.Field.analytic(spec), K)
    except H.HpsdfError as e:
        print("seed %d: product refused (%s)" % (seed, e)); continue
    ot = O.Tree.create(ocfg, O.AnalyticField(spec), K)
    same = blk == ot.to_block()
    tree = H.DeviceTree(ctx, blk)
    pts = np.concatenate([(O.splitmix64_points(6000, seed=seed) + 0.5) * (np.array(hi) - np.array(lo)) + np.array(lo),
                          (edge_points(np.random.default_rng(seed), 800) + 0.5) * (np.array(hi) - np.array(lo)) + np.array(lo)])
    otq = O.Tree.from_block(blk)
    q_ok = np.array_equal(bits(tree.query(pts)), bits(otq.query(pts)))
    gv, gg = tree.query_with_gradient(pts); wv, wg = otq.query_with_gradient(pts)
    g_ok = np.array_equal(bits(gv), bits(wv)) and np.array_equal(bits(gg), bits(wg))
    d = rng.normal(size=(1500, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = pts[:1500]; tm = np.full(1500, 3.0)
    h1, t1 = tree.query_ray(o, d, tm); h2, t2 = otq.query_ray(o, d, tm)
    r_ok = np.array_equal(h1, h2) and np.array_equal(bits(t1), bits(t2))
    ok = same and q_ok and g_ok and r_ok
    bad += 0 if ok else 1
    print("seed %3d: %d prims w%d target %g K %4d -> %5d nodes, %3d rounds, max degree %d | block %s query %s gradient %s rays %s  (%.1f s)"
          % (seed, len(spec), wtype, target, K, st["n_nodes"], st["rounds"], tree.info()["max_degree"], same, q_ok, g_ok, r_ok, time.time() - t0), flush=True)
print("FAILURES: %d of %d" % (bad, cases))
sys.exit(1 if bad else 0)
