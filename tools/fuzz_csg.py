"""Randomised sweep of the CSG rebuilds (Octree::UnionSDF / SubtractSDF / IntersectSDF, Octree.cpp:355-400; by hand, on a GPU box): a random
analytic field A is built into a tree, a random field B is combined with it (the old tree queried inside the new build's field), and the
rebuilt tree is compared with the oracle's -- MemoryBlock byte for byte, Query bit for bit.  Usage: python tools/fuzz_csg.py [cases] [first seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hpsdf_loader
import oracle as O
H = hpsdf_loader.load()
bits = lambda a: np.ascontiguousarray(a, np.float64).view(np.uint64)
cases, first = (int(sys.argv[1]) if len(sys.argv) > 1 else 16), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def prims(rng, n):
    spec = []
    for k in range(n):
        kind = int(rng.integers(0, 3))
        c = rng.uniform(-0.25, 0.25, 3)
        if kind == H.PRIM_SPHERE:
            par = list(c) + [float(rng.uniform(0.1, 0.35))]
        elif kind == H.PRIM_BOX:
            par = list(c) + list(rng.uniform(0.08, 0.25, 3))
        else:
            par = list(c) + [float(rng.uniform(0.12, 0.25)), float(rng.uniform(0.04, 0.08))]
        spec.append((kind, H.OP_UNION if k == 0 else int(rng.integers(0, 3)), [float(x) for x in par]))
    return spec


bad = 0
for seed in range(first, first + cases):
    rng = np.random.default_rng(seed)
    a_spec, b_spec = prims(rng, int(rng.integers(1, 3))), prims(rng, 1)
    target = float(rng.choice([1e-4, 1e-5, 1e-6]))
    K = int(rng.choice([256, 1024]))
    op = int(rng.integers(0, 3))
    name = {H.OP_UNION: "UnionSDF", H.OP_SUBTRACT: "SubtractSDF", H.OP_INTERSECT: "IntersectSDF"}[op]
    t0 = time.time()
    t = H.Octree(jobs_per_round=K)
    t.Create(H.make_config(target), H.Field.analytic(a_spec))
    first_blk = t.ToMemoryBlock()
    getattr(t, name)(H.Field.analytic(b_spec))
    got = t.ToMemoryBlock()
    ocfg = O.default_config(target)
    old = O.Tree.create(ocfg, O.AnalyticField(a_spec), K)
    want = O.Tree.create(ocfg, O.TreeCsgField(old, O.AnalyticField(b_spec), op), K)
    wb = want.to_block()
    pts = O.splitmix64_points(20000, seed=seed)
    same_q = np.array_equal(bits(t.Query(pts)), bits(want.query(pts)))
    ok = first_blk == old.to_block() and got == wb and same_q
    bad += 0 if ok else 1
    a = O.parse_block(got)
    print("seed %3d: %d + 1 prims, %-12s target %g K %4d -> %5d nodes | first build %s rebuilt block %s query %s (%.1f s)%s"
          % (seed, len(a_spec), name, target, K, len(a["degree"]), first_blk == old.to_block(), got == wb, same_q, time.time() - t0, "" if ok else "  <-- FAIL"), flush=True)
print("FAILURES: %d of %d" % (bad, cases))
sys.exit(1 if bad else 0)
