"""Randomised sweep of the split fit mode at its widest (run by hand on a GPU box): random analytic CSG fields, root boxes and round
sizes at deep thresholds, with EVERY from-scratch fit split (hpsdf_ctx_set_split_min_degree(2): rows of top degree by the bit-exact
kernel, the rows below them by the sum-factorised kernel of csrc/fit_low.hip) -- against the oracle: node array, Config and statistics
(the total error included) byte for byte, coefficients within 1e-12; and the host scheduler against the device-side frontier in the same
mode: identical bytes.  Usage: python tools/fuzz_split.py [cases] [first seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hpsdf_loader
import oracle as O
H = hpsdf_loader.load()
split = H.Context(0)
split.set_split_min_degree(2)
cases, first = (int(sys.argv[1]) if len(sys.argv) > 1 else 24), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for seed in range(first, first + cases):
    rng = np.random.default_rng(seed)
    spec = []
    for k in range(int(rng.integers(1, 4))):
        kind = int(rng.integers(0, 3))  # sphere, box, torus (a plane is a polynomial: nothing to refine)
        c = rng.uniform(-0.3, 0.3, 3)
        if kind == H.PRIM_SPHERE:
            par = list(c) + [float(rng.uniform(0.08, 0.35))]
        elif kind == H.PRIM_BOX:
            par = list(c) + list(rng.uniform(0.05, 0.25, 3))
        else:
            par = list(c) + [float(rng.uniform(0.1, 0.25)), float(rng.uniform(0.03, 0.08))]
        spec.append((kind, H.OP_UNION if k == 0 else int(rng.integers(0, 3)), [float(x) for x in par]))
    lo = tuple(float(x) for x in (-0.5 + rng.uniform(-0.2, 0.2, 3)).astype(np.float32))
    hi = tuple(float(x) for x in (0.5 + rng.uniform(-0.2, 0.3, 3)).astype(np.float32))
    target = float(rng.choice([1e-7, 3e-8, 1e-8, 3e-9]))
    K = int(rng.choice([256, 1024, 4096]))
    cfg, ocfg = H.make_config(target, lo, hi), O.default_config(target, lo, hi)
    t0 = time.time()
    try:
        blk, st = H.create_block(split, cfg, H.Field.analytic(spec), K)
    except H.HpsdfError as e:
        print("seed %d: product refused (%s)" % (seed, e)); continue
    ot = O.Tree.create(ocfg, O.AnalyticField(spec), K, threads=16)
    want = ot.to_block()
    a, b = O.parse_block(blk), O.parse_block(want)
    nc = len(b["coeffs"])
    nodes_same = len(blk) == len(want) and blk[8 + 8 * nc:] == want[8 + 8 * nc:]
    dco = float(np.abs(a["coeffs"] - b["coeffs"]).max()) if nodes_same else float("nan")
    stats_same = all(st[k] == ot.stats[k] for k in ("jobs", "rounds", "p_refines", "h_refines", "dropped") if k in ot.stats) and \
        np.float64(st["total_error"]).view(np.uint64) == np.float64(ot.stats.get("total_error", st["total_error"])).view(np.uint64)
    os.environ["HPSDF_HOST_FRONTIER"] = "1"
    hb, _ = H.create_block(split, cfg, H.Field.analytic(spec), K)
    os.environ["HPSDF_HOST_FRONTIER"] = "0"
    ok = nodes_same and stats_same and dco <= 1e-12 and hb == blk
    bad += 0 if ok else 1
    deg = a["degree"][a["degree"] != 13]
    print("seed %3d: %d prims target %g K %4d -> %5d nodes, %3d rounds, max degree %d | nodes+config %s stats %s max|dcoeff| %.1e host==device %s bytes==oracle %s (%.1f s)%s"
          % (seed, len(spec), target, K, st["n_nodes"], st["rounds"], int(deg.max()), nodes_same, stats_same, dco, hb == blk, blk == want, time.time() - t0,
             "" if ok else "  <-- FAIL"), flush=True)
print("FAILURES: %d" % bad)
sys.exit(1 if bad else 0)
