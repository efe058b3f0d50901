"""Octree::Create at the reference's DEFAULT threshold (Config.cpp:7: 1e-10, weighting None): the build the
reference did not finish in 900 s / 80 CPU-min during the survey."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import hpsdf_loader
import oracle as O
H = hpsdf_loader.load()
ctx = H.Context(0)
os.environ.pop("HPSDF_TRACE", None)
for name, field, target in (("sphere", H.Field.sphere(), 1e-9), ("sphere", H.Field.sphere(), 1e-10), ("union3", H.Field.union3(), 1e-9)):
    t0 = time.perf_counter()
    blk, st = H.create_block(ctx, H.make_config(target), field, 1024)
    dt = time.perf_counter() - t0
    pb = O.parse_block(blk)
    leaf = pb["degree"] != 13
    hist = {int(d): int((pb["degree"][leaf] == d).sum()) for d in np.unique(pb["degree"][leaf])}
    dh = {int(d): int((pb["depth"][leaf] == d).sum()) for d in np.unique(pb["depth"][leaf])}
    pts = O.splitmix64_points(200000, seed=3)
    q = H.DeviceTree(ctx, blk).query(pts)
    truth = (O.sphere_field() if name == "sphere" else O.union3_field()).eval(pts)
    print("%s @ %g: %.3f s  rounds %d jobs %d fits %d samples %.3g | nodes %d coeffs %d degrees %s depths %s | max|Query-F| %.2e"
          % (name, target, dt, st["rounds"], st["jobs"], st["fits"], st["samples"], st["n_nodes"], st["n_coeffs"], hist, dh,
             np.abs(q - truth).max()), flush=True)
