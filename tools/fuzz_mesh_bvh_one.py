"""One seed of tools/fuzz_mesh_bvh.py with every comparison reported on its own (which path differs from which, where, by how much).
usage: fuzz_mesh_bvh_one.py <seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, hpsdf_loader
from helpers import fuzz_mesh_case, hard_points, true_distance_f64
import oracle as O
H = hpsdf_loader.load(); ctx = H.Context(0)
bits = lambda a: np.ascontiguousarray(a, np.float64).view(np.uint64)
seed = int(sys.argv[1])
verts, tris, leaf, host, scale, shift = fuzz_mesh_case(seed)
os.environ["HPSDF_MESH_LEAF_TRIS"] = str(leaf)
if host:
    os.environ["HPSDF_MESH_HOST_BUILD"] = "1"
print("seed %d: %d tris, scale %s, shift %s, %s, leaf %d" % (seed, len(tris), scale, shift, "host tree" if host else "LBVH", leaf))
f = H.Field.mesh(ctx, verts, tris)
pts = hard_points(verts, tris, seed)
want = f.eval_naive(ctx, pts)
a, w = f.eval_lane(ctx, pts), f.eval_wave(ctx, pts)
for name, got in (("per-lane traversal", a), ("shared (wave) traversal", w)):
    d = np.nonzero(bits(got) != bits(want))[0]
    print("%s against the scan kernel: %d of %d points differ" % (name, len(d), len(pts)))
    for i in d[:8]:
        print("   point %d %s: scan %.17g, traversal %.17g, float64 truth %.17g" % (i, pts[i], want[i], got[i], true_distance_f64(verts, tris, pts[i])))
idx = np.arange(seed % 7, len(pts), 23)
one = np.concatenate([f.eval(ctx, pts[i:i + 1]) for i in idx])
d = np.nonzero(bits(one) != bits(want[idx]))[0]
print("calling-thread path against the scan kernel: %d of %d points differ" % (len(d), len(idx)))
for i in d[:8]:
    print("   point %d: scan %.17g, host %.17g" % (idx[i], want[idx[i]], one[i]))
ref = O.MeshField(verts, tris).signed_distance(pts)[0].astype(np.float64)
diff = np.nonzero(bits(ref) != bits(want))[0]
ext = max(float(np.linalg.norm(verts.max(0) - verts.min(0))), float(np.abs(verts).max()))
print("oracle's scan against the scan kernel: %d points differ (extent %.4g)" % (len(diff), ext))
for i in diff:
    d = true_distance_f64(verts, tris, pts[i])
    okp = abs(ref[i]) < d - 1e-6 * ext and -1e-5 * ext <= abs(want[i]) - d <= 5e-4 * max(ext, d)
    print("   point %d: oracle %.10g product %.10g truth %.10g | oracle below truth by %.3g (needs > %.3g), product - truth %.3g (allowed [%.3g, %.3g]) -> %s"
          % (i, ref[i], want[i], d, d - abs(ref[i]), 1e-6 * ext, abs(want[i]) - d, -1e-5 * ext, 5e-4 * max(ext, d), "artefact of the reference" if okp else "OUTSIDE THE RULE"))
