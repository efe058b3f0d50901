"""Randomised check of the BVH queries against the O(n) scan, on the GPU (by hand): random closed meshes (bumpy icospheres,
displaced tori) under random affine maps -- anisotropic scales down to 1e-3, translations up to 100 extents -- and point
sets that sit where the lower-bound filter and the leaf grouping could bite (on vertices / edges / faces, just off the
surface, the medial region, far away).  Per-lane traversal and the sampler's shared traversal must return the scan kernel's bits
-- no exceptions --, with the device-built LBVH (leaf sizes 1..16) and the host-built tree; and the scan kernel must return the
oracle's (= the reference's arithmetic) except where the reference's face-case point has left its triangle (closestSimplex's
stated rule, kernels.hip), which is checked against a float64 brute force.   usage: fuzz_mesh_bvh.py [cases] [first seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, hpsdf_loader
from helpers import fuzz_mesh_case, hard_points, true_distance_f64
H = hpsdf_loader.load(); ctx = H.Context(0)
cases, first = (int(sys.argv[1]) if len(sys.argv) > 1 else 24), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bits = lambda a: np.ascontiguousarray(a, np.float64).view(np.uint64)


bad = artefacts = 0
import oracle as O
check_ref = os.environ.get("FUZZ_MESH_NO_ORACLE") != "1"  # the oracle's scan is O(points x triangles) on one core
for seed in range(first, first + cases):
    verts, tris, leaf, host, scale, shift = fuzz_mesh_case(seed)
    os.environ["HPSDF_MESH_LEAF_TRIS"] = str(leaf)
    if host:
        os.environ["HPSDF_MESH_HOST_BUILD"] = "1"
    else:
        os.environ.pop("HPSDF_MESH_HOST_BUILD", None)
    t0 = time.time()
    f = H.Field.mesh(ctx, verts, tris)
    pts = hard_points(verts, tris, seed)
    want = f.eval_naive(ctx, pts)
    a, w = f.eval_lane(ctx, pts), f.eval_wave(ctx, pts)
    # strict: scan kernel, per-lane traversal and shared traversal share closestSimplex and its face-case rule -- one answer, always
    ok = np.array_equal(bits(a), bits(want)) and np.array_equal(bits(w), bits(want))
    # ... and the calling-thread path of scalar-sized calls (host copies of the arrays, the per-point traversal compiled for the host)
    idx = np.arange(seed % 7, len(pts), 23)
    one = np.concatenate([f.eval(ctx, pts[i:i + 1]) for i in idx])
    ok = ok and np.array_equal(bits(one), bits(want[idx]))
    note = ""
    if ok and check_ref and len(tris) <= 25000:
        # against the reference's arithmetic (the oracle's scan): a difference must be a reference artefact -- a face-case point that
        # left its triangle, i.e. a value below the float64 brute-force distance -- and the product's value that distance
        ref = O.MeshField(verts, tris).signed_distance(pts)[0].astype(np.float64)
        diff = np.nonzero(bits(ref) != bits(want))[0]
        # the mesh's scale as the traversal's slack sees it: its extent, or its largest coordinate if that is larger
        ext = max(float(np.linalg.norm(verts.max(0) - verts.min(0))), float(np.abs(verts).max()))
        for i in diff:
            d = true_distance_f64(verts, tris, pts[i])
            # the reference's value lies BELOW the true distance; the product's does not (beyond the slack), and above it by no more than
            # the rule's replacement -- the closest point of the triangle's BOUNDARY -- lies from the true closest point when that is inside
            # the needle: a fraction of the needle's width (seed 202707: 6e-6 on a mesh of extent 0.05 at 0.15; seed 910968, an icosphere
            # squashed 41 : 1: 3.4e-5 = 2.9e-4 of the extent, where the reference's own value is 5e-4 BELOW the distance)
            if not (abs(ref[i]) < d - 1e-6 * ext and -1e-5 * ext <= abs(want[i]) - d <= 5e-4 * max(ext, d)):
                ok = False
        if len(diff) and ok:
            note = " (%d point(s) where the reference's face-case point leaves a needle; product = float64 truth)" % len(diff)
            artefacts += len(diff)
    bad += 0 if ok else 1
    print("seed %3d: %6d tris, scale %s, shift %.1f, %s, leaf %2d, %d points: %s (%.1f s)"
          % (seed, len(tris), np.array2string(scale, precision=3), float(np.abs(shift).max()), "host tree" if host else "LBVH", leaf, len(pts),
             ("identical" + note) if ok else "DIFFERENT", time.time() - t0), flush=True)
    f.close()
print("FAILURES: %d of %d (points where the reference's own scan is off: %d)" % (bad, cases, artefacts))
