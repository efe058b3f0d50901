"""Randomised check of the BVH queries against the O(n) scan, on the GPU (by hand): random closed meshes (bumpy icospheres,
displaced tori) under random affine maps -- anisotropic scales down to 1e-3, translations up to 100 extents -- and point
sets that sit where the lower-bound filter and the leaf grouping could bite (on vertices / edges / faces, just off the
surface, the medial region, far away).  Per-lane traversal and the sampler's shared traversal must return the scan's bits,
with the device-built LBVH (leaf sizes 1..16) and the host-built tree.   usage: fuzz_mesh_bvh.py [cases] [first seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, hpsdf_loader
from helpers import icosphere, displaced_torus
from test_gpu_configs import _hard_points
H = hpsdf_loader.load(); ctx = H.Context(0)
cases, first = (int(sys.argv[1]) if len(sys.argv) > 1 else 24), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bits = lambda a: np.ascontiguousarray(a, np.float64).view(np.uint64)


def closest_pt_tri(p, a, b, c):  # Ericson, in float64
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = ab @ ap, ac @ ap
    if d1 <= 0 and d2 <= 0: return a
    bp = p - b; d3, d4 = ab @ bp, ac @ bp
    if d3 >= 0 and d4 <= d3: return b
    vc = d1 * d4 - d3 * d2
    if vc <= 0 and d1 >= 0 and d3 <= 0: return a + ab * (d1 / (d1 - d3))
    cp = p - c; d5, d6 = ab @ cp, ac @ cp
    if d6 >= 0 and d5 <= d6: return c
    vb = d5 * d2 - d1 * d6
    if vb <= 0 and d2 >= 0 and d6 <= 0: return a + ac * (d2 / (d2 - d6))
    va = d3 * d6 - d5 * d4
    if va <= 0 and (d4 - d3) >= 0 and (d5 - d6) >= 0: return b + (c - b) * ((d4 - d3) / ((d4 - d3) + (d5 - d6)))
    den = 1.0 / (va + vb + vc)
    return a + ab * (vb * den) + ac * (vc * den)


def scan_artefact(verts, tris, p, naive, bvh):
    """True if the exhaustive f32 scan, not the hierarchy, is the one that is off: on sliver triangles the reference's f32
    closest-point routine (Utility.cpp:5-97) can return a point outside the triangle, i.e. a distance below the triangle's
    own bounding-box distance, which no hierarchy reproduces.  Decided by an exhaustive scan in float64; the hierarchy's own
    value is held to 1e-5 of it (on needles the f32 routine's in-triangle answers carry that much conditioning error too:
    seed 100758, a sphere squashed to 1/1000 along x, is 1.7e-6 off)."""
    A, B, Cc = (verts[tris[:, k]].astype(np.float64) for k in range(3))
    d = min(np.linalg.norm(p - closest_pt_tri(p, A[i], B[i], Cc[i])) for i in range(len(tris)))
    return abs(abs(bvh) - d) <= 1e-5 * max(1.0, d) and abs(naive) < d - 1e-5


bad = artefacts = 0
for seed in range(first, first + cases):
    rng = np.random.default_rng(seed)
    if seed % 2 == 0:
        verts, tris = icosphere(int(rng.integers(2, 6)), 0.3)
        d = verts / np.linalg.norm(verts, axis=1, keepdims=True)
        verts = verts * (1 + rng.uniform(0, 0.3) * np.sin(rng.integers(2, 9) * d[:, 0] + seed) * np.cos(rng.integers(2, 9) * d[:, 1]))[:, None]
    else:
        verts, tris = displaced_torus(int(rng.integers(10, 120)), int(rng.integers(8, 90)), 0.28, 0.09, float(rng.uniform(0, 0.03)))
    scale = 10.0 ** rng.uniform(-3, 0, 3) if seed % 3 == 0 else np.ones(3)
    shift = rng.uniform(-1, 1, 3) * (100.0 if seed % 5 == 0 else 0.1)
    verts = (verts * scale + shift).astype(np.float32)
    leaf = int(rng.choice([1, 2, 4, 8, 16]))
    host = seed % 7 == 3
    os.environ["HPSDF_MESH_LEAF_TRIS"] = str(leaf)
    if host:
        os.environ["HPSDF_MESH_HOST_BUILD"] = "1"
    else:
        os.environ.pop("HPSDF_MESH_HOST_BUILD", None)
    t0 = time.time()
    f = H.Field.mesh(ctx, verts, tris)
    pts = _hard_points(None, verts, tris, seed)
    want = f.eval_naive(ctx, pts)
    a, w = f.eval_lane(ctx, pts), f.eval_wave(ctx, pts)
    ok = np.array_equal(bits(a), bits(want)) and np.array_equal(bits(w), bits(want))
    note = ""
    if not ok:
        # a point where the scan's value lies below the float64 truth: each traversal may either have come across the needle whose
        # closest point the f32 routine misplaced (and then agrees with the scan) or have pruned it by its box (and then holds the truth)
        diff = np.nonzero((bits(a) != bits(want)) | (bits(w) != bits(want)))[0]
        if len(diff) <= 8 and all(all(v[i] == want[i] or scan_artefact(verts, tris, pts[i], want[i], v[i]) for v in (a, w)) for i in diff):
            ok, note = True, " (%d point(s) where the f32 scan leaves a sliver triangle; hierarchy = float64 truth)" % len(diff)
            artefacts += len(diff)
    bad += 0 if ok else 1
    print("seed %3d: %6d tris, scale %s, shift %.1f, %s, leaf %2d, %d points: %s (%.1f s)"
          % (seed, len(tris), np.array2string(scale, precision=3), float(np.abs(shift).max()), "host tree" if host else "LBVH", leaf, len(pts),
             ("identical" + note) if ok else "DIFFERENT", time.time() - t0), flush=True)
    f.close()
print("FAILURES: %d of %d (points where the exhaustive f32 scan itself is off: %d)" % (bad, cases, artefacts))
