"""Randomised mesh parity sweep (by hand, on a GPU box): small random closed meshes (bumpy icospheres, displaced tori) with
random root boxes -- the tree built from the mesh field against the oracle's naive-scan build.  3 M samples per case."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, hpsdf_loader, oracle as O
from helpers import icosphere, displaced_torus
H = hpsdf_loader.load(); ctx = H.Context(0)
bad = 0
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 8)  # usage: fuzz_mesh_parity.py [first seed] [cases]
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    if seed % 2 == 0:
        verts, tris = icosphere(1 + seed % 4 // 2, 0.3, tuple(rng.uniform(-0.05, 0.05, 3)))
        d = verts / np.linalg.norm(verts, axis=1, keepdims=True)
        verts = (verts * (1 + 0.15 * np.sin(5 * d[:, 0] + seed) * np.cos(4 * d[:, 1]))[:, None]).astype(np.float32)
    else:
        verts, tris = displaced_torus(12 + 2 * seed, 8 + seed, 0.28, 0.09, 0.015)
    lo = tuple(float(x) for x in (verts.min(0) - rng.uniform(0.02, 0.1, 3)).astype(np.float32))
    hi = tuple(float(x) for x in (verts.max(0) + rng.uniform(0.02, 0.1, 3)).astype(np.float32))
    target = float(rng.choice([1e-4, 3e-5]))
    t0 = time.time()
    blk, st = H.create_block(ctx, H.make_config(target, lo, hi), H.Field.mesh(ctx, verts, tris), 1024)
    ot = O.Tree.create(O.default_config(target, lo, hi), O.MeshField(verts, tris), 1024)
    a, b = O.parse_block(blk), O.parse_block(ot.to_block())
    topo = np.array_equal(a["degree"], b["degree"]) and np.array_equal(a["childIdx"], b["childIdx"])
    dmax = float(np.abs(a["coeffs"] - b["coeffs"]).max()) if topo else float("nan")
    same = blk == ot.to_block()
    ok = same  # byte for byte (the device runs the host libm's acosf: csrc/acosf_host_libm.hpp)
    bad += 0 if ok else 1
    print("seed %d: %5d tris target %g -> %d nodes %d rounds | topology %s, max |dcoeff| %.2e, bytes identical %s (%.0f s)"
          % (seed, len(tris), target, st["n_nodes"], st["rounds"], topo, dmax, same, time.time() - t0), flush=True)
print("FAILURES:", bad)
