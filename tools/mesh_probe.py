"""Mesh-field Create at scale (stand-in for BASELINE configs 3-5: dragon/Ramesses are not in the mount)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hpsdf_loader
from helpers import icosphere, displaced_torus
H = hpsdf_loader.load()
ctx = H.Context(0)
for level in (a for a in sys.argv[1:] or ["5", "7", "8"]):
    t0 = time.time()
    if level == "torus":  # 2 097 152 triangles
        verts, tris = displaced_torus()
        level = "displaced torus 1024x1024"
    else:
        level = int(level)
        verts, tris = icosphere(level, 0.4)
        # bumpy sphere so that the SDF is not trivially a sphere
        d = verts / np.linalg.norm(verts, axis=1, keepdims=True)
        verts = (verts * (1.0 + 0.08 * np.sin(9 * d[:, 0]) * np.cos(7 * d[:, 1]) + 0.05 * np.sin(11 * d[:, 2]))[:, None]).astype(np.float32)
        level = "icosphere L%d" % level
    t1 = time.time()
    lo, hi = verts.min(0) - 0.02, verts.max(0) + 0.02
    t2 = time.time(); f = H.Field.mesh(ctx, verts, tris); t3 = time.time()
    cfg = H.make_config(1e-5, tuple(lo), tuple(hi))
    blk, st = H.create_block(ctx, cfg, f, 1024); t4 = time.time()
    if os.environ.get("HPSDF_MESH_STATS"):
        f.mesh_stats()
    blk, st = H.create_block(ctx, cfg, f, 1024); t5 = time.time()
    if os.environ.get("HPSDF_MESH_STATS"):
        ms = f.mesh_stats()
        q = max(1, ms["wave_queries"])
        print("   traversal: %d wave queries; per query of 64 samples: %.0f nodes visited, %.0f (lane, triangle) pairs through the "
              "lower-bound test, %.0f through the closest-point test"
              % (q, ms["node_visits"] / q, ms["tri_tests"] / q, ms["tri_test_lanes"] / q))
        print("              %.0f (lane, leaf) pairs queued, %.0f lower-bound batches, %.0f closest-point batches, %.1f of 64 seeds already the answer"
              % (ms["leaf_pairs"] / q, ms["bound_batches"] / q, ms["closest_batches"] / q, ms["seed_exact"] / q))
    for tgt in [float(x) for x in os.environ.get("MESH_PROBE_TARGETS", "").split(",") if x]:  # e.g. MESH_PROBE_TARGETS=1e-6,1e-7
        c2 = H.make_config(tgt, tuple(lo), tuple(hi))
        H.create_block(ctx, c2, f, 1024)
        ta = time.time(); b2, s2 = H.create_block(ctx, c2, f, 1024); tb = time.time()
        print("   Create %g: %.1f ms, %d rounds, %d nodes, %d samples = %.0f M samples/s"
              % (tgt, (tb - ta) * 1e3, s2["rounds"], s2["n_nodes"], s2["samples"], s2["samples"] / (tb - ta) / 1e6))
    pts = np.random.default_rng(1).uniform(lo, hi, (1_000_000, 3))
    t6 = time.time(); v = f.eval(ctx, pts); t7 = time.time()
    print("%s: %d tris | gen %.1fs | prepare (half-edges+BVH+upload) %.2fs | Create 1e-5: %.1f ms (first %.1f ms) "
          "nodes %d samples %d | field eval 1M pts %.1f ms (incl. PCIe)" % (level, len(tris), t1 - t0, t3 - t2, (t5 - t4) * 1e3,
          (t4 - t3) * 1e3, st["n_nodes"], st["samples"], (t7 - t6) * 1e3))
