"""Broad usage sweep on the GPU box under the library's DEFAULT settings: does anything an ordinary caller would try come back with an
error it should not?  Mesh fields of several sizes x rounds of 256..4096 jobs x thresholds 1e-4..1e-8, analytic fields with and without
weighting and continuity, the three CSG operations on the results -- every Create must either succeed or be refused by a build limit at
a tree size that deserves it (reported), and every result is queried.  No parity here (tools/fuzz_*.py do that); usage: usage_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, hpsdf_loader
from helpers import displaced_torus, icosphere
import oracle as O
H = hpsdf_loader.load(); ctx = H.Context(0)
pts = O.splitmix64_points(200000, seed=11) * 0.8
bad = refused = done = 0
def run(label, cfg, field, K, keep=None):
    global bad, refused, done
    t0 = time.time()
    try:
        blk, st = H.create_block(ctx, cfg, field, K)
        tree = H.DeviceTree(ctx, blk)
        v = tree.query(pts)
        ok = bool(np.all(np.isfinite(v[v < 1e300])))
        print("%-70s ok   %8d nodes %4d rounds %7.3f s%s" % (label, st["n_nodes"], st["rounds"], time.time() - t0, "" if ok else "  NON-FINITE QUERY VALUES"), flush=True)
        done += 1; bad += 0 if ok else 1
        if keep is not None: keep.append(blk)
        tree.close()
    except H.HpsdfError as e:
        s = str(e)
        if e.status == H.ERR_BUILD_LIMIT:
            refused += 1
            print("%-70s LIMIT after %.2f s: %s" % (label, time.time() - t0, s[s.find("after"):s.find("after") + 150]), flush=True)
        else:
            bad += 1
            print("%-70s ERROR status %d: %s" % (label, e.status, s[:200]), flush=True)
root = ((-0.45, -0.45, -0.2), (0.45, 0.45, 0.2))
for name, (verts, tris) in (("torus 192", displaced_torus(12, 8)), ("torus 6 k", displaced_torus(64, 48)), ("torus 131 k", displaced_torus(256, 256)), ("icosphere 81 k", icosphere(6))):
    f = H.Field.mesh(ctx, verts, tris)
    for K in (256, 1024, 4096):
        for t in (1e-4, 1e-5, 1e-6, 1e-7):
            r = root if name.startswith("torus") else ((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5))
            run("mesh %-14s K %4d target %g" % (name, K, t), H.make_config(t, *r), f, K)
    f.close()
for fname, f in (("sphere", H.Field.sphere()), ("union3", H.Field.union3())):
    for K in (256, 4096):
        for t in (1e-6, 1e-8):
            for w in (0, 1, 2):
                for cont in (False, True):
                    if cont and (K != 4096 or w): continue
                    cfg = H.make_config(t, continuity=cont)
                    cfg.nearnessWeighting_type, cfg.nearnessWeighting_strength = w, 3.0
                    run("%s K %4d target %g weighting %d continuity %d" % (fname, K, t, w, int(cont)), cfg, f, K)
oc = H.Octree(jobs_per_round=1024)
for op in ("UnionSDF", "SubtractSDF", "IntersectSDF"):
    t0 = time.time()
    try:
        oc.Create(H.make_config(1e-7), H.Field.sphere())
        getattr(oc, op)(H.Field.sphere((-0.25, 0.0, 0.0), 0.5))
        v = oc.Query(pts[:1000])
        print("%-70s ok   %7.3f s" % ("sphere @ 1e-7 then " + op, time.time() - t0), flush=True); done += 1
    except H.HpsdfError as e:
        bad += 1; print("%-70s ERROR status %d: %s" % (op, e.status, str(e)[:200]), flush=True)
print("SWEEP: %d built, %d refused by a build limit, %d errors" % (done, refused, bad))
sys.exit(1 if bad else 0)
