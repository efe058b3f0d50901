"""Time of hpsdf_field_create_mesh (upload + half-edge twins + BVH) on the 2 097 152-triangle torus, device vs host build."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hpsdf_loader
from helpers import displaced_torus, icosphere
H = hpsdf_loader.load()
ctx = H.Context(0)
for name, (v, t) in (("torus 2 097 152", displaced_torus()), ("icosphere L8 1 310 720", icosphere(8, 0.4))):
    for mode in ("device", "host"):
        os.environ["HPSDF_MESH_HOST_BUILD"] = "1" if mode == "host" else "0"
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            f = H.Field.mesh(ctx, v, t)
            ts.append((time.perf_counter() - t0) * 1e3)
            f.close()
        print("%s triangles, %s build: %s ms" % (name, mode, ", ".join("%.1f" % x for x in ts)), flush=True)
