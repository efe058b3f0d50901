"""Where does a *_host call cost the same on the calling thread and as a launch?  Per call size n: hpsdf_query_host / _gradient_host / _ray_host on
the headline tree and on union3 @ 1e-7, hpsdf_field_eval_host on a 1.3 M-triangle mesh -- each forced to the host path (limit 1e9) and to the device
path (HPSDF_SMALL_QUERIES_ON_DEVICE=1) in two child processes.  usage: python tools/host_call_thresholds.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, hpsdf_loader
from helpers import icosphere
H = hpsdf_loader.load()
ctx = H.Context(0)
rng = np.random.default_rng(0)
def timeit(f, reps):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e6
for name, target in (("headline", 1e-5), ("union3@1e-7", 1e-7)):
    tree = H.DeviceTree(ctx, H.create_block(ctx, H.make_config(target), H.Field.union3(), 1024)[0])
    for n in (1, 8, 32, 64, 128, 256, 512, 1024):
        p = rng.uniform(-0.5, 0.5, (n, 3)); d = rng.standard_normal((n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True); tm = np.full(n, 2.0)
        reps = 300
        print("%%s n %%4d: query %%7.1f us  gradient %%7.1f us  ray %%8.1f us" %% (name, n, timeit(lambda: tree.query(p), reps), timeit(lambda: tree.query_with_gradient(p), reps),
              timeit(lambda: tree.query_ray(p, d, tm), 60)), flush=True)
v, t = icosphere(8, 0.35)
f = H.Field.mesh(ctx, v, t)
for n in (1, 2, 3, 4, 8, 16, 32):
    p = rng.uniform(-0.45, 0.45, (n, 3))
    print("mesh 1.3 M triangles n %%3d: %%8.1f us" %% (n, timeit(lambda: f.eval(ctx, p), 60)), flush=True)
''' % (ROOT, ROOT)
for label, env in (("HOST path (limits 1e9)", {"HPSDF_HOST_QUERY_POINTS": "1000000000", "HPSDF_HOST_GRADIENT_POINTS": "1000000000", "HPSDF_HOST_RAYS": "1000000000", "HPSDF_HOST_MESH_POINTS": "1000000000"}),
                   ("DEVICE path (HPSDF_SMALL_QUERIES_ON_DEVICE=1)", {"HPSDF_SMALL_QUERIES_ON_DEVICE": "1"})):
    print("==== " + label, flush=True)
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
    print("\n".join(l for l in (r.stdout + r.stderr).splitlines() if "amdgpu.ids" not in l), flush=True)
