"""Latency of the host-array entry points (what the C++ drop-in's scalar Query(pt) pays per call)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ctypes as C
import hpsdf_loader
H = hpsdf_loader.load()
ctx = H.Context(0)
blk, _ = H.create_block(ctx, H.make_config(1e-5), H.Field.union3(), 1024)
tree = H.DeviceTree(ctx, blk)
L = H.lib()
for n in (1, 1000, 100000, 8000000):
    pts = np.random.default_rng(0).uniform(-0.5, 0.5, (n, 3))
    out = np.empty(n)
    call = lambda: H.check(L.hpsdf_query_host(ctx.handle, tree.handle, pts.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p)))
    call(); call()
    reps = 2000 if n == 1 else (200 if n <= 100000 else 5)
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    dt = (time.perf_counter() - t0) / reps
    print("hpsdf_query_host n=%8d: %10.1f us per call = %8.2f Mpts/s" % (n, dt * 1e6, n / dt / 1e6), flush=True)
