"""C2 (union3 @ 1e-5) and A1 (union3 @ 1e-7) Create through the device-side frontier, a few times, with HPSDF_TRACE --
run under rocprofv3 --kernel-trace for the per-kernel timeline (tools/frontier_trace.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HPSDF_TRACE"] = "1"
import numpy as np
import hpsdf_loader
H = hpsdf_loader.load()
ctx = H.Context(0)
for tg in (1e-5, 1e-7):
    f = H.Field.union3()
    ts = []
    for _ in range(12):
        t0 = time.perf_counter()
        H.create_block(ctx, H.make_config(tg), f, 1024)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("union3 @ %g: Create median %.3f ms, min %.3f ms" % (tg, float(np.median(ts)), min(ts)), flush=True)
