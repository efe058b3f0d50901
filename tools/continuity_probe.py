"""Create + continuity post-process on the reference's benchmark tree (sphere @ 1e-10, Exponential(3), strength 8), a few
times: for timing the solve loop (HPSDF_TRACE=1 prints the phases) and for rocprofv3 --kernel-trace --stats."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hpsdf_loader
H = hpsdf_loader.load()
ctx = H.Context(0)
cfg = H.make_config(1e-10, continuity=True)
cfg.nearnessWeighting_type, cfg.nearnessWeighting_strength = 2, 3.0
cfg.threadCount = 16
f = H.Field.sphere()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    t0 = time.perf_counter()
    blk, st = H.create_block(ctx, cfg, f, 1024)
    ms = (time.perf_counter() - t0) * 1e3
    print("Create + continuity %.2f ms  %s" % (ms, {k: round(v, 2) if isinstance(v, float) else v for k, v in H.continuity_last_stats().items()}), flush=True)
