"""What Octree::Create does under the reference's DEFAULT-constructed Config (Source/HP/Config.cpp:5-14: targetErrorThreshold 1e-10,
nearnessWeighting None, continuity ON with strength 8, root [-1/2, 1/2]^3) on the reference's own test field (the sphere of
HPUnitTests.cpp:48-51) -- the build the reference did not finish in 900 s / 80 CPU-min during the survey -- and on union3.
Every build either completes (statistics, time, accuracy) or is refused by the build limits (hpsdf_ctx_set_build_limits: status, message,
time to the refusal).  usage: python tools/default_config_create.py [max_bytes_GiB ...]   (no argument: the library's default limits)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import hpsdf_loader
import oracle as O
H = hpsdf_loader.load()
os.environ.pop("HPSDF_TRACE", None)


def free_gib():
    import torch
    f, t = torch.cuda.mem_get_info()
    return f / 2**30, t / 2**30


def one(ctx, name, field, cfg, K, what):
    t0 = time.perf_counter()
    try:
        blk, st = H.create_block(ctx, cfg, field, K)
    except H.HpsdfError as e:
        print("%-34s REFUSED after %.2f s with status %d: %s" % (what, time.perf_counter() - t0, e.status, e), flush=True)
        return
    dt = time.perf_counter() - t0
    pb = O.parse_block(blk)
    leaf = pb["degree"] != 13
    hist = {int(d): int((pb["degree"][leaf] == d).sum()) for d in np.unique(pb["degree"][leaf])}
    dh = {int(d): int((pb["depth"][leaf] == d).sum()) for d in np.unique(pb["depth"][leaf])}
    pts = O.splitmix64_points(200000, seed=3)
    q = H.DeviceTree(ctx, blk).query(pts)
    truth = (O.sphere_field() if name == "sphere" else O.union3_field()).eval(pts)
    print("%-34s %.3f s  rounds %d jobs %d fits %d samples %.3g | nodes %d coeffs %d (block %.1f MB) degrees %s depths %s | total error %.3e | max|Query-F| %.2e"
          % (what, dt, st["rounds"], st["jobs"], st["fits"], st["samples"], st["n_nodes"], st["n_coeffs"], len(blk) / 1e6, hist, dh, st["total_error"],
             np.abs(q - truth).max()), flush=True)


limits = [float(a) for a in sys.argv[1:]] or [0.0]
print("device memory: %.1f GiB free of %.1f" % free_gib(), flush=True)
for lim in limits:
    ctx = H.Context(0)
    if lim:
        ctx.set_build_limits(max_bytes=int(lim * 2**30))
    print("--- build limits: %s" % ("the library's defaults (nodes: none; bytes of nodes and coefficients: 1/64 of the free device memory, at least 1 GiB)" if not lim else "max_bytes = %g GiB" % lim), flush=True)
    for name, field in (("sphere", H.Field.sphere), ("union3", H.Field.union3)):
        for target in (1e-9, 1e-10):
            cfg = H.make_config(target)
            one(ctx, name, field(), cfg, 1024, "%s @ %g, continuity off, K=1024" % (name, target))
        dflt = H.Config()  # Config.cpp:5-14 as it stands: 1e-10, continuity ON
        dflt.threadCount = 1
        one(ctx, name, field(), dflt, 1024, "%s under Config() itself" % name)
        one(ctx, name, field(), H.make_config(1e-10), 4096, "%s @ 1e-10, K=4096" % name)
    ctx.close()
