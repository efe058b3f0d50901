"""The reference's own benchmark list (Source/Tests/HPBenchmarks.cpp:25-236) on the GPU path:
Creation @1e-10 Exponential(3), 8 M random Query, 200^3 grid Query, 8 M QueryWithGradient, UnionSDF @1e-8."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import ctypes as C
import hpsdf_loader
import oracle as O
H = hpsdf_loader.load()
stream = torch.cuda.Stream()


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


with torch.cuda.stream(stream):
    ctx = H.Context(0, stream.cuda_stream)
    cfg = H.make_config(1e-10)
    cfg.nearnessWeighting_type, cfg.nearnessWeighting_strength = 2, 3.0   # HPBenchmarks.cpp:34-39
    sphere = H.Field.sphere()
    res = {}
    blk = [None]
    def create():
        blk[0], res["st"] = H.create_block(ctx, cfg, sphere, 1024)
    print("Creation (sphere, 1e-10, Exponential 3): %.2f ms  %s" % (timed(create), {k: res["st"][k] for k in ("n_nodes", "n_coeffs", "jobs", "rounds")}), flush=True)
    cfgc = H.make_config(1e-10, continuity=True)
    cfgc.nearnessWeighting_type, cfgc.nearnessWeighting_strength = 2, 3.0
    cfgc.threadCount = 16
    def create_c():
        H.create_block(ctx, cfgc, sphere, 1024)
    ms = timed(create_c)
    print("Creation + continuity (strength 8): %.2f ms  continuity %s" % (ms, {k: round(v, 2) if isinstance(v, float) else v for k, v in H.continuity_last_stats().items()}), flush=True)
    tree = H.DeviceTree(ctx, blk[0])
    n = 8_000_000
    pts = torch.from_numpy(O.splitmix64_points(n, seed=5)).cuda()
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    grad = torch.empty(3 * n, dtype=torch.float64, device="cuda")
    print("8 M random Query (HBM-resident): %.3f ms" % timed(lambda: tree.query_device(pts.data_ptr(), n, out.data_ptr()), 10), flush=True)
    g = torch.linspace(-0.5, 0.5, 200, dtype=torch.float64, device="cuda")
    grid = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).contiguous()
    print("200^3 grid Query: %.3f ms" % timed(lambda: tree.query_device(grid.data_ptr(), len(grid), out.data_ptr()), 10), flush=True)
    L = H.lib()
    print("8 M QueryWithGradient: %.3f ms" % timed(lambda: H.check(L.hpsdf_query_gradient_device(
        ctx.handle, tree.handle, C.c_void_p(pts.data_ptr()), n, C.c_void_p(out.data_ptr()), C.c_void_p(grad.data_ptr()))), 10), flush=True)
    # UnionSDF @1e-8 (HPBenchmarks.cpp:206-236): sphere tree, then union with the mirrored sphere
    cfg8 = H.make_config(1e-8)
    oc = H.Octree(jobs_per_round=1024)
    other = H.Field.sphere((-0.25, 0.0, 0.0), 0.5)
    def union():
        oc.Create(cfg8, sphere)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        oc.UnionSDF(other)
        torch.cuda.synchronize()
        res["union"] = (time.perf_counter() - t0) * 1e3
    union(); union()
    print("UnionSDF (1e-8): %.2f ms  %s" % (res["union"], {k: oc.stats[k] for k in ("n_nodes", "n_coeffs", "jobs", "rounds")}), flush=True)
