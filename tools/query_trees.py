"""Query throughput on trees of increasing refinement (10 M SplitMix64 points resident in HBM)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import hpsdf_loader
import oracle as O
H = hpsdf_loader.load()
stream = torch.cuda.Stream()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
with torch.cuda.stream(stream):
    ctx = H.Context(0, stream.cuda_stream)
    pts = torch.from_numpy(O.splitmix64_points(n)).cuda()
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    grad = torch.empty(3 * n, dtype=torch.float64, device="cuda")
    for name, field, target in (("C2 union3 1e-5", H.Field.union3(), 1e-5), ("union3 1e-6", H.Field.union3(), 1e-6),
                                ("A1 union3 1e-7", H.Field.union3(), 1e-7), ("A2 sphere 1e-8", H.Field.sphere(), 1e-8),
                                ("union3 1e-8", H.Field.union3(), 1e-8)):
        t0 = time.perf_counter()
        blk, st = H.create_block(ctx, H.make_config(target), field, 1024)
        tc = (time.perf_counter() - t0) * 1e3
        tree = H.DeviceTree(ctx, blk)
        info = tree.info()
        for _ in range(2):
            tree.query_device(pts.data_ptr(), n, out.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(10):
            tree.query_device(pts.data_ptr(), n, out.data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        pb = O.parse_block(blk)
        leaf = pb["degree"] != 13
        hist = {int(d): int((pb["degree"][leaf] == d).sum()) for d in np.unique(pb["degree"][leaf])}
        print("%-16s create %7.2f ms  nodes %6d leaves %6d max depth %d degrees %s | query %7.1f us = %6.1f Gpts/s (%.2f of HBM peak)"
              % (name, tc, info["n_nodes"], info["n_leaves"], info["max_depth"], hist, ms * 1e3, n / ms / 1e6, 32 * n / ms / 1e6 / 8000),
              flush=True)
    # QueryWithGradient on the last two kinds of tree (the reference's 8 M-point gradient benchmark, HPBenchmarks.cpp:169-203)
    import ctypes as C
    for name, field, target in (("C2 union3 1e-5", H.Field.union3(), 1e-5), ("A2 sphere 1e-8", H.Field.sphere(), 1e-8)):
        blk, st = H.create_block(ctx, H.make_config(target), field, 1024)
        tree = H.DeviceTree(ctx, blk)
        L = H.lib()
        call = lambda: H.check(L.hpsdf_query_gradient_device(ctx.handle, tree.handle, C.c_void_p(pts.data_ptr()), n,
                                                             C.c_void_p(out.data_ptr()), C.c_void_p(grad.data_ptr())))
        call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5):
            call()
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("%-16s QueryWithGradient %8.1f us = %6.1f Gpts/s (56 B/pt: %.2f of HBM peak)" % (name, ms * 1e3, n / ms / 1e6, 56 * n / ms / 1e6 / 8000), flush=True)
