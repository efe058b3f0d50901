"""QueryRay throughput: a 1024 x 1024 pinhole camera looking at the union3 scene from inside the root."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, torch
import hpsdf_loader
H = hpsdf_loader.load()
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    ctx = H.Context(0, stream.cuda_stream)
    for target in (1e-5, 1e-7):
        blk, _ = H.create_block(ctx, H.make_config(target), H.Field.union3(), 1024)
        tree = H.DeviceTree(ctx, blk)
        w = 1024
        u, v = np.meshgrid(np.linspace(-0.6, 0.6, w), np.linspace(-0.6, 0.6, w))
        d = np.stack([u, v, np.ones_like(u)], -1).reshape(-1, 3)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        o = np.tile(np.array([0.0, 0.0, -0.49]), (len(d), 1))
        n = len(d)
        do, dd = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
        tm = torch.full((n,), 2.0, dtype=torch.float64, device="cuda")
        hit = torch.zeros(n, dtype=torch.uint8, device="cuda")
        t = torch.zeros(n, dtype=torch.float64, device="cuda")
        L = H.lib()
        call = lambda: H.check(L.hpsdf_query_ray_device(ctx.handle, tree.handle, C.c_void_p(do.data_ptr()), C.c_void_p(dd.data_ptr()),
                                                        C.c_void_p(tm.data_ptr()), n, C.c_void_p(hit.data_ptr()), C.c_void_p(t.data_ptr())))
        call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print("union3 @ %g: %d rays, %.1f %% hit: %.2f ms = %.1f Mrays/s" % (target, n, 100.0 * hit.float().mean().item(), ms, n / ms / 1e3), flush=True)
