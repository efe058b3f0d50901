#!/usr/bin/env python3
"""Where query_general_lds_kernel's time goes on a refined tree (union3 @ 1e-7, 10 M random points): the kernel as it is, and with one
link of its chain taken out at a time (HPSDF_QUERY_LAB, queryGeneralBody's LAB: the values of those runs are not the tree's).  Also the
same points sorted by leaf (the lanes of a wave share lines) and the headline tree (one line a point, no walk) through the same kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
# the lab kernels live in lib/libhpsdf_lab.so only (round 6: the production library neither holds them nor reads HPSDF_QUERY_LAB);
# build it on demand -- BEFORE anything touches the GPU -- and load it instead of libhpsdf.so
import importlib.util
_spec = importlib.util.spec_from_file_location("hpsdf_build", os.path.join(ROOT, "hp-adaptive-signed-distance-field-octree_amd", "build.py"))
_b = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_b); _b.build_lab()
os.environ["HPSDF_LIBRARY"] = "lab"
import hpsdf_loader
import oracle as O
H = hpsdf_loader.load()
stream = torch.cuda.Stream()


def timed(tree, pts, reps=10):
    m = len(pts)
    out = torch.empty(m, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):
        tree.query_device(pts.data_ptr(), m, out.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        tree.query_device(pts.data_ptr(), m, out.data_ptr())
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, out


with torch.cuda.stream(stream):
    ctx = H.Context(0, stream.cuda_stream)
    n = 10_000_000
    rnd = torch.from_numpy(O.splitmix64_points(n)).cuda()
    blk, st = H.create_block(ctx, H.make_config(1e-7), H.Field.union3(), 1024)
    tree = H.DeviceTree(ctx, blk)
    base, ref = timed(tree, rnd)
    print("union3 @ 1e-7 (%d nodes), %d random points, query_general_lds_kernel" % (st["n_nodes"], n))
    print("  %-74s %7.1f us  (%.3f of HBM peak at 32 B a point)" % ("as it is", base, 32.0 * n / (base * 1e-6) / 8e12))
    names = {1: "no polynomial: the fetched rows are touched, not evaluated", 2: "every lane fetches its wave's first leaf: the same instructions, no gather traffic",
             3: "no second line for degree-3 leaves", 4: "no walk below the top table"}
    for lab in (1, 2, 3, 4):
        os.environ["HPSDF_QUERY_LAB"] = str(lab)
        us, _ = timed(tree, rnd)
        print("  %-74s %7.1f us  (%+5.1f %%)" % (names[lab], us, (us - base) / base * 100), flush=True)
    del os.environ["HPSDF_QUERY_LAB"]
    # the same points, a wave's lanes in the same or neighbouring leaves
    cell = ((rnd + 0.5) * 64.0).floor().clamp_(0, 63).to(torch.int64)
    key = torch.zeros(n, dtype=torch.int64, device="cuda")
    for b in range(6):  # Morton order of the depth-6 cells: the leaves' own order
        for a in range(3):
            key |= ((cell[:, a] >> b) & 1) << (3 * b + a)
    srt = rnd[torch.argsort(key)].contiguous()
    us, out = timed(tree, srt)
    print("  %-74s %7.1f us  (%+5.1f %%)" % ("the points in the leaves' order (lanes share lines)", us, (us - base) / base * 100))
    os.environ["HPSDF_QUERY_LAB"] = "1"
    us, _ = timed(tree, srt)
    print("  %-74s %7.1f us  (%+5.1f %%)" % ("... and no polynomial", us, (us - base) / base * 100))
    del os.environ["HPSDF_QUERY_LAB"]
    # QueryWithGradient on the same tree (query_general_grad_kernel; 56 B a point)
    import ctypes as C
    grad = torch.empty((n, 3), dtype=torch.float64, device="cuda")
    gout = torch.empty(n, dtype=torch.float64, device="cuda")
    call = lambda: H.check(H.lib().hpsdf_query_gradient_device(ctx.handle, tree.handle, C.c_void_p(rnd.data_ptr()), n, C.c_void_p(gout.data_ptr()), C.c_void_p(grad.data_ptr())))
    torch.cuda.synchronize()
    for _ in range(2):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(10):
        call()
    e1.record(stream)
    torch.cuda.synchronize()
    gus = e0.elapsed_time(e1) / 10 * 1e3
    print("  %-74s %7.1f us  (%.3f of HBM peak at 56 B a point)" % ("QueryWithGradient, as it is", gus, 56.0 * n / (gus * 1e-6) / 8e12))

    def timed_grad(pp):
        c2 = lambda: H.check(H.lib().hpsdf_query_gradient_device(ctx.handle, tree.handle, C.c_void_p(pp.data_ptr()), n, C.c_void_p(gout.data_ptr()), C.c_void_p(grad.data_ptr())))
        torch.cuda.synchronize()
        for _ in range(2):
            c2()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record(stream)
        for _ in range(10):
            c2()
        a1.record(stream)
        torch.cuda.synchronize()
        return a0.elapsed_time(a1) / 10 * 1e3

    # the gradient kernel (query_general_grad_kernel, 4 waves a workgroup, 3 a SIMD) link by link; in the lab runs the second pass over the
    # deferred points (degree 4: ~20 us of the figure above) is not launched at all
    gnames = {5: "the value's arithmetic only (no one-sided sums, no divisions, no square root)", 6: "the six IEEE divisions and the square root left out",
              1: "no polynomial at all: the fetched rows are touched, not evaluated", 2: "every lane fetches its wave's first leaf: no gather traffic",
              3: "no second line for degree-3 leaves", 4: "no walk below the top table"}
    for lab in (5, 6, 1, 2, 3, 4):
        os.environ["HPSDF_QUERY_LAB"] = str(lab)
        us = timed_grad(rnd)
        print("  %-74s %7.1f us  (%+5.1f %%)" % ("gradient: " + gnames[lab], us, (us - gus) / gus * 100), flush=True)
    del os.environ["HPSDF_QUERY_LAB"]
    us = timed_grad(srt)
    print("  %-74s %7.1f us  (%+5.1f %%)" % ("gradient: the points in the leaves' order", us, (us - gus) / gus * 100))
    hb, _ = H.create_block(ctx, H.make_config(1e-5), H.Field.union3(), 1024)
    ht = H.DeviceTree(ctx, hb)
    us_top, _ = timed(ht, rnd)
    print("headline tree (4096 degree-2 leaves at depth 4), query_kernel (one line a point, fat table)   %7.1f us" % us_top)
