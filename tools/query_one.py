"""A few Query launches on one tree (for profiling): python3 tools/query_one.py <target> [grad]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ctypes as C
import numpy as np, torch
import hpsdf_loader
import oracle as O
H = hpsdf_loader.load()
target = float(sys.argv[1])
grad = len(sys.argv) > 2
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    ctx = H.Context(0, stream.cuda_stream)
    blk, _ = H.create_block(ctx, H.make_config(target), H.Field.union3(), 1024)
    tree = H.DeviceTree(ctx, blk)
    n = 10_000_000
    pts = torch.from_numpy(O.splitmix64_points(n)).cuda()
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    g = torch.empty(3 * n, dtype=torch.float64, device="cuda") if grad else None
    torch.cuda.synchronize()
    for _ in range(6):
        if grad:
            H.check(H.lib().hpsdf_query_gradient_device(ctx.handle, tree.handle, C.c_void_p(pts.data_ptr()), n, C.c_void_p(out.data_ptr()), C.c_void_p(g.data_ptr())))
        else:
            tree.query_device(pts.data_ptr(), n, out.data_ptr())
    torch.cuda.synchronize()
