"""Query for three point orders: random, 230^3 grid (z fastest), cell-sorted -- 12 M points each, so that points +
results (384 MB) do not fit the 256 MB Infinity Cache (8 M-point sets are served from it between launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import hpsdf_loader
import oracle as O
H = hpsdf_loader.load()
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    ctx = H.Context(0, stream.cuda_stream)
    for name, target in (("C2 union3 1e-5", 1e-5), ("A1 union3 1e-7", 1e-7)):
        blk, _ = H.create_block(ctx, H.make_config(target), H.Field.union3(), 1024)
        tree = H.DeviceTree(ctx, blk)
        n = 12_000_000
        rnd = torch.from_numpy(O.splitmix64_points(n)).cuda()
        g = torch.linspace(-0.5, 0.5, 230, dtype=torch.float64, device="cuda")
        grid = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).contiguous()
        cell = ((rnd + 0.5) * 16.0).floor().clamp_(0, 15).to(torch.int64)
        srt = rnd[torch.argsort(cell[:, 0] * 256 + cell[:, 1] * 16 + cell[:, 2])].contiguous()
        for pname, pts in (("random", rnd), ("230^3 grid", grid), ("cell-sorted", srt)):
            m = len(pts)
            out = torch.empty(m, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            for _ in range(2):
                tree.query_device(pts.data_ptr(), m, out.data_ptr())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(10):
                tree.query_device(pts.data_ptr(), m, out.data_ptr())
            e1.record(stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print("%-16s %-12s %8d pts: %7.1f us = %6.1f Gpts/s (%.2f of HBM peak)" % (name, pname, m, ms * 1e3, m / ms / 1e6, 32 * m / ms / 1e6 / 8000), flush=True)
