"""Query on the headline tree vs number of points."""
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
    blk, _ = H.create_block(ctx, H.make_config(1e-5), H.Field.union3(), 1024)
    tree = H.DeviceTree(ctx, blk)
    allp = torch.from_numpy(O.splitmix64_points(16_000_000)).cuda()
    out = torch.empty(16_000_000, dtype=torch.float64, device="cuda")
    cell = ((allp + 0.5) * 16.0).floor().clamp_(0, 15).to(torch.int64)
    srt = allp[torch.argsort(cell[:, 0] * 256 + cell[:, 1] * 16 + cell[:, 2])].contiguous()
    del cell
    for rep in range(3):
      for dd in ("1", "0"):
        os.environ["HPSDF_QUERY_DEDUPE"] = dd
        for n in (10_000_000, 16_000_000):
          for pname, pts in (("random", allp), ("sorted", srt)):
            torch.cuda.synchronize()
            for _ in range(3):
                tree.query_device(pts.data_ptr(), n, out.data_ptr())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(20):
                tree.query_device(pts.data_ptr(), n, out.data_ptr())
            e1.record(stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            print("dedupe=%s %-7s n=%9d: %7.1f us = %6.1f Gpts/s (%.3f of HBM peak)" % (dd, pname, n, ms * 1e3, n / ms / 1e6, 32 * n / ms / 1e6 / 8000), flush=True)
