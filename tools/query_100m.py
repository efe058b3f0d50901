"""SURVEY 8(d) steady state: Query over 100 M points (3.2 GB of points + results in HBM) on the headline tree and on
union3 @ 1e-7; a sample of the results is checked against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import hpsdf_loader
import oracle as O
H = hpsdf_loader.load()
N = 100_000_000
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    ctx = H.Context(0, stream.cuda_stream)
    base = torch.from_numpy(O.splitmix64_points(10_000_000)).cuda()
    # 100 M distinct points: ten shifted copies of the 10 M SplitMix64 set, wrapped into the root
    pts = torch.cat([((base + 0.5 + 0.0371 * k) % 1.0) - 0.5 for k in range(10)]).contiguous()
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    for target in (1e-5, 1e-7):
        blk, _ = H.create_block(ctx, H.make_config(target), H.Field.union3(), 1024)
        tree = H.DeviceTree(ctx, blk)
        for _ in range(2):
            tree.query_device(pts.data_ptr(), N, out.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5):
            tree.query_device(pts.data_ptr(), N, out.data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        idx = torch.arange(0, N, N // 5000, device="cuda")
        want = O.Tree.from_block(blk).query(pts[idx].cpu().numpy())
        ok = np.array_equal(out[idx].cpu().numpy(), want)
        print("union3 @ %g: %d points %.3f ms = %.1f Gpts/s (%.3f of HBM peak), sample of %d bit-identical to the oracle: %s"
              % (target, N, ms, N / ms / 1e6, 32 * N / ms / 1e6 / 8000, len(idx), ok), flush=True)
