"""Query on the headline tree against the size of the point set: where the Infinity Cache (256 MB) stops holding the input between launches.
usage: python tools/query_sizes_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import hpsdf_loader
H = hpsdf_loader.load()
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    ctx = H.Context(0, stream.cuda_stream)
    rng = np.random.default_rng(5)
    big = torch.from_numpy(rng.random((100_000_000, 3)) - 0.5).cuda()
    out = torch.empty(100_000_000, dtype=torch.float64, device="cuda")
    for target in (1e-5, 1e-7):
        blk, _ = H.create_block(ctx, H.make_config(target), H.Field.union3(), 1024)
        tree = H.DeviceTree(ctx, blk)
        for n in (4_000_000, 8_000_000, 10_000_000, 12_000_000, 16_000_000, 24_000_000, 40_000_000, 100_000_000):
            reps = max(5, 400_000_000 // n)
            for _ in range(3):
                tree.query_device(big.data_ptr(), n, out.data_ptr())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                tree.query_device(big.data_ptr(), n, out.data_ptr())
            e1.record(stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            print("union3 @ %g: %9d points (%4d MB in, %4d MB out)  %.4f ms = %6.1f Gpts/s = %.3f of HBM peak" % (
                target, n, n * 24 >> 20, n * 8 >> 20, ms, n / ms / 1e6, 32 * n / ms / 1e6 / 8000), flush=True)
