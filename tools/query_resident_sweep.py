"""query_resident_kernel (part of the top table resident in LDS) against query_kernel on the headline tree, 10 M random points in HBM,
and on cell-sorted points; every shape must return query_kernel's bits.   usage: python tools/query_resident_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, hpsdf_loader, oracle as O
H = hpsdf_loader.load()
stream = torch.cuda.Stream()
n = 10_000_000
with torch.cuda.stream(stream):
    ctx = H.Context(0, stream.cuda_stream)
    blk, _ = H.create_block(ctx, H.make_config(1e-5), H.Field.union3(), 1024)
    tree = H.DeviceTree(ctx, blk)
    pts = torch.from_numpy(O.splitmix64_points(n)).cuda()
    cell = ((pts + 0.5) * 16.0).floor().clamp_(0, 15).to(torch.int64)
    srt = pts[torch.argsort(cell[:, 0] + 16 * cell[:, 1] + 256 * cell[:, 2])].contiguous()
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    ref = {}
    for shape in ["0", "16:0", "16:1024", "16:512", "12:0", "8:0", "8:1024"]:
        os.environ["HPSDF_QUERY_RESIDENT"] = shape
        row = []
        for name, p in (("random", pts), ("cell-sorted", srt)):
            for _ in range(3):
                tree.query_device(p.data_ptr(), n, out.data_ptr())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(20):
                tree.query_device(p.data_ptr(), n, out.data_ptr())
            e1.record(stream)
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            if shape == "0":
                ref[name] = out.clone()
            same = bool(torch.equal(out.view(torch.int64), ref[name].view(torch.int64)))
            row.append("%s %6.1f us = %5.1f Gpts/s (%.3f of HBM peak) bits %s" % (name, us, n / us / 1e3, 32 * n / us / 1e3 / 8000, "same" if same else "DIFFERENT"))
        print("HPSDF_QUERY_RESIDENT=%-8s %s" % (shape, " | ".join(row)), flush=True)
