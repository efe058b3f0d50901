"""Development probe run on the GPU box: parity of field / fit / query against the oracle + rough timings."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import hpsdf_loader, oracle as O
H = hpsdf_loader.load()

ctx = H.Context(0)
print("ctx ok")
# 1. field parity (exercises device sqrt)
pts = O.splitmix64_points(300000, seed=7)
for name, hf, of in (("sphere", H.Field.sphere(), O.sphere_field()), ("union3", H.Field.union3(), O.union3_field())):
    g = hf.eval(ctx, pts); w = of.eval(pts)
    nz = np.sum(g.view(np.uint64) != w.view(np.uint64))
    print("field", name, "bit mismatches", nz, "max abs diff", np.abs(g - w).max())
# 2. create parity
for name, hf, of, tg, K in (("C1", H.Field.sphere(), O.sphere_field(), 1e-4, 1024), ("C2", H.Field.union3(), O.union3_field(), 1e-5, 1024),
                            ("A2", H.Field.sphere(), O.sphere_field(), 1e-8, 1024), ("A1", H.Field.union3(), O.union3_field(), 1e-7, 1024)):
    cfg = H.make_config(tg)
    t0 = time.time(); blk, st = H.create_block(ctx, cfg, hf, K); t1 = time.time()
    blk2, st2 = H.create_block(ctx, cfg, hf, K); t2 = time.time()
    ob = O.Tree.create(O.default_config(tg), of, K).to_block(); t3 = time.time()
    same = blk == ob
    msg = ""
    if not same and len(blk) == len(ob):
        a = O.parse_block(blk); b = O.parse_block(ob)
        msg = " coeff maxdiff %g nodes_equal %s" % (np.abs(a["coeffs"] - b["coeffs"]).max(), np.array_equal(a["degree"], b["degree"]))
    print(name, "block identical:", same, len(blk), len(ob), msg, "gpu create %.1f ms (first %.1f ms) oracle %.0f ms" % ((t2 - t1) * 1e3, (t1 - t0) * 1e3, (t3 - t2) * 1e3), st)
    # 3. query parity
    q = O.splitmix64_points(200000)
    q[:100] *= 3.0
    tr = H.DeviceTree(ctx, ob)
    g = tr.query(q); w = O.Tree.from_block(ob).query(q)
    print("   query bit mismatches", np.sum(g.view(np.uint64) != w.view(np.uint64)), "max abs", np.abs(g - w)[np.isfinite(g - w)].max())
# 4. query throughput, device resident
import torch
n = 10_000_000
x = torch.from_numpy(O.splitmix64_points(n)).cuda()
out = torch.empty(n, dtype=torch.float64, device="cuda")
ctx2 = H.Context(0, torch.cuda.current_stream().cuda_stream)
blk, _ = H.create_block(ctx2, H.make_config(1e-5), H.Field.union3(), 1024)
tr = H.DeviceTree(ctx2, blk)
for _ in range(3): tr.query_device(x.data_ptr(), n, out.data_ptr())
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): tr.query_device(x.data_ptr(), n, out.data_ptr())
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("query 10M pts: %.3f ms -> %.1f Mpts/s, %.1f GB/s algorithmic" % (ms, n / ms / 1e3, n * 32 / ms / 1e6))
# 5. fit microbench
for p in (2, 3, 4, 5):
    ms = H.bench_fit(ctx, H.make_config(1e-5), H.Field.union3(), p, 5, 32768, 3)
    nq = 4 * p + 1
    fl = 2 * H.NCOEF[p] * nq ** 3 * 32768
    print("fit p=%d 32768 cells: %.3f ms  -> %.2f TFLOP/s algorithmic, %.1f Mfits/s" % (p, ms, fl / ms / 1e9, 32768 / ms / 1e3))
