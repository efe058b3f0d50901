import sys
sys.path.insert(0, "/root/repo")
import hpsdf_loader
H = hpsdf_loader.load()
ctx = H.Context(0)
cfg = H.make_config(1e-5)
plane = H.Field.analytic([(H.PRIM_PLANE, H.OP_UNION, [0.3, -0.2, 0.5, 0.1])])
u3 = H.Field.union3()
for p in (2, 3, 4, 5):
    cells = 65536 if p <= 3 else 16384
    flops = 2.0 * H.NCOEF[p] * (4 * p + 1) ** 3 * cells
    print("p%d: union3 %.2f TF, plane %.2f TF" % (p, flops / H.bench_fit(ctx, cfg, u3, p, 5, cells, 3) / 1e9, flops / H.bench_fit(ctx, cfg, plane, p, 5, cells, 3) / 1e9), flush=True)
