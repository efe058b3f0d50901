import os, sys
sys.path.insert(0, "/root/repo")
import hpsdf_loader
H = hpsdf_loader.load()
fast = H.Context(0); fast.set_fast_fit(True)
cfg = H.make_config(1e-5)
plane = H.Field.analytic([(H.PRIM_PLANE, H.OP_UNION, [0.3, -0.2, 0.5, 0.1])])
u3 = H.Field.union3()
for p in (4, 5, 6, 7, 8, 9):
    for cells in (1024, 4096, 16384, 65536 if p <= 5 else 32768):
        flops = 2.0 * H.NCOEF[p] * (4 * p + 1) ** 3 * cells
        ms = H.bench_fit(fast, cfg, plane, p, 5, cells, 3)
        ms2 = H.bench_fit(fast, cfg, u3, p, 5, cells, 3)
        print("p%d %6d cells: plane %.2f TF (%.1f %%)   union3 %.2f TF (%.1f %%)" % (p, cells, flops / ms / 1e9, 100 * flops / ms / 1e9 / 78.6, flops / ms2 / 1e9, 100 * flops / ms2 / 1e9 / 78.6), flush=True)
