#!/usr/bin/env python3
"""The fit micro-benchmark of bench.py for the degrees that have more than one kernel: 16 384 depth-5 cells, union3 field, exact (every
row on fit_kernel) and default (split: top-degree rows exact, the rows below by fit_low_kernel) modes.  usage: fit_modes_bench.py [degrees]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hpsdf_loader
H = hpsdf_loader.load()
ctx = H.Context(0)
exact = H.Context(0)
exact.set_fit_mode(H.FIT_EXACT)
cfg = H.make_config(1e-5)
u3 = H.Field.union3()
plane = H.Field.analytic([(H.PRIM_PLANE, H.OP_UNION, [0.3, -0.2, 0.5, 0.1])])
degs = [int(a) for a in sys.argv[1:]] or [4, 5, 6, 7, 8]
for p in degs:
    cells = 65536 if p <= 3 else 16384
    flops = 2.0 * H.NCOEF[p] * (4 * p + 1) ** 3 * cells
    e, d = H.bench_fit(exact, cfg, u3, p, 5, cells, 3), H.bench_fit(ctx, cfg, u3, p, 5, cells, 3)
    ec = H.bench_fit(exact, cfg, plane, p, 5, cells, 3)
    print("p%d  exact %7.2f ms (%5.2f TF = %4.1f %% of 78.6)   default %7.2f ms (%5.2f TF = %4.1f %%)   exact, contraction only %7.2f ms"
          % (p, e, flops / e / 1e9, flops / e / 1e9 / 0.786, d, flops / d / 1e9, flops / d / 1e9 / 0.786, ec), flush=True)
