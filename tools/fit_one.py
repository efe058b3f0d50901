"""One fit micro-benchmark launch set (for profiling): python3 tools/fit_one.py <field> <degree> <cells> [exact|split|fast]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hpsdf_loader
H = hpsdf_loader.load()
name, p, cells = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f = {"union3": H.Field.union3, "sphere": H.Field.sphere,
     "plane": lambda: H.Field.analytic([(H.PRIM_PLANE, 0, [0.3, -0.2, 0.5, 0.1])])}[name]()
ctx = H.Context(0)
if len(sys.argv) > 4:  # exact | split (the default) | fast
    ctx.set_fit_mode({"exact": H.FIT_EXACT, "split": H.FIT_SPLIT, "default": H.FIT_SPLIT, "fast": H.FIT_FAST}[sys.argv[4]])
ms = H.bench_fit(ctx, H.make_config(1e-5), f, p, 5, cells, 5)
print("%s p=%d cells=%d: %.1f us/launch, %.2f TFLOP/s algorithmic" % (name, p, cells, ms * 1e3, 2.0 * H.NCOEF[p] * (4 * p + 1) ** 3 * cells / ms / 1e9))
