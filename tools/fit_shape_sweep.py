import os, sys
sys.path.insert(0, "/root/repo")
import hpsdf_loader
H = hpsdf_loader.load()
ctx = H.Context(0)
cfg = H.make_config(1e-5)
u3 = H.Field.union3(); sp = H.Field.sphere()
for g in (0, 4, 6, 8, 10, 12, 16, 20, 25):
    if g: os.environ["HPSDF_FIT_G"] = str(g)
    else: os.environ.pop("HPSDF_FIT_G", None)
    r = []
    for f in (u3, sp):
        for cells in (4096, 32768):
            r.append(H.bench_fit(ctx, cfg, f, 2, 4, cells, 10) * 1e3)
    print("HPSDF_FIT_G=%-3s p2: union3 4096 cells %.1f us, 32768 cells %.1f us | sphere 4096 cells %.1f us, 32768 cells %.1f us" % (g or "dflt", *r), flush=True)
