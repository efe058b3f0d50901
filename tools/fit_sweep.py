import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r)
import hpsdf_loader
H = hpsdf_loader.load()
ctx = H.Context(0)
cfg = H.make_config(1e-5)
for name, f in (("union3", H.Field.union3()), ("sphere", H.Field.sphere()), ("plane", H.Field.analytic([(H.PRIM_PLANE, 0, [0.3, -0.2, 0.5, 0.1])]))):
    for p, cells in ((2, 4096), (2, 65536), (3, 4096), (3, 32768), (4, 4096)):
        ms = H.bench_fit(ctx, cfg, f, p, 5, cells, 5)
        print("G=%%s %%-7s p=%%d cells=%%6d : %%8.1f us  %%6.2f TFLOP/s alg" %% (os.environ.get("HPSDF_FIT_G", "auto"), name, p, cells, ms * 1e3, 2.0 * H.NCOEF[p] * (4 * p + 1) ** 3 * cells / ms / 1e9))
''' % ROOT
for g in ("", "1", "2", "4", "8", "12", "25"):
    env = dict(os.environ)
    if g:
        env["HPSDF_FIT_G"] = g
    else:
        env.pop("HPSDF_FIT_G", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(r.stdout, r.stderr[-500:] if r.returncode else "")
