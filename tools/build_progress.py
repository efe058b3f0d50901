"""Round-by-round progress of a Create (stepwise C API): python tools/build_progress.py <field> <target> [max_rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import hpsdf_loader
H = hpsdf_loader.load()
name, target = sys.argv[1], float(sys.argv[2])
max_rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 400
field = {"sphere": H.Field.sphere, "union3": H.Field.union3}[name]()
ctx = H.Context(0)
b = H.Build(H.make_config(target), 1024, 0, 1)
t0 = time.perf_counter()
r = 0
while r < max_rounds:
    n = b.select()
    if n == 0:
        break
    jobs = b.jobs(n)
    degs = np.bincount([j.degree for j in jobs], minlength=13)
    deps = np.bincount([j.depth for j in jobs], minlength=11)
    b.compute(ctx, field)
    hdr = b.results_host(ctx)
    b.apply(hdr.reshape(n, 9))
    st = b.stats()
    if r < 10 or r % 10 == 0:
        print("round %4d: %4d jobs degrees %s depths %s | total err %.3e nodes %d p %d h %d dropped %d | %.2f s"
              % (r, n, {i: int(c) for i, c in enumerate(degs) if c}, {i: int(c) for i, c in enumerate(deps) if c}, st["total_error"],
                 st["n_nodes"], st["p_refines"], st["h_refines"], st["dropped"], time.perf_counter() - t0), flush=True)
    r += 1
print("stopped after %d rounds, %.2f s: %s" % (r, time.perf_counter() - t0, b.stats()), flush=True)
