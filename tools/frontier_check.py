#!/usr/bin/env python3
"""Device-side frontier (csrc/frontier.hip) against the host scheduler (csrc/builder.cpp): same MemoryBlock byte for byte,
and the time of both.  usage: python tools/frontier_check.py [--mesh]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hpsdf_loader
from helpers import icosphere, displaced_torus
H = hpsdf_loader.load()
ctx = H.Context(0)


def run(name, cfg, field, K, reps=5):
    out = {}
    for mode in ("host", "device"):
        os.environ["HPSDF_HOST_FRONTIER"] = "1" if mode == "host" else "0"
        blk, st = H.create_block(ctx, cfg, field, K)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            blk, st = H.create_block(ctx, cfg, field, K)
            ts.append((time.perf_counter() - t0) * 1e3)
        out[mode] = (blk, st, float(np.median(ts)))
    same = out["host"][0] == out["device"][0]
    sh, sd = out["host"][1], out["device"][1]
    keys = ("rounds", "jobs", "p_refines", "h_refines", "dropped", "fits", "samples", "n_nodes", "n_leaves", "n_coeffs")
    stats_same = all(sh[k] == sd[k] for k in keys) and sh["total_error"] == sd["total_error"]
    print("%-34s K=%-5d nodes %-7d rounds %-3d host %8.3f ms  device %8.3f ms  block %s  stats %s" % (
        name, K, sd["n_nodes"], sd["rounds"], out["host"][2], out["device"][2], "identical" if same else "DIFFERS",
        "identical" if stats_same else "DIFFER %s vs %s" % ({k: sh[k] for k in keys}, {k: sd[k] for k in keys})), flush=True)
    return same and stats_same


ok = True
ok &= run("C1 sphere 1e-4", H.make_config(1e-4), H.Field.sphere(), 1024)
ok &= run("C2 union3 1e-5", H.make_config(1e-5), H.Field.union3(), 1024)
ok &= run("union3 1e-6", H.make_config(1e-6), H.Field.union3(), 1024)
ok &= run("A1 union3 1e-7", H.make_config(1e-7), H.Field.union3(), 1024)
ok &= run("A1 union3 1e-7", H.make_config(1e-7), H.Field.union3(), 256)
ok &= run("A2 sphere 1e-8", H.make_config(1e-8), H.Field.sphere(), 1024)
ok &= run("union3 1e-8", H.make_config(1e-8), H.Field.union3(), 1024, reps=3)
ok &= run("union3 1e-8", H.make_config(1e-8), H.Field.union3(), 4096, reps=3)
ok &= run("union3 1e-8", H.make_config(1e-8), H.Field.union3(), 64, reps=1)
ok &= run("sphere075 root[-.25,5] 1e-6", H.make_config(1e-6, (-0.25,) * 3, (5.0,) * 3), H.Field.sphere((0.25, 0, 0), 0.75), 1024)
if "--mesh" in sys.argv:
    v, t = displaced_torus(12, 8)
    ok &= run("torus 192 tris 1e-6", H.make_config(1e-6, (-0.45, -0.45, -0.2), (0.45, 0.45, 0.2)), H.Field.mesh(ctx, v, t), 256)
    v, t = icosphere(7, 0.4)
    lo, hi = v.min(0) - 0.02, v.max(0) + 0.02
    f = H.Field.mesh(ctx, v, t)
    ok &= run("icosphere L7 1e-5", H.make_config(1e-5, tuple(lo), tuple(hi)), f, 1024, reps=3)
    ok &= run("icosphere L7 1e-6", H.make_config(1e-6, tuple(lo), tuple(hi)), f, 1024, reps=2)
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
