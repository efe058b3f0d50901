#!/usr/bin/env python3
"""One analytic build from a spec on the command line, with its statistics or the product's refusal (used to look at single fuzz seeds).
usage: python tools/one_build.py '<python list of (kind, op, params)>' '<lo>' '<hi>' <target> <K>"""
import ast, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hpsdf_loader
H = hpsdf_loader.load()
spec, lo, hi = ast.literal_eval(sys.argv[1]), ast.literal_eval(sys.argv[2]), ast.literal_eval(sys.argv[3])
target, K = float(sys.argv[4]), int(sys.argv[5])
ctx = H.Context(0)
free0, total = torch.cuda.mem_get_info()
t0 = time.time()
try:
    blk, st = H.create_block(ctx, H.make_config(target, lo, hi), H.Field.analytic(spec), K)
    print("built: %d bytes, %s, %.1f s" % (len(blk), {k: st[k] for k in ("n_nodes", "n_leaves", "n_coeffs", "rounds", "jobs")}, time.time() - t0))
except H.HpsdfError as e:
    free1, _ = torch.cuda.mem_get_info()
    print("refused after %.1f s: %s; device memory free before %.1f GB, at the refusal %.1f GB of %.1f GB" % (time.time() - t0, e, free0 / 1e9, free1 / 1e9, total / 1e9))
