"""Randomised continuity sweep (by hand, on a GPU box): the post-process with the solve on the device, with the solve on the
host, and as part of Create must give the same block bit for bit (and the same iteration count, residual and jump
energies), for random fields, thresholds, strengths and host thread counts; and the matrix assembled on the device must equal
the host assembly's arrays.   usage: fuzz_continuity.py [cases] [first seed]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, hpsdf_loader, oracle as O
H = hpsdf_loader.load(); ctx = H.Context(0)
bad = 0
cases, first = (int(sys.argv[1]) if len(sys.argv) > 1 else 16), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
for seed in range(first, first + cases):
    rng = np.random.default_rng(500 + seed)
    spec = []
    for k in range(int(rng.integers(1, 4))):
        kind = int(rng.integers(0, 3)); c = rng.uniform(-0.3, 0.3, 3)
        par = list(c) + ([float(rng.uniform(0.1, 0.35))] if kind == 0 else list(rng.uniform(0.05, 0.25, 3)) if kind == 1 else [float(rng.uniform(0.1, 0.25)), float(rng.uniform(0.03, 0.08))])
        spec.append((kind, 0 if k == 0 else int(rng.integers(0, 3)), [float(x) for x in par]))
    target = float(rng.choice([1e-5, 1e-6, 1e-7, 3e-8]))
    cfg0 = H.make_config(target); cfg0.continuity_strength = float(rng.choice([1.0, 8.0, 50.0]))
    cfg0.threadCount = int(rng.choice([1, 4, 16]))
    b0, st = H.create_block(ctx, cfg0, H.Field.analytic(spec), 1024)
    t0 = time.time(); dev, sd = H.continuity_post_process(bytes(b0), ctx=ctx); td = time.time() - t0
    t0 = time.time(); host, sh = H.continuity_post_process(bytes(b0)); th = time.time() - t0
    cfg1 = H.make_config(target, continuity=True); cfg1.continuity_strength = cfg0.continuity_strength; cfg1.threadCount = cfg0.threadCount
    b1, _ = H.create_block(ctx, cfg1, H.Field.analytic(spec), 1024)
    rp, col, val, _ = H.continuity_matrix(b0, 4)
    drp, dcol, dval, _ = H.continuity_matrix_device(ctx, b0)
    same_matrix = np.array_equal(rp, drp) and np.array_equal(col, dcol) and np.array_equal(val.view(np.uint64), dval.view(np.uint64))
    ok = same_matrix and dev == host and b1[:-80] == dev[:-80] and  sd["iterations"] == sh["iterations"] and sd["residual"] == sh["residual"] and sd["jump_before"] == sh["jump_before"] and sd["jump_after"] == sh["jump_after"]
    if not ok:
        nd = sum(x != y for x, y in zip(dev, host)); nb = sum(x != y for x, y in zip(dev, b1))
        print("   dev==host %s (%d bytes differ), Create==dev %s (%d bytes differ), stats it %s res %s jb %s ja %s" % (dev == host, nd, b1 == dev, nb,
              sd["iterations"] == sh["iterations"], sd["residual"] == sh["residual"], sd["jump_before"] == sh["jump_before"], sd["jump_after"] == sh["jump_after"]))
    bad += 0 if ok else 1
    print("seed %d: %d nodes target %g strength %g threads %d: %d unknowns nnz %d, %d iterations, jump %.3e -> %.3e | device == host == Create: %s (device %.1f ms, host %.1f ms)"
          % (seed, st["n_nodes"], target, cfg0.continuity_strength, cfg0.threadCount, st["n_coeffs"], sd["nnz"], sd["iterations"], sd["jump_before"], sd["jump_after"], ok, td * 1e3, th * 1e3), flush=True)
print("FAILURES:", bad)
