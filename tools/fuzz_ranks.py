"""Randomised sweep of hpsdf_create_distributed on simulated ranks (threads on ONE GPU, tools/frontier_ranks_check.py's exchange):
random analytic CSG fields, root boxes, thresholds, round sizes, weightings and world sizes 2..4 -- every rank's block against the
single-rank block byte for byte (which tools/fuzz_parity.py compares with the oracle's), statistics too.  Weighted builds take the
device frontier's "replica" mode (one more all-gather a round).  Usage: python tools/fuzz_ranks.py [cases] [first seed]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import hpsdf_loader
H = hpsdf_loader.load()


class DevPtr:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def run_world(world, cfg, spec, K):
    ctxs = [H.Context(0) for _ in range(world)]
    fields = [H.Field.analytic(spec) for _ in ctxs]
    barrier = threading.Barrier(world)
    bufs, out, errs = [None] * world, [None] * world, []

    def gather_for(rank):
        def gather(d_buf, nbytes, stream):
            ctxs[rank].synchronize()
            bufs[rank] = d_buf
            barrier.wait()
            mine = torch.as_tensor(DevPtr(d_buf, nbytes * world), device="cuda")
            for r in range(world):
                if r != rank:
                    other = torch.as_tensor(DevPtr(bufs[r], nbytes * world), device="cuda")
                    mine[r * nbytes:(r + 1) * nbytes].copy_(other[r * nbytes:(r + 1) * nbytes])
            torch.cuda.synchronize()
            barrier.wait()
        return gather

    def worker(rank):
        try:
            out[rank] = H.create_block_distributed(ctxs[rank], cfg, fields[rank], K, rank, world, gather_for(rank))
        except BaseException as e:  # noqa: BLE001
            errs.append((rank, e))
            barrier.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0][1]
    return out


cases, first = (int(sys.argv[1]) if len(sys.argv) > 1 else 24), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = H.Context(0)
bad = 0
keys = ("rounds", "jobs", "p_refines", "h_refines", "dropped", "fits", "samples", "n_nodes", "n_leaves", "n_coeffs", "total_error")
for seed in range(first, first + cases):
    rng = np.random.default_rng(seed)
    spec = []
    for k in range(int(rng.integers(1, 4))):
        kind = int(rng.integers(0, 4))
        c = rng.uniform(-0.3, 0.3, 3)
        if kind == H.PRIM_SPHERE:
            par = list(c) + [float(rng.uniform(0.08, 0.35))]
        elif kind == H.PRIM_BOX:
            par = list(c) + list(rng.uniform(0.05, 0.25, 3))
        elif kind == H.PRIM_TORUS_Y:
            par = list(c) + [float(rng.uniform(0.1, 0.25)), float(rng.uniform(0.03, 0.08))]
        else:
            nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
            par = list(nrm) + [float(rng.uniform(-0.2, 0.2))]
        spec.append((kind, H.OP_UNION if k == 0 else int(rng.integers(0, 3)), [float(x) for x in par]))
    lo = tuple(float(x) for x in (-0.5 + rng.uniform(-0.2, 0.2, 3)).astype(np.float32))
    hi = tuple(float(x) for x in (0.5 + rng.uniform(-0.2, 0.3, 3)).astype(np.float32))
    target = float(rng.choice([1e-5, 1e-6, 3e-7, 1e-7, 3e-8]))
    K = int(rng.choice([64, 256, 1024, 4096]))
    wtype = int(rng.choice([0, 1, 2, 2]))
    world = int(rng.integers(2, 5))
    cfg = H.make_config(target, lo, hi)
    if wtype:
        cfg.nearnessWeighting_type, cfg.nearnessWeighting_strength = wtype, float(rng.choice([2.0, 3.0]))
    t0 = time.time()
    try:
        one, st = H.create_block(ctx, cfg, H.Field.analytic(spec), K)
    except H.HpsdfError as e:
        print("seed %d: refused on one rank (%s)" % (seed, e)); continue
    try:
        res = run_world(world, cfg, spec, K)
    except H.HpsdfError as e:
        bad += 1
        print("seed %d: %d ranks FAILED where one rank built (%s)" % (seed, world, e)); continue
    same = all(b == one for b, _ in res)
    # (a rank counts the fits and samples of its own slices: their sums over the ranks are the single-rank figures)
    stats = all(all(s[k] == st[k] for k in keys if k not in ("fits", "samples")) for _, s in res) and \
        all(sum(s[k] for _, s in res) == st[k] for k in ("fits", "samples"))
    if not stats:
        print("   differing:", [(k, st[k], [s[k] for _, s in res]) for k in keys if any(s[k] != st[k] for _, s in res)])
    frontier = [s["device_frontier"] for _, s in res]
    ok = same and stats
    bad += 0 if ok else 1
    print("seed %3d: %d prims w%d target %g K %4d world %d -> %5d nodes, %3d rounds | blocks %s stats %s device frontier %s  (%.1f s)"
          % (seed, len(spec), wtype, target, K, world, st["n_nodes"], st["rounds"], same, stats, frontier, time.time() - t0), flush=True)
print("FAILURES: %d of %d" % (bad, cases))
sys.exit(1 if bad else 0)
