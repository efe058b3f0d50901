#!/usr/bin/env python3
"""hpsdf_create_distributed with N ranks simulated on ONE GPU: N threads, one context each, the all-gather callback is a
thread barrier plus device-to-device copies between the ranks' buffers.  Every rank's block must equal the world-1 block
byte for byte.  usage: python tools/frontier_ranks_check.py [--mesh]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import hpsdf_loader
from helpers import displaced_torus
H = hpsdf_loader.load()


class DevPtr:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def run_world(world, cfg, make_field, K):
    ctxs = [H.Context(0) for _ in range(world)]
    fields = [make_field(c) for c in ctxs]
    barrier = threading.Barrier(world)
    bufs = [None] * world
    out = [None] * world
    errs = []

    def gather_for(rank):
        def gather(d_buf, nbytes, stream):
            ctxs[rank].synchronize()          # this rank's part is complete
            bufs[rank] = d_buf
            barrier.wait()
            mine = torch.as_tensor(DevPtr(d_buf, nbytes * world), device="cuda")
            for r in range(world):
                if r != rank:
                    other = torch.as_tensor(DevPtr(bufs[r], nbytes * world), device="cuda")
                    mine[r * nbytes:(r + 1) * nbytes].copy_(other[r * nbytes:(r + 1) * nbytes])
            torch.cuda.synchronize()
            barrier.wait()                    # nobody overwrites a buffer somebody is still reading
        return gather

    def worker(rank):
        try:
            out[rank] = H.create_block_distributed(ctxs[rank], cfg, fields[rank], K, rank, world, gather_for(rank))
        except BaseException as e:  # noqa: BLE001
            errs.append((rank, e))
            barrier.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0][1]
    return out, (time.perf_counter() - t0) * 1e3


def timed_worlds(name, cfg, make_field, K, worlds=(2, 4), reps=5):
    """Median time of a Create on `world` simulated ranks (contexts, fields and threads made once; every repetition starts behind a
    barrier) beside the single-rank time; blocks compared with the single-rank block."""
    ctx = H.Context(0)
    f1 = make_field(ctx)
    one, st = H.create_block(ctx, cfg, f1, K)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        H.create_block(ctx, cfg, f1, K)
        ts.append((time.perf_counter() - t0) * 1e3)
    base = float(np.median(ts))
    print("%-40s K=%-5d one rank %.3f ms (%d rounds, device frontier %d)" % (name, K, base, st["rounds"], st["device_frontier"]), flush=True)
    ok = True
    for world in worlds:
        ctxs = [H.Context(0) for _ in range(world)]
        fields = [make_field(c) for c in ctxs]
        barrier = threading.Barrier(world)
        bufs = [None] * world
        times = [[] for _ in range(world)]
        same = [True] * world
        frontier = [None] * world
        errs = []

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
                g = gather_for(rank)
                for rep in range(reps + 1):
                    barrier.wait()
                    t0 = time.perf_counter()
                    blk, stw = H.create_block_distributed(ctxs[rank], cfg, fields[rank], K, rank, world, g)
                    if rep:
                        times[rank].append((time.perf_counter() - t0) * 1e3)
                    same[rank] &= blk == one
                    frontier[rank] = (stw["device_frontier"], stw.get("exchanges"))
            except BaseException as e:  # noqa: BLE001
                errs.append((rank, e))
                barrier.abort()

        th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0][1]
        ms = max(float(np.median(t)) for t in times)
        ok &= all(same)
        print("%-40s K=%-5d world %d: %.3f ms = %.2f x one rank (slowest rank's median of %d; simulated exchange: thread barrier + device copies); %s; (device frontier, exchanges) %s"
              % (name, K, world, ms, ms / base, reps, "identical on every rank" if all(same) else "DIFFERS", frontier[0]), flush=True)
    return ok


def check(name, cfg, make_field, K):
    ctx = H.Context(0)
    one, st = H.create_block(ctx, cfg, make_field(ctx), K)
    ok = True
    for world in (2, 4, 8):
        res, ms = run_world(world, cfg, make_field, K)
        same = all(b == one for b, _ in res)
        ok &= same
        print("%-28s K=%-5d world %d: %s (%d rounds, %d nodes, %.1f ms with the simulated exchange)" % (
            name, K, world, "identical on every rank" if same else "DIFFERS", res[0][1]["rounds"], res[0][1]["n_nodes"], ms), flush=True)
    return ok


def weighted_config(target, wtype, strength):  # wtype: 1 Polynomial, 2 Exponential (Config::NearnessWeighting)
    cfg = H.make_config(target)
    cfg.nearnessWeighting_type, cfg.nearnessWeighting_strength = wtype, strength
    return cfg


if "--weighted" in sys.argv:
    ok = True
    for host in ("0", "1"):
        os.environ["HPSDF_HOST_FRONTIER"] = host
        tag = "host scheduler" if host == "1" else "device frontier"
        if host == "0":  # (the harness itself: the same tree without weights -- one exchange a round instead of two)
            ok &= timed_worlds("sphere 1e-8 unweighted, " + tag, H.make_config(1e-8), lambda c: H.Field.sphere(), 1024)
        ok &= timed_worlds("sphere 1e-8 Exponential(3), " + tag, weighted_config(1e-8, 2, 3.0), lambda c: H.Field.sphere(), 1024)
        ok &= timed_worlds("sphere 1e-10 Exponential(3), " + tag, weighted_config(1e-10, 2, 3.0), lambda c: H.Field.sphere(), 1024)
        ok &= timed_worlds("union3 1e-7 Polynomial(2), " + tag, weighted_config(1e-7, 1, 2.0), lambda c: H.Field.union3(), 1024)
    print("ALL IDENTICAL" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
ok = True
ok &= check("C2 union3 1e-5", H.make_config(1e-5), lambda c: H.Field.union3(), 1024)
ok &= check("A1 union3 1e-7", H.make_config(1e-7), lambda c: H.Field.union3(), 1024)
ok &= check("A1 union3 1e-7", H.make_config(1e-7), lambda c: H.Field.union3(), 256)
ok &= check("A2 sphere 1e-8", H.make_config(1e-8), lambda c: H.Field.sphere(), 1024)
if "--mesh" in sys.argv:
    v, t = displaced_torus(12, 8)
    ok &= check("torus mesh 1e-6", H.make_config(1e-6, (-0.45, -0.45, -0.2), (0.45, 0.45, 0.2)), lambda c: H.Field.mesh(c, v, t), 256)
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
