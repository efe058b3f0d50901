#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json config[1] -- union of sphere/box/torus, targetError = 1e-5,
Create() on the GPU(s) and Query() over 10 M SplitMix64(12345) points per GPU, resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one batched Query() pass over this rank's 10 M points (the data-parallel leg of the
metric; weak scaling: every rank queries its own 10 M points against the replicated tree, no
collective).  Create() -- sharded over the ranks with one RCCL all-gather per round -- is timed in the
same run and reported as create_ms next to it.  Rank 0 prints ONE JSON line.

roofline: the dominant kernel is query_kernel, HBM-bound, 32 algorithmic bytes per point
(24 in + 8 out); its average launch duration is measured live with events on the
stream it is launched on.  cpu_baseline: the CPU oracle ("port", one core) on the same points,
rank 0 at N = 1 only -- the only leg that touches oracle/ (it also checks a sample of the timed GPU outputs
against it); the oracle is never the measured path.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6  # MI355X datasheet, vector = matrix FP64
N_POINTS = 10_000_000
TARGET = 1e-5
JOBS_PER_ROUND = 1024
CREATE_REPS = 21  # Create() repetitions behind create_ms (their median)


def splitmix64_points(n, seed=12345):
    """SURVEY 8(d)'s query set: SplitMix64(seed) -> (u >> 11) * 2^-53 - 0.5, xyz interleaved: n uniform points of the unit root box."""
    idx = np.arange(1, 3 * n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53 - 0.5).reshape(n, 3)


def counter_record(name, keys):
    """The counter figures of profiles/<name> (tools/pmc_json.py) if the kernels' source still hashes to what was profiled, else the
    same keys as null -- a secondary roofline never shows counters of a kernel that has changed since."""
    import hashlib
    out = {k: None for k in keys}
    out["measured_from"] = None
    path = os.path.join(ROOT, "profiles", name)
    try:
        rec = json.load(open(path))
        files = {"mesh_pmc.json": ["kernels.hip", "device_types.hpp", "acosf_host_libm.hpp"],
                 "fit_pmc.json": ["kernels.hip", "fit_low.hip", "field_eval.hpp", "device_types.hpp"],
                 "fit_mfma_pmc.json": ["fit_mfma.hip", "field_eval.hpp", "device_types.hpp"]}[name]
        h = hashlib.sha256()
        for f in files:
            h.update(open(os.path.join(ROOT, "hp-adaptive-signed-distance-field-octree_amd", "csrc", f), "rb").read())
        cur = h.hexdigest()[:16]
        out["measured_from"] = {"file": "profiles/" + name, "profile": rec.get("profile"), "source_sha16": rec.get("source_sha16"),
                                "current_source_sha16": cur, "stale": rec.get("source_sha16") != cur}
        if rec.get("source_sha16") == cur:
            for k in keys:
                out[k] = rec.get(k)
    except Exception:  # noqa: BLE001  (no record: nulls)
        pass
    return out


def visible_gpus():
    """GPUs this process may use, counted WITHOUT a call into the HIP runtime (the launcher must not have initialised the GPU when it
    starts the ranks): the visibility variables if they are set, else the KFD topology's nodes with compute units; torch's own count
    (which can go through hipGetDeviceCount on builds without amdsmi) only if neither is there."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    try:
        import glob
        n = 0
        for props in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            for line in open(props):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
        if n:
            return n
    except Exception:  # noqa: BLE001
        pass
    return torch.cuda.device_count()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run --nproc-per-node N bench.py <same
    arguments>` as a child process (one rank per GPU, rendezvous on 127.0.0.1), pass rank 0's JSON line through and return the
    child's exit code.  The parent makes no GPU call before or after (visible_gpus() reads the environment and sysfs)."""
    import socket
    import subprocess
    have = visible_gpus()
    if have < n and os.environ.get("HPSDF_BENCH_SHARE_GPU") != "1":
        print("bench.py: --gpus %d but %d GPU(s) visible (HPSDF_BENCH_SHARE_GPU=1 rehearses the N-rank path on one GPU over gloo)"
              % (n, have), file=sys.stderr)
        return 2
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's peer buffers need it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in child.stdout:  # the ranks print nothing but rank 0's line; anything else is passed on to stderr
        if out.startswith('{"metric"'):
            line = out
        else:
            sys.stderr.write(out)
    rc = child.wait()
    if line is not None:
        sys.stdout.write(line)
        sys.stdout.flush()
    elif rc == 0:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--points", type=int, default=N_POINTS, help="query points per GPU")
    ap.add_argument("--batches", type=int, default=4, help="distinct point batches the timed Query steps walk through (1 = the same "
                    "batch every step, which the 256 MB Infinity Cache then holds)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fit-bench", action="store_true", help="skip the steady-state fit micro-benchmark (fit_microbench)")
    ap.add_argument("--no-refined", action="store_true",
                    help="skip the extra Query / QueryWithGradient timings on the refined tree (union3 @ 1e-7)")
    ap.add_argument("--mesh", default="torus",
                    help="Create() on a mesh field (BASELINE configs 2-4 shape; dragon.obj / Ramesses.obj are not in the reference "
                         "mount): 'torus' = displaced torus grid, 2 097 152 triangles (the north_star's 2 M-triangle mesh); "
                         "an integer L = bumpy icosphere of subdivision level L (7 = 327 680, 8 = 1 310 720); 'none' = skip")
    ap.add_argument("--no-sorted-ceiling", action="store_true",
                    help="skip Query on the same points sorted by depth-4 cell (locality ceiling, SURVEY 8d)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started bare (`python bench.py --gpus N`): this process becomes the launcher.  Nothing here has touched the GPU
        # yet (torch is imported, no device call made), and the ranks are CHILD processes -- never an exec.
        raise SystemExit(self_launch(args.gpus))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher's rank count and --gpus must agree" % (args.gpus, world))
    # functional dry-run of the N > 1 path on a one-GPU box: every rank on device 0, exchange over gloo
    share_gpu = os.environ.get("HPSDF_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            import datetime
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(minutes=5))
    backend = dist.get_backend() if world > 1 else None

    import hpsdf_loader
    H = hpsdf_loader.load()
    import importlib
    D = importlib.import_module("hpsdf_amd.distributed")

    stream = torch.cuda.Stream()  # the kernels launch on this stream; torch events on it time them
    with torch.cuda.stream(stream):
        ctx = H.Context(local, stream.cuda_stream)
        cfg = H.make_config(TARGET)
        field = H.Field.union3()

        # ---------------- Create(): sharded over the ranks, timed over a few repetitions
        # N > 1: create_ms is the policy a user gets by default ("auto": an analytic field is cheap, so every rank
        # builds the whole tree and nothing is exchanged); create_sharded_ms forces the sharded frontier with its
        # all-gather per round (the mesh-field path), whose fixed costs exceed this 0.3 ms build.
        def timed_create(policy):
            def create():
                return D.create_distributed(ctx, cfg, field, JOBS_PER_ROUND, policy=policy) if world > 1 else \
                    H.create_block(ctx, cfg, field, JOBS_PER_ROUND)
            blk, st = create()  # warm-up (also first hipMalloc of the arena)
            times = []
            # Create() hands back the finished block in host memory: the call's own wall time is the figure.  One rank: the calls
            # follow each other directly, as a caller's loop would (what a build leaves on the stream for the next one -- the reset
            # of the frontier's tables -- is then inside the next call's time, not hidden behind a synchronize); several ranks
            # start every repetition together.
            for _ in range(CREATE_REPS if world == 1 else 5):
                if world > 1:
                    dist.barrier()
                    torch.cuda.synchronize()
                t0 = time.perf_counter()
                blk, st = create()
                times.append((time.perf_counter() - t0) * 1e3)
            torch.cuda.synchronize()
            ms = float(np.median(times))
            per_rank = [ms]
            if world > 1:  # every rank's own median; the job's figure is the slowest rank's
                t = torch.zeros(world, device="cuda", dtype=torch.float64)
                t[rank] = ms
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                per_rank = [float(x) for x in t.tolist()]
                ms = max(per_rank)
            return blk, st, ms, times, per_rank

        block, stats, create_ms, create_times, _ = timed_create("auto")
        create_sharded_ms, create_sharded = None, None

        # ---------------- Query(): this rank's points, resident in HBM
        n = args.points
        pts = splitmix64_points(n, seed=12345 + rank)
        d_xyz = torch.from_numpy(pts).cuda()
        # The timed steps walk through several DISTINCT batches of the same shape: one batch of 10 M points is 240 MB, and since the
        # results leave with non-temporal stores (round 5) nothing displaces it from the 256 MB Infinity Cache between launches -- a
        # step over the same batch again then reads its points from that cache (+ 22 %: `repeated_batch` below), which is not what
        # the HBM roofline names.  Four batches (960 MB) cannot be held; `value` is the rate of points that come from HBM.
        batches = [d_xyz] + [torch.from_numpy(splitmix64_points(n, seed=12345 + rank + 7919 * b)).cuda() for b in range(1, max(1, args.batches))]
        d_out = torch.empty(n, dtype=torch.float64, device="cuda")
        tree = H.DeviceTree(ctx, block)
        torch.cuda.synchronize()
        turn = [0]

        def step():
            b = batches[turn[0] % len(batches)]
            turn[0] += 1
            tree.query_device(b.data_ptr(), n, d_out.data_ptr())

        for _ in range(args.warmup):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(args.steps):
            step()
        e1.record(stream)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0  # this rank's K steps; the MAX over ranks is taken below, behind the closing barrier
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        kernel_ms = e0.elapsed_time(e1) / args.steps  # average launch duration of query_kernel
        if world > 1:
            t = torch.tensor([wall], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())

        # the same kernel over ONE batch again and again (what rounds 1-4 timed; see above)
        def again():
            tree.query_device(d_xyz.data_ptr(), n, d_out.data_ptr())
        for _ in range(3):
            again()
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record(stream)
        for _ in range(20):
            again()
        r1.record(stream)
        torch.cuda.synchronize()
        repeated_ms = r0.elapsed_time(r1) / 20
        # sanity: the output is the real answer (spot parity against the oracle on rank 0; d_out now holds the first batch's values)
        got = d_out[:: max(1, n // 2000)].cpu().numpy()
        # ... and on EVERY rank, with or without the CPU-baseline leg, a cross-check that needs no oracle: the timed
        # kernel's values against the library's other implementation of Query -- calls of <= 32 points are answered on the calling thread
        # (csrc/host_query.cpp: the same statements compiled for the host) -- on 16 strided groups of 32 points, bit for bit
        stride_g = max(1, (n - 32) // 16)
        for g0 in range(0, n - 31, stride_g):
            pts32 = d_xyz[g0:g0 + 32].cpu().numpy()
            assert np.array_equal(tree.query(pts32).view(np.uint64), d_out[g0:g0 + 32].cpu().numpy().view(np.uint64)), \
                "the timed Query kernel and the host-answered Query disagree (rank %d, points %d..)" % (rank, g0)

        # ---------------- Create() with the frontier sharded over the ranks (one all-gather per round).  Behind the headline leg and
        # inside a try: this is the part of the run that needs RCCL to take an in-place all-gather on the library's stream, which no
        # box this was developed on could exercise with more than one GPU -- if it fails, the line still carries the weak-scaling Query
        # numbers and says what went wrong instead of taking the whole record down.
        if world > 1:
            try:
                block_s, st_s, create_sharded_ms, _, per_rank = timed_create("shard")
                assert block_s == block, "sharded and replicated Create disagree"
                create_sharded = {"ms_per_rank": per_rank, "exchanges_per_create": st_s.get("exchanges"), "rounds": st_s["rounds"],
                                  "backend": dist.get_backend()}
            except Exception as e:  # noqa: BLE001
                create_sharded_ms, create_sharded = None, {"error": "%s: %s" % (type(e).__name__, str(e)[:500])}
                args.mesh = "none"  # (the mesh leg shards the same way)

        # locality ceiling: the same points sorted by depth-4 cell, so neighbouring lanes share tree lines
        sorted_ms = None
        if rank == 0 and not args.no_sorted_ceiling:
            def by_cell(x):
                cell = ((x + 0.5) * 16.0).floor().clamp_(0, 15).to(torch.int64)
                o = torch.argsort(cell[:, 0] * 256 + cell[:, 1] * 16 + cell[:, 2])
                return x[o].contiguous(), o

            d_sorted, order = by_cell(d_xyz)
            sorted_batches = [d_sorted] + [by_cell(b)[0] for b in batches[1:]]  # (walked in turn, as the headline leg's)
            d_out2 = torch.empty_like(d_out)
            torch.cuda.synchronize()
            for b in sorted_batches:
                tree.query_device(b.data_ptr(), n, d_out2.data_ptr())
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record(stream)
            for i in range(8):
                tree.query_device(sorted_batches[(i + 1) % len(sorted_batches)].data_ptr(), n, d_out2.data_ptr())
            s1.record(stream)
            tree.query_device(d_sorted.data_ptr(), n, d_out2.data_ptr())
            torch.cuda.synchronize()
            sorted_ms = s0.elapsed_time(s1) / 8
            del sorted_batches
            assert torch.equal(d_out2, d_out[order]), "sorted-point Query differs from the unsorted one"
            del d_sorted, d_out2, order

        # Beyond the headline config: the same points against a tree the hp-refinement has actually worked on
        # (union3 @ 1e-7: ~12 k nodes, degrees 2-4, depths 4-6) -- Query goes through
        # query_general_kernel there -- and QueryWithGradient (reference benchmark HPBenchmarks.cpp:169-203).
        refined = None
        if rank == 0 and not args.no_refined:
            import ctypes as C
            blk_r, st_r = H.create_block(ctx, H.make_config(1e-7), field, JOBS_PER_ROUND)
            create_r_all = []
            torch.cuda.synchronize()
            for _ in range(11):  # the calls back to back, their median: as create_ms
                t0 = time.perf_counter()
                blk_r, st_r = H.create_block(ctx, H.make_config(1e-7), field, JOBS_PER_ROUND)
                create_r_all.append((time.perf_counter() - t0) * 1e3)
            torch.cuda.synchronize()
            create_r_ms = float(np.median(create_r_all))
            tree_r = H.DeviceTree(ctx, blk_r)
            d_grad = torch.empty(3 * n, dtype=torch.float64, device="cuda")
            d_out2 = torch.empty_like(d_out)
            L = H.lib()

            def timed(fn, reps=5):
                fn()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                for _ in range(reps):
                    fn()
                b.record(stream)
                torch.cuda.synchronize()
                return a.elapsed_time(b) / reps

            def nxt():  # (the batches in turn, as the headline leg)
                b = batches[turn[0] % len(batches)]
                turn[0] += 1
                return b.data_ptr()

            q_ms = timed(lambda: tree_r.query_device(nxt(), n, d_out2.data_ptr()), reps=8)
            tree_r.query_device(d_xyz.data_ptr(), n, d_out2.data_ptr())
            torch.cuda.synchronize()
            got_r = d_out2[:: max(1, n // 2000)].cpu().numpy()
            g_ms = timed(lambda: H.check(L.hpsdf_query_gradient_device(ctx.handle, tree_r.handle, C.c_void_p(nxt()), n,
                                                                      C.c_void_p(d_out2.data_ptr()), C.c_void_p(d_grad.data_ptr()))), reps=8)
            gc_ms = timed(lambda: H.check(L.hpsdf_query_gradient_device(ctx.handle, tree.handle, C.c_void_p(nxt()), n,
                                                                       C.c_void_p(d_out2.data_ptr()), C.c_void_p(d_grad.data_ptr()))), reps=8)
            refined = {"tree": "union3 @ 1e-7, K=%d: %d nodes, %d leaves, %d coeffs, max degree %d, max depth %d"
                               % (JOBS_PER_ROUND, st_r["n_nodes"], st_r["n_leaves"], st_r["n_coeffs"], tree_r.info()["max_degree"],
                                  tree_r.info()["max_depth"]),
                       "create_ms": create_r_ms, "create_ms_all": create_r_all, "create_jobs": st_r["jobs"], "create_rounds": st_r["rounds"],
                       "query_ms": q_ms, "query_mpts_per_s": n / q_ms / 1e3, "query_frac_hbm_peak": 32.0 * n / (q_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "query_with_gradient_ms": g_ms, "query_with_gradient_mpts_per_s": n / g_ms / 1e3,
                       "query_with_gradient_frac_hbm_peak": 56.0 * n / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "headline_tree_query_with_gradient_ms": gc_ms}
            refined["_blk"] = blk_r
            refined["_got"] = got_r
            del d_grad, d_out2

        mesh = None
        if args.mesh != "none":
            # BASELINE configs 2-4 in shape: a closed triangle mesh as the field, root = mesh box, targetError 1e-5 and
            # 1e-6; with N > 1 ranks the frontier is sharded (one all-gather per round, hpsdf_create_distributed).
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from helpers import icosphere, displaced_torus
            if args.mesh == "torus":
                verts, tris = displaced_torus()
                mname = "displaced torus grid 1024 x 1024"
            else:
                verts, tris = icosphere(int(args.mesh), 0.4)
                dirs = verts / np.linalg.norm(verts, axis=1, keepdims=True)
                verts = (verts * (1.0 + 0.08 * np.sin(9 * dirs[:, 0]) * np.cos(7 * dirs[:, 1])
                                  + 0.05 * np.sin(11 * dirs[:, 2]))[:, None]).astype(np.float32)
                mname = "bumpy icosphere, level %s" % args.mesh
            lo, hi = verts.min(0) - 0.02, verts.max(0) + 0.02
            # hpsdf_field_create_mesh end to end (upload of vertices + indices, twin half-edges and LBVH on the device): the
            # first call also pays for the first allocations of this size and rocPRIM's first launch, so it is listed apart
            prep_all = []
            mfield = None
            for _ in range(4):
                if mfield is not None:
                    mfield.close()
                t0 = time.perf_counter()
                mfield = H.Field.mesh(ctx, verts, tris)
                prep_all.append((time.perf_counter() - t0) * 1e3)
            mesh = {"mesh": mname, "triangles": int(len(tris)), "prepare_ms": sorted(prep_all[1:])[1], "prepare_first_call_ms": prep_all[0],
                    "prepare_ms_all": prep_all, "n_gpus": world}
            for tgt, key in ((1e-5, "1e-5"), (1e-6, "1e-6")):
                mcfg = H.make_config(tgt, tuple(lo), tuple(hi))

                def mcreate():
                    if world > 1:
                        return D.create_distributed(ctx, mcfg, mfield, JOBS_PER_ROUND, policy="shard")
                    return H.create_block(ctx, mcfg, mfield, JOBS_PER_ROUND)
                try:
                    mcreate()
                except Exception as e:  # noqa: BLE001  (N > 1 only in practice: see the sharded Create above)
                    mesh["error_" + key] = "%s: %s" % (type(e).__name__, str(e)[:500])
                    break
                mt = []
                for _ in range(3):
                    if world > 1:
                        dist.barrier()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    mblk, mst = mcreate()
                    torch.cuda.synchronize()
                    mt.append((time.perf_counter() - t0) * 1e3)
                ms = float(np.median(mt))
                if world > 1:
                    t = torch.zeros(world, device="cuda", dtype=torch.float64)
                    t[rank] = ms
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                    mesh["create_ms_per_rank_" + key] = [float(x) for x in t.tolist()]
                    mesh["exchanges_per_create_" + key] = mst.get("exchanges")
                    ms = float(t.max().item())
                mesh["create_ms_" + key] = ms
                mesh["tree_" + key] = {"nodes": mst["n_nodes"], "rounds": mst["rounds"], "samples": mst["samples"],
                                       "msamples_per_s": mst["samples"] / ms / 1e3}
            mesh["create_ms"] = mesh.get("create_ms_1e-5")  # the north_star's "2 M-tri mesh at targetError 1e-5"
            # What bounds the sampling leg: mesh_sample_kernel is VALU-issue bound, not HBM bound (8 B written per sample against ~1.4 KB
            # gathered, mostly from L2 / Infinity Cache).  The live part is samples/s; the counter figures (issue fraction = SQ_INSTS_VALU
            # x 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), lane utilisation, bytes per sample) cannot be read from inside this
            # process: they are those of profiles/mesh_pmc.json (tools/mesh_pmc.sh), stamped with a hash of the kernels' source and
            # reported as null when the source has changed since they were taken.
            mesh["roofline"] = {"kernel": "mesh_sample_kernel", "bound": "valu-issue", "unit": "G samples/s",
                                "achieved": mesh["tree_1e-6"]["msamples_per_s"] / 1e3 if "tree_1e-6" in mesh else None,
                                "algorithmic_bytes_per_sample": 8}
            mesh["roofline"].update(counter_record("mesh_pmc.json", ("frac_valu_issue", "lane_utilisation", "valu_insts_per_64_samples",
                                                                     "fetched_bytes_per_sample", "written_bytes_per_sample")))
            del mfield

        fit = None
        if not args.no_fit_bench and rank == 0:
            # Steady-state fit micro-benchmark: from-scratch fits of a lattice of depth-5 cells, per degree; the
            # default bit-exact kernel and the opt-in matrix-core kernel (hpsdf_ctx_set_fast_fit, degrees 4..9), each with
            # the headline field and with a field that costs nothing (contraction only).  Algorithmic flops:
            # 2 ncoef(p) (4p+1)^3 per fit, against the FP64 peak (78.6 TFLOP/s, vector = matrix on this chip: measured
            # 77.6 with back-to-back v_mfma_f64_16x16x4_f64, tools/mfma_f64_rate.hip).
            fit = {"default_mode": "HPSDF_FIT_SPLIT: from-scratch fits of degree >= 6 keep their rows of top degree on the bit-exact kernel (errors, "
                                   "decisions, topology canonical) and compute the rows below them by sum factorisation from the same samples (fit_low_kernel; "
                                   "HPSDF_LOW_KERNEL=mfma: the direct contraction on the matrix cores); degrees 2-5 are the bit-exact kernel throughout"}
            fast_ctx = H.Context(local, stream.cuda_stream)
            fast_ctx.set_fast_fit(True)
            exact_ctx = H.Context(local, stream.cuda_stream)
            exact_ctx.set_fit_mode(H.FIT_EXACT)
            plane = H.Field.analytic([(H.PRIM_PLANE, H.OP_UNION, [0.3, -0.2, 0.5, 0.1])])  # F costs ~nothing: contraction only
            for p in (2, 3, 4, 5, 6, 7, 8):  # p in {2..8}; degrees > 5 run the any-degree kernel by default
                cells = 65536 if p <= 3 else 16384
                flops = 2.0 * H.NCOEF[p] * (4 * p + 1) ** 3 * cells
                ms = H.bench_fit(ctx, cfg, field, p, 5, cells, 3)
                ms_c = H.bench_fit(ctx, cfg, plane, p, 5, cells, 3)
                fit["p%d" % p] = {"cells": cells, "ms": ms, "tflops_algorithmic": flops / ms / 1e9,
                                  "frac_fp64_peak": flops / ms / 1e9 / FP64_PEAK_TFLOPS,
                                  "contraction_only_ms": ms_c, "contraction_only_tflops": flops / ms_c / 1e9,
                                  "contraction_only_frac_fp64_peak": flops / ms_c / 1e9 / FP64_PEAK_TFLOPS}
                fit["p%d" % p]["kernel"] = ("fit_kernel (top-degree rows, bit-exact) + fit_low_kernel (the rows below them, sum-factorised, FMA)" if p >= 6
                                            else "fit_kernel (bit-exact)")
                if p >= 6:  # what the default replaced: every row on the bit-exact kernel
                    ems = H.bench_fit(exact_ctx, cfg, field, p, 5, cells, 3)
                    fit["p%d" % p]["exact_fit"] = {"kernel": "fit_kernel (bit-exact, every row)", "ms": ems, "tflops_algorithmic": flops / ems / 1e9,
                                                   "frac_fp64_peak": flops / ems / 1e9 / FP64_PEAK_TFLOPS}
                if p >= 2:  # (degrees 2-3 reach the matrix cores in this micro-benchmark only: builds send degrees >= 4 there)
                    fms = H.bench_fit(fast_ctx, cfg, field, p, 5, cells, 3)
                    fms_c = H.bench_fit(fast_ctx, cfg, plane, p, 5, cells, 3)
                    fit["p%d" % p]["fast_fit"] = {"kernel": "fit_mfma_kernel (v_mfma_f64_16x16x4_f64)", "ms": fms,
                                                  "tflops_algorithmic": flops / fms / 1e9,
                                                  "frac_fp64_peak": flops / fms / 1e9 / FP64_PEAK_TFLOPS,
                                                  "contraction_only_ms": fms_c, "contraction_only_tflops": flops / fms_c / 1e9,
                                                  "contraction_only_frac_fp64_peak": flops / fms_c / 1e9 / FP64_PEAK_TFLOPS}
            fast_ctx.close()
            exact_ctx.close()
            # what bounds the exact fit is VALU issue (the field's instructions and the un-fusable multiplies), not FP64 throughput
            # counted in FMAs: the issue fraction per degree, from profiles/fit_pmc.json (tools/fit_pmc_all.sh; null when stale)
            fit["roofline"] = {"bound": "valu-issue", "peak_fp64_tflops": FP64_PEAK_TFLOPS}
            fit["roofline"].update(counter_record("fit_pmc.json", ("degrees",)))
            # the matrix-core fit's utilisation from counters (north_star: "MFMA utilisation for the fit against gfx950 peak"):
            # SQ_VALU_MFMA_BUSY_CYCLES against the kernel's active cycles, profiles/fit_mfma_pmc.json (tools/fit_mfma_pmc.sh; null when stale)
            mf = counter_record("fit_mfma_pmc.json", ("degrees",))
            for p in (4, 6, 8):
                d = (mf.get("degrees") or {}).get("p%d" % p) or {}
                if "fast_fit" in fit.get("p%d" % p, {}):
                    fit["p%d" % p]["fast_fit"]["mfma_busy"] = (d.get("union3") or {}).get("mfma_busy")
                    fit["p%d" % p]["fast_fit"]["contraction_only_mfma_busy"] = (d.get("plane") or {}).get("mfma_busy")
            fit["fast_fit_counters"] = mf["measured_from"]

    ms_per_step = wall * 1e3 / args.steps
    value = world * n * args.steps / wall / 1e6  # Mpts/s, whole job
    achieved = 32.0 * n / (kernel_ms * 1e-3) / 1e9  # GB/s, algorithmic bytes / launch duration

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return


    # roofline.traffic: HBM bytes per query_kernel launch from the PMC passes of tools/profile.sh (separate rocprofv3 --pmc
    # runs of this very command; FETCH_SIZE / WRITE_SIZE corrected as MI355X_MICROARCH.md prescribes).  Counters cannot be
    # read from inside this process, so the figure is the one committed with the profile -- stamped with what it was taken
    # from, and dropped (null) when the kernel source has changed since.
    traffic, traffic_from = None, None
    pmc = os.path.join(ROOT, "profiles", "query_pmc.json")
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc))
            import hashlib
            src = os.path.join(ROOT, "hp-adaptive-signed-distance-field-octree_amd", "csrc", "kernels.hip")
            text = open(src).read()
            a_ = text.index("template <int TOPD, bool DEDUPE, bool GRAD>")
            b_ = text.index("// 16-byte chunks a leaf of degree d occupies")
            qsha = hashlib.sha256(text[a_:b_].encode()).hexdigest()[:16]
            traffic_from = {"profile": rec.get("profile"), "query_kernel_sha16": rec.get("query_kernel_sha16"),
                            "current_query_kernel_sha16": qsha}
            if rec.get("query_kernel_sha16") in (None, qsha):
                traffic = rec.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "Query() Mpts/sec at targetError=1e-5 (union3 field, 10M SplitMix64 pts per GPU, HBM-resident); "
                  "Create() ms reported as create_ms",
        "value": value, "unit": "Mpts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        # what the collective layer saw (N > 1): ranks, backend (nccl = RCCL over xGMI), all-gathers per sharded Create
        "world": world, "backend": backend,
        "exchanges_per_create": None if create_sharded is None else create_sharded.get("exchanges_per_create"),
        "config": {"workload": "BASELINE configs[1]: union(sphere,box,torus) analytic SDF, targetError=1e-5, "
                               "continuity off, %d random Query() points per GPU a step; the steps walk %d distinct batches in turn, so "
                               "the points come from HBM and not from the 256 MB Infinity Cache" % (n, len(batches)),
                   "jobs_per_round": JOBS_PER_ROUND, "points_per_gpu": n, "sharding": "replicated tree, points split",
                   "point_batches": len(batches)},
        # one batch queried again and again: its 240 MB stay in the 256 MB Infinity Cache between launches (results bypass it), so this
        # is NOT an HBM figure -- reported because rounds 1-4 timed it this way and a caller re-querying one batch sees it
        "repeated_batch": {"avg_launch_ms": repeated_ms, "mpts_per_s": n / repeated_ms / 1e3,
                           "algorithmic_gbps": 32.0 * n / (repeated_ms * 1e-3) / 1e9,
                           "note": "one 10 M batch queried again and again, as rounds 1-4 timed `value` (r04: 90.4 Gpts/s with plain result stores): "
                                   "the points are served from the Infinity Cache, so this is not priced against HBM; `value` walks "
                                   "config.point_batches distinct batches"},
        "create_ms": create_ms, "create_sharded_ms": create_sharded_ms, "create_sharded": create_sharded,
        "create": {"nodes": stats["n_nodes"], "leaves": stats["n_leaves"], "coeffs": stats["n_coeffs"],
                   "rounds": stats["rounds"], "jobs": stats["jobs"], "fits": stats["fits"], "samples": stats["samples"],
                   "block_bytes": len(block), "ms_all": create_times},
        "roofline": {"kernel": "query_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_measured_from": traffic_from,
                     "algorithmic_bytes_per_launch": 32 * n,
                     "avg_launch_ms": kernel_ms,
                     # what actually limits the kernel: every random point pulls one 128-byte
                     # top-table line out of L2 besides its 24 + 8 streamed bytes; MI355X_MICROARCH.md measures 66-73
                     # GB/s per CU for rows gathered from an XCD's L2
                     "l2_to_l1_fill": {"bytes_per_point": 160, "achieved_gbps_per_cu": 160.0 * n / (kernel_ms * 1e-3) / 1e9 / 256,
                                       "measured_ceiling_gbps_per_cu": [66, 73]}},
    }
    refined_check = None
    if refined is not None:
        refined_check = (refined.pop("_blk"), refined.pop("_got"))
        out["refined_tree"] = refined
    if sorted_ms is not None:
        out["query_cell_sorted_points"] = {"avg_launch_ms": sorted_ms, "mpts_per_s": n / sorted_ms / 1e3,
                                           "frac_hbm_peak": 32.0 * n / (sorted_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if mesh:
        out["mesh_create"] = mesh
    if fit:
        out["fit_microbench"] = fit
    if world == 1 and not args.no_cpu_baseline:
        # The cpu_baseline leg -- the one place this file touches oracle/: the CPU restatement timed on the host's cores, and,
        # since it is here, the timed GPU outputs checked against it (a sample of the 10 M values of both trees, bit for bit).
        import oracle as O
        otree = O.Tree.from_block(block)
        assert np.array_equal(got, otree.query(pts[:: max(1, n // 2000)])), "timed Query output differs from the oracle"
        if refined_check is not None:
            want_r = O.Tree.from_block(refined_check[0]).query(pts[:: max(1, n // 2000)])
            assert np.array_equal(refined_check[1], want_r), "timed Query output on the refined tree differs from the oracle"
        # bounded sample (~10 s of CPU work): whole passes of the oracle's Query over the same points until 8 s have
        # gone by; the oracle's Create of the same config three times
        m, passes, tq = n, 0, 0.0
        while tq < 8.0:
            t0 = time.perf_counter()
            otree.query(pts[:m])
            tq += time.perf_counter() - t0
            passes += 1
        tcs = []
        for _ in range(3):
            t0 = time.perf_counter()
            O.Tree.create(O.default_config(TARGET), O.union3_field(), JOBS_PER_ROUND)
            tcs.append(time.perf_counter() - t0)
        # the same Query on every host core (Octree::Query is const: the reference's own parallel use), ~4 s: one pthread per
        # core inside the oracle's C loop (ora_query_batch_mt) -- no interpreter, no GIL between the cores and the points
        ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        visible = ncores
        try:  # a container's CPU share (cgroup v2 cpu.max / v1 cfs quota) can be far below the cores it can see
            quota = None
            if os.path.exists("/sys/fs/cgroup/cpu.max"):
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                quota = None if q == "max" else float(q) / float(per)
            elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
                q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                quota = None if q <= 0 else q / per
            if quota:
                ncores = max(1, min(ncores, int(quota + 0.999)))
        except Exception:
            pass
        t0 = time.perf_counter()
        otree.query(pts[:m], threads=ncores, passes=4)  # warm, and a first estimate of a pass
        est = (time.perf_counter() - t0) / 4
        apasses = int(max(4, min(4000, 4.0 / max(est, 1e-4))))  # ~4 s in ONE call: the threads start once, each repeats its part
        t0 = time.perf_counter()
        otree.query(pts[:m], threads=ncores, passes=apasses)
        tall = time.perf_counter() - t0
        tca = []
        for _ in range(3):  # ... and Create with a round's jobs on every core (same tree: the jobs of a round are pure)
            t0 = time.perf_counter()
            O.Tree.create(O.default_config(TARGET), O.union3_field(), JOBS_PER_ROUND, threads=ncores)
            tca.append(time.perf_counter() - t0)
        out["cpu_baseline_all_cores"] = {"value": apasses * m / tall / 1e6, "unit": "Mpts/s", "cores": ncores, "kind": "port",
                                         "sample": "oracle Query() over the same %d points cut into %d contiguous parts, one pthread each (started once), "
                                                   "%d whole passes (%.1f s); oracle Create() with a round's jobs on %d pthreads, median of 3"
                                                   % (m, ncores, apasses, tall, ncores),
                                         "create_ms": float(np.median(tca)) * 1e3, "visible_cpus": visible}
        out["cpu_baseline"] = {"value": passes * m / tq / 1e6, "unit": "Mpts/s", "cores": 1, "kind": "port",
                               "sample": "oracle Query() over the same %d points, %d whole passes (%.1f s); "
                                         "oracle Create() of the same config, median of 3" % (m, passes, tq),
                               "create_ms": float(np.median(tcs)) * 1e3, "host_cpus": os.cpu_count()}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
