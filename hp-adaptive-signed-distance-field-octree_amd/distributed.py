"""Multi-GPU Create/Query: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Create
    The tree, the frontier and the field are replicated; every rank selects the same jobs and
    computes only its cost-balanced slice of each round on its GPU.  Per round the ranks exchange
    the 9 errors of every job (72 B/job) with ONE all-gather of equal (padded) parts -- the
    coefficients stay in the arena of the rank that fitted them.  After the last round one
    all-gather reassembles the packed coefficients on every rank.  Every rank applies identical
    results in identical order, so the MemoryBlock is byte-identical for any world size
    (tests/test_distributed_gloo.py, tests/test_gpu_configs.py).

    On GPUs the whole loop sits behind the C ABI (hpsdf_create_distributed) and this module only supplies
    the all-gather.  Fields the GPU evaluates itself (analytic, mesh, tree-CSG), with or without
    nearness weighting, take the device-side frontier (csrc/frontier.hip) on up to 8 ranks: selection,
    slicing, decision and bookkeeping run on every rank's GPU and the exchange points are all-gathers of
    device buffers.  A weighted build hands the rows each round fitted to every rank (one more
    all-gather per round, of the round's part of the arena: a weighted incremental fit copies the
    node's previous rows and may run on any rank) and needs no exchange of packed coefficients at the
    end.  Host callbacks and more than 8 ranks run the host scheduler's rounds (the per-round errors
    then pass through host memory, staged through a device buffer for the all-gather).  The Python
    round loop below is the same algorithm over the stepwise C API: the CPU tests drive it (compute=
    hook), and HPSDF_PYTHON_ROUND_LOOP=1 forces it on GPUs.

Continuity (config.continuity.enforce)
    The host-side post-process runs on every rank on its identical copy of the assembled block; its
    reductions are thread-count independent, so the ranks stay byte-identical without a collective.

Query
    The tree is replicated (<= a few MB); points are split contiguously; no collective.

The exchange runs on whatever device the process group lives on: HBM tensors for nccl, host tensors
for gloo (CPU tests inject oracle-computed job results through ``compute=``; the product path
always computes on the GPU).
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import Build, JOB_HEADER_DOUBLES, continuity_post_process


class _DevPtr:
    """Wraps a raw HBM pointer for torch.as_tensor (no copy)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def _gather_padded(local, pad_to, group, world):
    """all-gather of equal-size (padded) 1-D float64 tensors -> [world, pad_to] tensor."""
    buf = torch.zeros(pad_to, dtype=torch.float64, device=local.device)
    buf[: local.numel()] = local
    out = torch.empty(world * pad_to, dtype=torch.float64, device=local.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    return out.view(world, pad_to)


class _DevBytes:
    """Raw HBM bytes for torch.as_tensor (no copy)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_allgather(ctx, group=None):
    """The in-place all-gather hpsdf_create_distributed asks for, over torch.distributed.

    backend nccl (RCCL over xGMI): ONE all_gather_into_tensor on the device buffer itself, issued with the context's
    stream as torch's current stream -- the collective is ordered behind the kernels that wrote this rank's part and the
    kernels that read the result are ordered behind the collective, by stream order alone: no host synchronisation, no
    staging copy (rank r's part already sits at its place in the receive buffer; RCCL's all-gather is in-place when the
    send buffer is recvbuf + rank * count).  backend gloo (CPU tests, one-GPU rehearsals): through host memory.

    ``gather.calls`` counts the exchanges (bench.py reports it per Create)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    on_device = dist.get_backend(group) == "nccl"
    device = torch.device("cuda", ctx.device)
    streams = {}
    state = {"in_place": True}

    def gather(d_buf, nbytes, stream):
        gather.calls += 1
        buf = torch.as_tensor(_DevBytes(d_buf, nbytes * world), device=device)
        mine = buf[rank * nbytes:(rank + 1) * nbytes]
        if on_device:
            key = int(stream or 0)
            ext = streams.get(key)
            if ext is None:
                ext = streams[key] = torch.cuda.ExternalStream(key, device=device) if key else torch.cuda.default_stream(device)
            with torch.cuda.stream(ext):
                if state["in_place"]:
                    try:
                        dist.all_gather_into_tensor(buf, mine, group=group)
                        return
                    except (RuntimeError, ValueError) as e:
                        # ONLY the argument check of a torch build that rejects an input aliasing the output: it is raised before
                        # anything is enqueued and on every rank alike (same torch, same call), so every rank takes the copy below
                        # from now on.  Anything else -- an RCCL error, out of memory -- is this rank's alone: issuing a second
                        # collective here would leave the ranks' sequences out of step, so it is passed on.
                        msg = str(e).lower()
                        if not any(k in msg for k in ("overlap", "alias", "in-place", "inplace", "same tensor", "same memory")):
                            raise
                        state["in_place"] = False
                dist.all_gather_into_tensor(buf, mine.clone(), group=group)
        else:
            ctx.synchronize()  # this rank's part was written on the context's stream
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, mine.cpu(), group=group)
            buf.copy_(torch.cat(parts))
            torch.cuda.current_stream().synchronize()
    gather.calls = 0
    return gather


def create_distributed(ctx, config, field, K=0, group=None, compute=None, policy="shard"):
    """Octree::Create over the ranks of ``group``.  Returns (block bytes, stats) on every rank.

    policy "shard": every round's jobs are split over the ranks and their errors all-gathered (what the
    north_star asks for; pays when the field is expensive -- meshes, host callbacks).  "replicate": every rank
    builds the whole tree by itself (deterministic, so the blocks are identical; no exchange) -- for analytic
    fields a whole Create is a fraction of a millisecond and any exchange costs more than it saves.  "auto":
    replicate for GPU-evaluated analytic fields, shard otherwise.

    compute(build, jobs, first, count) -> (headers [count,9] ndarray) may replace the GPU leg
    (test hook: it must also ``build.inject`` the coefficients of its jobs).
    """
    if policy not in ("shard", "replicate", "auto"):
        raise ValueError("policy must be shard, replicate or auto")
    if compute is None and (policy == "replicate" or (policy == "auto" and getattr(field, "kind", None) == "analytic")):
        from . import create_block
        return create_block(ctx, config, field, K)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    on_gpu = compute is None
    pod = config.to_pod() if hasattr(config, "to_pod") else config
    if on_gpu and world > 1 and os.environ.get("HPSDF_PYTHON_ROUND_LOOP") != "1":
        # the whole sharded build behind the C ABI (hpsdf_create_distributed): the device-side frontier for fields the GPU
        # evaluates itself (weighted or not, up to 8 ranks), the host scheduler's rounds for the rest -- both over this all-gather
        from . import create_block_distributed
        gather = device_allgather(ctx, group)
        block, stats = create_block_distributed(ctx, config, field, K, rank, world, gather)
        stats["exchanges"] = gather.calls
        return block, stats
    # Nearness-weighted builds: the weight (pow / exp) is applied on the host (one libm for GPU path and oracle), so the
    # per-round errors pass through host memory, and after every round the ranks hand each other the coefficient arrays
    # that round accepted -- an incremental fit copies the node's previous rows and may run on any rank.
    weighted = pod.weighting_type != 0
    backend_dev = torch.device("cuda", ctx.device) if (ctx is not None and dist.is_initialized() and dist.get_backend(group) == "nccl") \
        else torch.device("cpu")
    dev = torch.device("cuda", ctx.device) if (on_gpu and not weighted) else (backend_dev if world > 1 else torch.device("cpu"))
    b = Build(config, K, rank, world)
    while True:
        n = b.select()
        if n == 0:
            break
        first, count = b.slice()
        if on_gpu and not weighted:
            b.compute(ctx, field)
            ptr, nd = b.results_device()
            ctx.synchronize()  # the fit kernel ran on the context stream; the collective runs on torch's
            local = torch.as_tensor(_DevPtr(ptr, nd), device=dev) if nd else torch.empty(0, dtype=torch.float64, device=dev)
        elif on_gpu:
            b.compute(ctx, field)
            local = torch.from_numpy(b.results_host(ctx)).to(dev)
        else:
            local = torch.from_numpy(np.ascontiguousarray(compute(b, b.jobs(n), first, count), np.float64).reshape(-1)).to(dev)
        if world == 1:
            headers = local.cpu().numpy().reshape(n, JOB_HEADER_DOUBLES)
        else:
            pad = b.max_slice() * JOB_HEADER_DOUBLES
            gathered = _gather_padded(local, pad, group, world).cpu().numpy()
            headers = np.empty((n, JOB_HEADER_DOUBLES))
            for r in range(world):
                f, c = b.slice(r)
                headers[f:f + c] = gathered[r, : c * JOB_HEADER_DOUBLES].reshape(c, JOB_HEADER_DOUBLES)
        b.apply(headers)
        if weighted and world > 1:
            rc = b.rows_counts()
            mine = torch.from_numpy(b.rows_pack_host(ctx if on_gpu else None, rc[rank])).to(dev)
            g = _gather_padded(mine, max(1, max(rc)), group, world).cpu().numpy()
            b.rows_unpack_host(ctx if on_gpu else None, [g[r, : rc[r]] for r in range(world)])
    _, counts = b.layout()
    if on_gpu and dev.type == "cuda":
        ptr, nd = b.pack_device(ctx)
        ctx.synchronize()
        mine = torch.as_tensor(_DevPtr(ptr, nd), device=dev) if nd else torch.empty(0, dtype=torch.float64, device=dev)
    else:
        mine = torch.from_numpy(b.pack_host(ctx if on_gpu else None, counts[rank]))
    if world == 1:
        packs = [mine.cpu().numpy()]
    else:
        g = _gather_padded(mine, max(1, max(counts)), group, world).cpu().numpy()
        packs = [g[r, : counts[r]] for r in range(world)]
    block = b.assemble(packs)
    stats = b.stats()
    b.close()
    if b.pod.continuity_enforce:
        # Octree.cpp:341-344: host-side post-process; deterministic, so every rank computes the same block
        # from its identical copy and no exchange is needed
        block, cstats = continuity_post_process(block, ctx=ctx if on_gpu else None)
        stats["continuity"] = cstats
    return block, stats


def shard_points(n, rank, world):
    """Contiguous split of n query points: rank r gets [lo, hi)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
