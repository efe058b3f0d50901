"""Parity tests proper (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the
same inputs.  Bar: the north_star asks for 1e-6 abs on coefficients and Query values and bit-exact
topology; the kernels run the reference's operation order with no fused multiply-add, so the tests
demand BIT-IDENTICAL MemoryBlocks and Query values, which implies both."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from conftest import bits
from helpers import (oracle_field, product_field, query_points, edge_points, synthetic_block, deep_chain_block, icosphere, displaced_torus, sha)

pytestmark = pytest.mark.gpu
DBL_MAX = np.finfo(np.float64).max
TOL = 1e-6  # north_star tolerance (abs) on coefficients and Query values


# ------------------------------------------------------------------ fields (the sampling leg)
@pytest.mark.parametrize("name", ["sphere", "union3", "sphere075"])
def test_field_values_bitwise(H, O, ctx, name):
    pts = np.concatenate([O.splitmix64_points(400000, seed=3), edge_points(np.random.default_rng(8), 4000)[:3000]])
    pts = pts[np.all(np.isfinite(pts), axis=1)]
    got = product_field(H, name).eval(ctx, pts)
    want = oracle_field(O, name).eval(pts)
    assert np.array_equal(bits(got), bits(want))


def test_field_primitives_and_ops_bitwise(H, O, ctx):
    spec = [(O.PRIM_BOX, O.OP_UNION, [0.1, 0.0, -0.1, 0.2, 0.15, 0.1]),
            (O.PRIM_SPHERE, O.OP_SUBTRACT, [0.2, 0.1, 0.0, 0.12]),
            (O.PRIM_TORUS_Y, O.OP_INTERSECT, [0.0, 0.0, 0.0, 0.3, 0.2]),
            (O.PRIM_PLANE, O.OP_UNION, [0.6, 0.0, 0.8, 0.35])]
    pts = O.splitmix64_points(200000, seed=21) * 1.5
    got = H.Field.analytic(spec).eval(ctx, pts)
    want = O.AnalyticField(spec).eval(pts)
    assert np.array_equal(bits(got), bits(want))
    # the square root's special inputs (field_eval.hpp sqrtExact: 0, subnormal and tiny squared distances, overflow to infinity), alone in
    # a wave and mixed with ordinary points
    c = np.array([0.2, 0.1, 0.0])
    offs = [0.0, 5e-324, 1e-310, 1e-200, 3e-162, 1.5e-154, 1e-120, 1e-115, 1e-100, 1e-30, 1e150, 1e154, 1.4e154, 1e200, 1e308]
    special = np.array([c + np.array([o, 0.0, 0.0]) for o in offs] + [c + np.array([o, -o, o]) for o in offs] + [c - np.array([0.0, o, 0.0]) for o in offs])
    sphere = [(O.PRIM_SPHERE, O.OP_UNION, [0.2, 0.1, 0.0, 0.12])]
    torus = [(O.PRIM_TORUS_Y, O.OP_UNION, [0.2, 0.1, 0.0, 0.0, 0.05])]
    for sp in (sphere, torus, spec):
        for batch in (special, np.concatenate([special, pts[:83]]), np.repeat(special, 64, axis=0)):
            assert np.array_equal(bits(H.Field.analytic(sp).eval(ctx, batch)), bits(O.AnalyticField(sp).eval(batch))), sp[0][0]


# ------------------------------------------------------------------ Create
@pytest.mark.parametrize("case", ["C1_sphere_1e-4", "C2_union3_1e-5", "A1_union3_1e-7_K1024", "A1_union3_1e-7_K256",
                                  "A2_sphere_1e-8_K1024", "D1_sphere075_customroot_1e-6"])
def test_create_block_identical_to_oracle(H, O, ctx, golden, case):
    g = golden["blocks"][case]
    cfg = H.make_config(g["target"], g["root_min"], g["root_max"])
    blk, st = H.create_block(ctx, cfg, product_field(H, g["field"]), g["K"])
    # committed fixture (oracle-generated) ...
    assert len(blk) == g["block_bytes"]
    assert hashlib.sha256(blk).hexdigest() == g["block_sha256"]
    assert st["n_nodes"] == g["n_nodes"] and st["n_coeffs"] == g["n_coeffs"] and st["jobs"] == g["stats"]["jobs"]
    # ... and the oracle run live on the same inputs, with the tolerance view for the record
    ob = O.Tree.create(O.default_config(g["target"], g["root_min"], g["root_max"]), oracle_field(O, g["field"]), g["K"]).to_block()
    a, b = O.parse_block(blk), O.parse_block(ob)
    assert np.array_equal(a["degree"], b["degree"]) and np.array_equal(a["childIdx"], b["childIdx"])  # topology
    assert np.abs(a["coeffs"] - b["coeffs"]).max() <= TOL
    assert blk == ob


def test_create_with_host_callback_equals_device_field(H, O, ctx, golden):
    """Create(config, std::function): host threads sample, the GPU fits -- same block as the analytic field."""
    import math

    def sphere(pt, thread_idx):
        dx, dy, dz = pt[0] - 0.25, pt[1], pt[2]
        return math.sqrt(dx * dx + (dy * dy + dz * dz)) - 0.5

    cfg = H.make_config(1e-4, threads=3)
    blk, _ = H.create_block(ctx, cfg, H.Field.callback(sphere), 1024)
    b = bytearray(blk)
    b[-80 + 48:-80 + 56] = np.array([1], np.uint64).tobytes()  # golden block was built with threadCount = 1
    assert hashlib.sha256(bytes(b)).hexdigest() == golden["blocks"]["C1_sphere_1e-4"]["block_sha256"]


def test_create_is_deterministic_and_stepwise_equals_oneshot(H, ctx):
    cfg = H.make_config(1e-7)
    f = H.Field.union3()
    one, _ = H.create_block(ctx, cfg, f, 512)
    two, _ = H.create_block(ctx, cfg, f, 512)
    assert one == two
    b = H.Build(cfg, 512, 0, 1)
    while True:
        n = b.select()
        if n == 0:
            break
        b.compute(ctx, f)
        b.apply(b.results_host(ctx).reshape(n, 9))
    tot, counts = b.layout()
    assert b.assemble([b.pack_host(ctx, counts[0])]) == one


def test_job_results_match_the_oracle_job_by_job(H, O, ctx):
    """A job's nine errors -- the incremental fit's (EstimatePImprovement, Octree.cpp:829-856) and the eight child fits'
    (EstimateHImprovement, :804-826) -- as the GPU leaves them for the decision, against ora_job on the same cell, degree and
    error: every round of union3 at 1e-7 (K = 256), 48 jobs spread over the round, bit for bit.  Unlike a block comparison this
    also sees the errors of the alternative a job did NOT take."""
    cfg, ocfg, f, of = H.make_config(1e-7), O.default_config(1e-7), H.Field.union3(), oracle_field(O, "union3")
    b = H.Build(cfg, 256, 0, 1)
    rounds = checked = 0
    degrees = set()
    while True:
        n = b.select()
        if n == 0:
            break
        jobs = b.jobs(n)
        b.compute(ctx, f)
        hdr = b.results_host(ctx).reshape(n, 9)
        for j in np.unique(np.linspace(0, n - 1, 48 if rounds else 24).astype(int)):
            jb = jobs[j]
            prev = None if jb.coarse else np.zeros(O.NCOEF[jb.degree])  # (an unweighted incremental fit forms the new rows only)
            res, _, _ = O.job(of, ocfg, tuple(jb.aabb_min), tuple(jb.aabb_max), jb.depth, jb.degree, jb.err, prev)
            want = np.array([res.p_err] + list(res.h_err))
            live = np.ones(9, bool)
            if jb.coarse:
                live[1:] = False  # a coarse job has no child fits (:836-843)
            assert np.array_equal(bits(hdr[j][live]), bits(want[live])), (rounds, j, jb.degree, jb.depth)
            checked += 1
            degrees.add(int(jb.degree))
        b.apply(hdr)
        rounds += 1
    assert rounds >= 10 and checked > 400 and max(degrees) >= 3


def test_distributed_driver_world1(H, ctx, golden):
    import importlib
    D = importlib.import_module("hpsdf_amd.distributed")
    g = golden["blocks"]["A2_sphere_1e-8_K1024"]
    blk, _ = D.create_distributed(ctx, H.make_config(g["target"]), H.Field.sphere(), g["K"])
    assert hashlib.sha256(blk).hexdigest() == g["block_sha256"]


def test_simulated_ranks_on_one_gpu_give_identical_block(H, ctx, golden):
    """The sharded path with every rank's slice computed on this GPU: byte-identical for 1/2/4 ranks."""
    g = golden["blocks"]["A1_union3_1e-7_K1024"]
    cfg = H.make_config(g["target"])
    f = H.Field.union3()
    for world in (2, 4):
        builds = [H.Build(cfg, g["K"], r, world) for r in range(world)]
        while True:
            n = builds[0].select()
            for b in builds[1:]:
                assert b.select() == n
            if n == 0:
                break
            hdr = np.zeros((n, 9))
            for b in builds:
                b.compute(ctx, f)
                first, count = b.slice()
                hdr[first:first + count] = b.results_host(ctx).reshape(count, 9)
            for b in builds:
                b.apply(hdr)
        lays = [b.layout() for b in builds]
        packs = [builds[r].pack_host(ctx, lays[r][1][r]) for r in range(world)]
        assert hashlib.sha256(builds[0].assemble(packs)).hexdigest() == g["block_sha256"]


# ------------------------------------------------------------------ Query
@pytest.mark.parametrize("case", ["C1_sphere_1e-4", "A1_union3_1e-7_K1024", "D1_sphere075_customroot_1e-6"])
def test_query_bitwise_and_golden(H, O, ctx, golden, case):
    g = golden["blocks"][case]
    ot = O.Tree.create(O.default_config(g["target"], g["root_min"], g["root_max"]), oracle_field(O, g["field"]), g["K"])
    tree = H.DeviceTree(ctx, ot.to_block())
    info = tree.info()
    assert info["n_nodes"] == g["n_nodes"] and info["n_coeffs"] == g["n_coeffs"]
    pts = query_points(O, g["root_min"], g["root_max"])
    got = tree.query(pts)
    assert sha(got) == g["query_sha256"]  # committed fixture
    assert np.array_equal(bits(got), bits(ot.query(pts)))
    lo, hi = np.array(g["root_min"]), np.array(g["root_max"])
    e = (edge_points(np.random.default_rng(5)) + 0.5) * (hi - lo) + lo
    ge, we = tree.query(e), ot.query(e)
    assert np.array_equal(bits(ge), bits(we))
    assert (ge == DBL_MAX).sum() > 0
    assert np.abs(ge[we < 1e300] - we[we < 1e300]).max() <= TOL  # the stated tolerance, for the record


def test_query_all_degrees_and_depths_bitwise(H, O, ctx):
    """Hand-made blocks with leaves of every degree 0..12 (unrolled p <= 5 and the generic path)."""
    rng = np.random.default_rng(42)
    for degs, root in (([0, 1, 2, 3, 4, 5, 6, 7], ((-0.5,) * 3, (0.5,) * 3)),
                       ([8, 9, 10, 11, 12, 6, 7, 2], ((-1.0, 0.0, 2.0), (3.0, 0.5, 2.25)))):
        blk = synthetic_block(rng, degs, depth=2, root_min=root[0], root_max=root[1])
        lo, hi = np.array(root[0]), np.array(root[1])
        pts = (edge_points(rng, 8000) + 0.5) * (hi - lo) + lo
        got = H.DeviceTree(ctx, blk).query(pts)
        want = O.Tree.from_block(blk).query(pts)
        assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("case", ["C2_union3_1e-5", "A1_union3_1e-7_K1024", "A2_sphere_1e-8_K1024"])
def test_query_few_points_bitwise(H, O, ctx, golden, case):
    """Up to 256 points take the single-launch path (query_few_kernel; what a scalar Query(pt) sends): same values as
    the batched kernels and the oracle, for n = 1, a ragged count, and on both sides of the switch."""
    g = golden["blocks"][case]
    blk, _ = H.create_block(ctx, H.make_config(g["target"], g["root_min"], g["root_max"]), product_field(H, g["field"]), g["K"])
    tree, otree = H.DeviceTree(ctx, blk), O.Tree.from_block(blk)
    pts = np.concatenate([O.splitmix64_points(300, seed=77), edge_points(np.random.default_rng(5))])
    want = otree.query(pts)
    big = tree.query(pts)
    assert np.array_equal(bits(big), bits(want))
    for n in (1, 2, 63, 64, 65, 255, 256, 257):
        assert np.array_equal(bits(tree.query(pts[:n])), bits(want[:n])), n
    assert np.array_equal(bits(tree.query(pts[-40:])), bits(want[-40:]))  # boundary / outside points
    # QueryWithGradient has the same switch (query_grad_few_kernel)
    wv, wg = otree.query_with_gradient(pts)
    init = np.full((len(pts), 3), 7.0)  # rows of outside points keep the caller's values
    for n in (1, 64, 65, 256, 257, len(pts)):
        gv, gg = tree.query_with_gradient(pts[:n], init[:n])
        assert np.array_equal(bits(gv), bits(wv[:n])), n
        inside = wv[:n] < 1e300
        assert np.array_equal(bits(gg[inside]), bits(wg[:n][inside])), n
        assert np.all(gg[~inside] == 7.0)


def test_scalar_calls_answered_on_the_host_equal_the_kernels(H, O, ctx, golden):
    """Calls of up to 32 points (a scalar Octree::Query(pt), Octree.cpp:662-702; the loops of HPUnitTests.cpp:64-75) are evaluated
    on the calling thread from the tree handle's copy of the block (csrc/host_query.cpp): the kernels' statements in the kernels'
    order, so the same bits as the batched kernel, as query_few_kernel (calls of 33..256 points) and as the oracle --
    point by point over the edge-point set (cell faces, mid-planes, the root's boundary, outside points, NaN), on refined trees
    and on synthetic trees with leaves of every degree 0..12 down to depth 10, values and gradients."""
    rng = np.random.default_rng(3)
    g = golden["blocks"]["A1_union3_1e-7_K1024"]
    blocks = [H.create_block(ctx, H.make_config(g["target"], g["root_min"], g["root_max"]), product_field(H, g["field"]), g["K"])[0],
              synthetic_block(rng, list(range(13)) * 2, depth=2), deep_chain_block(rng)]
    for blk in blocks:
        tree, otree = H.DeviceTree(ctx, blk), O.Tree.from_block(blk)
        pts = np.concatenate([O.splitmix64_points(500, seed=9), edge_points(np.random.default_rng(5), 1500),
                              np.array([[np.nan, 0.0, 0.0], [0.1, np.inf, 0.0], [0.5, 0.5, 0.5], [-0.5, -0.5, -0.5]])])
        want = otree.query(pts)
        big = tree.query(pts)                                           # batched kernels
        assert np.array_equal(bits(big), bits(want))
        one = np.concatenate([tree.query(pts[i:i + 1]) for i in range(len(pts))])   # host, one point a call
        assert np.array_equal(bits(one), bits(big))
        some = np.concatenate([tree.query(pts[i:i + 32]) for i in range(0, len(pts), 32)])  # host, 32 a call
        assert np.array_equal(bits(some), bits(big))
        wv, wg = otree.query_with_gradient(pts)
        init = np.full((len(pts), 3), 7.0)
        gv = np.empty(len(pts))
        gg = np.empty((len(pts), 3))
        for i in range(0, len(pts), 5):
            gv[i:i + 5], gg[i:i + 5] = tree.query_with_gradient(pts[i:i + 5], init[i:i + 5])
        bv, bg = tree.query_with_gradient(pts, init)
        assert np.array_equal(bits(gv), bits(bv)) and np.array_equal(bits(gg), bits(bg))
        inside = wv < 1e300
        assert np.array_equal(bits(gv), bits(wv)) and np.array_equal(bits(gg[inside]), bits(wg[inside])) and np.all(gg[~inside] == 7.0)
        few = np.concatenate([tree.query(pts[i:i + 40]) for i in range(0, len(pts), 40)])   # 33..256 points: ONE launch of query_few_kernel
        assert np.array_equal(bits(few), bits(big))


@pytest.mark.parametrize("host_build", [False, True])
def test_mesh_scalar_calls_answered_on_the_host_equal_the_kernels(H, O, ctx, monkeypatch, host_build):
    """Mesh::SignedDistanceAtPt(pt, bvh) (Mesh.cpp:54-63) one point at a time -- what a user's SDF lambda written against the
    reference does per sample: calls of one or two points on a plain mesh field are answered on the calling thread from host copies
    of the field's arrays (capi.cpp meshHostMirror, kernels.hip meshEvalHostPoints: the per-point traversal compiled for the
    host from the statements the device runs).  Same bits as the batched device paths and the O(n) scan -- on a smooth mesh, on
    the reference's own mesh, on a needle mesh, for points on vertices / edges / faces, in the medial region, far away, NaN."""
    from helpers import fuzz_mesh_case, hard_points
    if host_build:
        monkeypatch.setenv("HPSDF_MESH_HOST_BUILD", "1")
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "halfedge_fail_mesh.npz"))
    meshes = [icosphere(3, 0.35), (d["verts"], d["tris"].astype(np.uint64)), fuzz_mesh_case(100758)[:2]]
    for k, (verts, tris) in enumerate(meshes):
        f = H.Field.mesh(ctx, verts, tris)
        pts = np.concatenate([hard_points(verts, tris, k)[::5], np.array([[np.nan, 0, 0], [np.inf, 0, 0]])])
        want = f.eval_naive(ctx, pts)                       # the O(n) scan kernel
        big = f.eval(ctx, pts)                              # one batched call: the shared traversal on the device
        assert np.array_equal(bits(big), bits(want))
        one = np.concatenate([f.eval(ctx, pts[i:i + 1]) for i in range(len(pts))])      # host, one point a call
        assert np.array_equal(bits(one), bits(want)), k
        two = np.concatenate([f.eval(ctx, pts[i:i + 2]) for i in range(0, len(pts), 2)])   # host, two points a call
        assert np.array_equal(bits(two), bits(want)), k
        some = np.concatenate([f.eval(ctx, pts[i:i + 32]) for i in range(0, len(pts), 32)])  # from three points on: one launch a call
        assert np.array_equal(bits(some), bits(want)), k
        f.release_host_copies()                             # the copies go, the next scalar call fetches them again
        again = np.concatenate([f.eval(ctx, pts[i:i + 1]) for i in range(0, len(pts), 7)])
        assert np.array_equal(bits(again), bits(want[::7])), k
        f.close()


def test_scalar_calls_from_many_threads_at_once(H, O, ctx, golden):
    """What the reference's users do: an SDF lambda that calls Mesh::SignedDistanceAtPt(pt, bvh) per sample from threadCount threads
    (Source/Tests/MeshingUnitTests.cpp), and Query(pt) loops on several threads.  The host-answered calls share one context, one tree
    handle and one field handle (whose host mirror the first call builds -- here several first calls race for it); every thread must
    get the single-threaded bits."""
    import threading
    verts, tris = icosphere(4, 0.35)
    f = H.Field.mesh(ctx, verts, tris)
    g = golden["blocks"]["A1_union3_1e-7_K1024"]
    tree = H.DeviceTree(ctx, H.create_block(ctx, H.make_config(g["target"]), product_field(H, g["field"]), g["K"])[0])
    pts = O.splitmix64_points(4000, seed=17) * 0.9
    want_m = f.eval_naive(ctx, pts)
    want_q = tree.query(pts)
    nthreads, out_m, out_q, errs = 8, {}, {}, []

    def work(t):
        try:
            idx = range(t, len(pts), nthreads)
            out_m[t] = np.array([f.eval(ctx, pts[i:i + 1])[0] for i in idx])
            out_q[t] = np.array([tree.query(pts[i:i + 1])[0] for i in idx])
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for t in range(nthreads):
        assert np.array_equal(bits(out_m[t]), bits(want_m[t::nthreads])), t
        assert np.array_equal(bits(out_q[t]), bits(want_q[t::nthreads])), t
    f.close()


def test_query_rejects_bad_blocks(H, ctx):
    for bad in (b"", b"\x00" * 50, np.array([1 << 40], np.uint64).tobytes() + b"\x00" * 300):
        with pytest.raises(H.HpsdfError):
            H.DeviceTree(ctx, bad)
    rng = np.random.default_rng(1)
    blk = bytearray(synthetic_block(rng, [2] * 8))
    nco = int(np.frombuffer(bytes(blk[:8]), np.uint64)[0])
    node1 = 8 + 8 * nco + 8 + 56
    blk[node1 + 8:node1 + 12] = np.array([-0.4], np.float32).tobytes()  # child box no longer an octant
    with pytest.raises(H.HpsdfError) as e:
        H.DeviceTree(ctx, bytes(blk))
    assert e.value.status == H.ERR_UNSUPPORTED


def test_query_full_size_properties(H, O, ctx):
    """BASELINE config[1] size: 10 M SplitMix64 points.  The oracle checks a strided sample; the whole
    set is checked through properties: idempotence, permutation equivariance, outside -> DBL_MAX,
    |Query - F| <= 0.05 wherever defined."""
    import torch
    n = 10_000_000
    blk, _ = H.create_block(ctx, H.make_config(1e-5), H.Field.union3(), 1024)
    tree = H.DeviceTree(ctx, blk)
    pts = O.splitmix64_points(n)
    pts[::1000] *= 2.5
    d_in = torch.from_numpy(pts).cuda()
    d_out = torch.empty(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()  # torch's stream produced the inputs; the context launches on its own stream
    tree.query_device(d_in.data_ptr(), n, d_out.data_ptr())
    ctx.synchronize()
    a = d_out.cpu().numpy()
    perm = torch.randperm(n, device="cuda")
    d_in2 = d_in[perm].contiguous()
    d_out2 = torch.empty_like(d_out)
    torch.cuda.synchronize()
    tree.query_device(d_in2.data_ptr(), n, d_out2.data_ptr())
    ctx.synchronize()
    assert torch.equal(d_out2, d_out[perm])
    outside = np.any(np.abs(pts.astype(np.float32)) > np.float32(0.5), axis=1)
    assert np.array_equal(a == DBL_MAX, outside)
    sub = slice(0, n, 997)
    want = O.Tree.from_block(blk).query(pts[sub])
    assert np.array_equal(bits(a[sub]), bits(want))
    inside = ~outside[sub]
    assert np.abs(a[sub][inside] - O.union3_field().eval(pts[sub][inside])).max() <= 0.05


def test_python_octree_mirror(H, O, golden):
    """The host-side mirror of the reference class: Create / Query / To-FromMemoryBlock / copy."""
    cfg = H.make_config(1e-4)
    t = H.Octree(jobs_per_round=1024)
    t.Create(cfg, H.Field.sphere())
    blk = t.ToMemoryBlock()
    assert hashlib.sha256(blk).hexdigest() == golden["blocks"]["C1_sphere_1e-4"]["block_sha256"]
    t2 = H.Octree()
    t2.FromMemoryBlock(blk)
    t3 = t.copy()
    p = O.splitmix64_points(5000)
    q = t.Query(p)
    assert np.array_equal(q, t2.Query(p)) and np.array_equal(q, t3.Query(p))
    assert t.Query([2.0, 0.0, 0.0]) == DBL_MAX and isinstance(t.Query([0.0, 0.1, 0.2]), float)
    # Source/Tests/HPUnitTests.cpp:46-77,115-154: |Query - true| <= 0.01
    assert np.abs(q - O.sphere_field().eval(p)).max() <= 0.01
    t.Clear()
    with pytest.raises(H.HpsdfError):
        t.Query(p)


# ------------------------------------------------------------------ CSG rebuilds
@pytest.mark.parametrize("op_name", ["UnionSDF", "IntersectSDF", "SubtractSDF"])
def test_csg_rebuild_bitwise(H, O, ctx, op_name):
    """Source/Tests/HPUnitTests.cpp:207-282 with a mirrored sphere; target 1e-6 keeps the oracle fast."""
    target, K = 1e-6, 1024
    op = {"UnionSDF": H.OP_UNION, "IntersectSDF": H.OP_INTERSECT, "SubtractSDF": H.OP_SUBTRACT}[op_name]
    t = H.Octree(jobs_per_round=K)
    t.Create(H.make_config(target), H.Field.sphere((0.25, 0, 0), 0.5))
    getattr(t, op_name)(H.Field.sphere((-0.25, 0, 0), 0.5))
    ocfg = O.default_config(target)
    old = O.Tree.create(ocfg, O.sphere_field((0.25, 0, 0), 0.5), K)
    want = O.Tree.create(ocfg, O.TreeCsgField(old, O.sphere_field((-0.25, 0, 0), 0.5), op), K)
    assert t.ToMemoryBlock() == want.to_block()
    p = O.splitmix64_points(20000, seed=77)
    a, b = O.sphere_field((0.25, 0, 0), 0.5).eval(p), O.sphere_field((-0.25, 0, 0), 0.5).eval(p)
    true = {"UnionSDF": np.minimum(a, b), "IntersectSDF": np.maximum(a, b), "SubtractSDF": np.maximum(-a, b)}[op_name]
    assert np.abs(t.Query(p) - true).max() <= 0.05


# ------------------------------------------------------------------ fit known answers, degree by degree
@pytest.mark.parametrize("degree", list(range(2, 12)))
def test_fit_kernels_match_the_oracle_at_every_degree(H, O, ctx, degree):
    """Octree::FitPolynomial (Octree.cpp:1007-1093) cell by cell through hpsdf_fit_cells, in the three fit modes
    (hpsdf_ctx_set_fit_mode).  EXACT: the term-by-term kernel -- degree-specialised bodies for 2..5, the any-degree body for 6..11,
    which no BASELINE-sized build reaches -- returns the oracle's coefficients and error bit for bit.  SPLIT (the default; made to
    split from degree 2 here, 6 is the default): the ERROR and the rows of top degree, which alone enter it (:1062-1069), bit for
    bit; the rows below them, from the sum-factorised kernel, to 1e-13 of the coefficients' scale.  FAST: everything to rounding."""
    depth, n = 3, 40 if degree <= 8 else 12
    cfg, ocfg = H.make_config(1e-5), O.default_config(1e-5)
    exact = H.Context(0)
    exact.set_fit_mode(H.FIT_EXACT)
    got_c, got_e = H.fit_cells(exact, cfg, product_field(H, "union3"), degree, depth, n)
    exact.close()
    of = oracle_field(O, "union3")
    side, h = 1 << depth, np.float32(1.0) / np.float32(1 << depth)
    for i in range(n):
        ix, iy, iz = i % side, (i // side) % side, i // (side * side)
        bmin = np.array([np.float32(-0.5) + np.float32(k) * h for k in (ix, iy, iz)], np.float32)
        want_c, want_e = O.fit_polynomial(of, ocfg, bmin, bmin + h, degree, depth)
        assert np.array_equal(bits(got_c[i]), bits(want_c)), (degree, i)
        assert bits(np.array([got_e[i]]))[0] == bits(np.array([want_e]))[0], (degree, i)
    scale = np.abs(got_c).max()
    # the default context: splits from degree 6; below that it IS the exact kernel
    dc, de = H.fit_cells(ctx, cfg, product_field(H, "union3"), degree, depth, n)
    assert ctx.fit_mode() == H.FIT_SPLIT
    assert np.array_equal(bits(de), bits(got_e))
    if degree < 6:
        assert np.array_equal(bits(dc), bits(got_c))
    split = H.Context(0)
    split.set_split_min_degree(2)   # every from-scratch fit split: the rows below the top degree by csrc/fit_low.hip (sum factorisation)
    for c, (sc, se) in ((ctx, (dc, de)), (split, H.fit_cells(split, cfg, product_field(H, "union3"), degree, depth, n))):
        nlow = int(H.NCOEF[degree - 1])
        assert np.array_equal(bits(se), bits(got_e)), degree                                  # errors: bit for bit
        assert np.array_equal(bits(sc[:, nlow:]), bits(got_c[:, nlow:])), degree               # rows of top degree: bit for bit
        assert np.abs(sc[:, :nlow] - got_c[:, :nlow]).max() <= 1e-13 * scale, degree           # the rows below: the matrix cores
    if degree >= 4:
        assert not np.array_equal(sc, got_c)  # (the matrix-core kernel did run)
    split.close()
    fast = H.Context(0)
    fast.set_fast_fit(True)
    fc, fe = H.fit_cells(fast, cfg, product_field(H, "union3"), degree, depth, n)
    fast.close()
    assert np.abs(fc - got_c).max() <= 1e-13 * scale
    assert np.abs(fe - got_e).max() <= 1e-9 * got_e.max() + 1e-30
    assert not np.array_equal(fc, got_c)  # (another arithmetic: the matrix-core kernel did run)


@pytest.fixture
def left_assoc(H, O):
    """Both sides under the OTHER reading of Eigen's 3-vector reductions, (a . b) . c; restored afterwards (the switch is process-wide)."""
    H.set_reduction_order(1)
    O.set_reduction_order(1)
    yield
    H.set_reduction_order(0)
    O.set_reduction_order(0)


def test_reduction_order_switch_matches_the_oracle(H, O, ctx, golden, left_assoc, monkeypatch):
    """hpsdf_set_reduction_order(1): which way Eigen associates prod() / norm() / normalize() of a Vector3d depends on how the
    reference's Eigen was built (include/hpsdf.h); with the switch thrown on both sides the product is still the oracle bit for bit --
    field values, single fits, whole blocks from both schedulers, gradients from the kernels and from the host-answered scalar calls --
    and it is not a no-op: the blocks differ from the default order's."""
    assert H.reduction_order() == 1 and O.reduction_order() == 1
    pts = O.splitmix64_points(200000, seed=3)
    for name in ("sphere", "union3"):
        assert np.array_equal(bits(product_field(H, name).eval(ctx, pts)), bits(oracle_field(O, name).eval(pts)))
    spec = [(O.PRIM_BOX, O.OP_UNION, [0.1, 0.0, -0.1, 0.2, 0.15, 0.1]), (O.PRIM_SPHERE, O.OP_SUBTRACT, [0.2, 0.1, 0.0, 0.12])]
    assert np.array_equal(bits(H.Field.analytic(spec).eval(ctx, pts * 1.5)), bits(O.AnalyticField(spec).eval(pts * 1.5)))
    # single fits: the degree-specialised bodies, the any-degree body and (default context, degree >= 6) the split kernels' exact half
    exact = H.Context(0)
    exact.set_fit_mode(H.FIT_EXACT)
    cfg, ocfg, of = H.make_config(1e-5), O.default_config(1e-5), oracle_field(O, "union3")
    depth, n = 3, 12
    side, h = 1 << depth, np.float32(1.0) / np.float32(1 << depth)
    for degree in (2, 3, 4, 5, 7, 9):
        got_c, got_e = H.fit_cells(exact, cfg, product_field(H, "union3"), degree, depth, n)
        dc, de = H.fit_cells(ctx, cfg, product_field(H, "union3"), degree, depth, n)
        nlow = int(H.NCOEF[degree - 1])
        for i in range(n):
            ix, iy, iz = i % side, (i // side) % side, i // (side * side)
            bmin = np.array([np.float32(-0.5) + np.float32(k) * h for k in (ix, iy, iz)], np.float32)
            want_c, want_e = O.fit_polynomial(of, ocfg, bmin, bmin + h, degree, depth)
            assert np.array_equal(bits(got_c[i]), bits(want_c)), (degree, i)
            assert bits(np.array([got_e[i]]))[0] == bits(np.array([want_e]))[0] == bits(np.array([de[i]]))[0], (degree, i)
            assert np.array_equal(bits(dc[i, nlow:]), bits(want_c[nlow:])), (degree, i)
        assert np.abs(dc - got_c).max() <= 1e-13 * np.abs(got_c).max()
    # whole blocks: device frontier and host scheduler, exact mode: the oracle's bytes; the default (split) mode: its nodes and statistics
    for case in ("C2_union3_1e-5", "A1_union3_1e-7_K1024", "D1_sphere075_customroot_1e-6"):
        g = golden["blocks"][case]
        hcfg = H.make_config(g["target"], g["root_min"], g["root_max"])
        ot = O.Tree.create(O.default_config(g["target"], g["root_min"], g["root_max"]), oracle_field(O, g["field"]), g["K"])
        ob = ot.to_block()
        blk, st = H.create_block(exact, hcfg, product_field(H, g["field"]), g["K"])
        assert blk == ob, case
        assert hashlib.sha256(blk).hexdigest() != g["block_sha256"], case   # not the default order's block
        monkeypatch.setenv("HPSDF_HOST_FRONTIER", "1")
        assert H.create_block(exact, hcfg, product_field(H, g["field"]), g["K"])[0] == ob, case
        monkeypatch.setenv("HPSDF_HOST_FRONTIER", "0")
        sblk, sst = H.create_block(ctx, hcfg, product_field(H, g["field"]), g["K"])
        a, b = O.parse_block(sblk), O.parse_block(ob)
        assert np.array_equal(a["degree"], b["degree"]) and np.array_equal(a["childIdx"], b["childIdx"]), case
        assert all(sst[k] == st[k] for k in st if k not in ("fit_mode", "split_fits")), case
        assert np.abs(a["coeffs"] - b["coeffs"]).max() <= 1e-12, case
        # gradients: kernels (many points), the host-answered scalar calls (<= 32 points) -- and the values, which do not depend on it
        tree = H.DeviceTree(ctx, ob)
        qp = np.concatenate([O.splitmix64_points(30000, seed=31), edge_points(np.random.default_rng(2), 1000)])
        init = np.full((len(qp), 3), 7.0)
        gv, gg = tree.query_with_gradient(qp, init)
        wv, wg = ot.query_with_gradient(qp, init)
        assert np.array_equal(bits(gv), bits(wv)) and np.array_equal(bits(gg), bits(wg)), case
        for lo in range(0, 320, 32):
            sv, sg = tree.query_with_gradient(qp[lo:lo + 32], init[lo:lo + 32])
            assert np.array_equal(bits(sv), bits(wv[lo:lo + 32])) and np.array_equal(bits(sg), bits(wg[lo:lo + 32])), (case, lo)
    exact.close()
    rng = np.random.default_rng(4)
    blk = synthetic_block(rng, [8, 9, 10, 11, 12, 6, 7, 2], depth=2)
    p2 = rng.uniform(-0.5, 0.5, (3000, 3))
    a, b = H.DeviceTree(ctx, blk).query_with_gradient(p2), O.Tree.from_block(blk).query_with_gradient(p2)
    assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(bits(a[1]), bits(b[1]))
    # the other kernel families: sampled fields (a host callback; anisotropic root, so that the cell-scale product has three different
    # factors), mesh fields (the sampler feeds the fit from samples) and the tree-CSG wrapper around an analytic field
    import math

    def sphere(pt, thread_idx):  # the user's own Eigen: here the switched order, as oracle's sphere under ora_set_reduction_order(1)
        dx, dy, dz = pt[0] - 0.05, pt[1], pt[2] + 0.02
        return math.sqrt((dx * dx + dy * dy) + dz * dz) - 0.3

    root = ((-0.4, -0.45, -0.5), (0.5, 0.4, 0.45))
    cb, _ = H.create_block(ctx, H.make_config(1e-6, *root), H.Field.callback(sphere), 1024)
    want = O.Tree.create(O.default_config(1e-6, *root), O.sphere_field((0.05, 0.0, -0.02), 0.3), 1024).to_block()
    assert cb == want
    verts, tris = icosphere(1, 0.3)
    mroot = ((-0.4, -0.4, -0.4), (0.4, 0.45, 0.4))
    mb, _ = H.create_block(ctx, H.make_config(1e-4, *mroot), H.Field.mesh(ctx, verts, tris), 1024)
    assert mb == O.Tree.create(O.default_config(1e-4, *mroot), O.MeshField(verts, tris), 1024).to_block()
    t = H.Octree(jobs_per_round=1024)
    t.Create(H.make_config(1e-6), H.Field.sphere((0.25, 0, 0), 0.5))
    t.SubtractSDF(H.Field.sphere((-0.25, 0, 0), 0.5))
    ocfg = O.default_config(1e-6)
    oldt = O.Tree.create(ocfg, O.sphere_field((0.25, 0, 0), 0.5), 1024)
    assert t.ToMemoryBlock() == O.Tree.create(ocfg, O.TreeCsgField(oldt, O.sphere_field((-0.25, 0, 0), 0.5), H.OP_SUBTRACT), 1024).to_block()
    O.set_reduction_order(0)
    assert not np.array_equal(bits(O.Tree.from_block(blk).query_with_gradient(p2)[1]), bits(b[1]))  # normalize() did change
    assert cb != O.Tree.create(O.default_config(1e-6, *root), O.sphere_field((0.05, 0.0, -0.02), 0.3), 1024).to_block()


# ------------------------------------------------------------------ mesh field (SURVEY 8 a-M)
def test_device_acosf_is_the_host_libms(H, O, ctx):
    """The one libm call of the mesh path (Mesh.cpp:226-231, the angle weights of a vertex pseudo-normal): the device runs
    the host libm's algorithm (csrc/acosf_host_libm.hpp; tests/test_product_cpu.py sweeps its host compilation over every
    float) -- here the DEVICE's results, every 61st float of [-1, 1], dense windows where the branches meet, and outside."""
    one = 0x3F800000
    for first, stride, n in ((0, 61, one // 61 + 1), (0x80000000, 61, one // 61 + 1),  # [0, 1] and [-0, -1]
                             (0x3F000000 - 50000, 1, 100000), (0xBF000000 - 50000, 1, 100000),  # around +-0.5
                             (one - 200000, 1, 200100), (0x80000000 + one - 200000, 1, 200100),  # up to +-1 and just beyond
                             (0x23000000 - 1000, 1, 2000), (0, 1, 4096)):  # the 2^-57 cut, zero and denormals
        got, want = H.selftest_acosf(ctx, first, stride, n), O.acosf_batch(first, stride, n)
        assert np.array_equal(got.view(np.uint32)[~np.isnan(want)], want.view(np.uint32)[~np.isnan(want)])
        assert np.array_equal(np.isnan(got), np.isnan(want))


def test_mesh_field_matches_naive_oracle(H, O, ctx):
    verts, tris = icosphere(2, 0.35, (0.05, -0.02, 0.01))
    mf, of = H.Field.mesh(ctx, verts, tris), O.MeshField(verts, tris)
    pts = O.splitmix64_points(20000, seed=9)
    got = mf.eval(ctx, pts)
    want, tri, simp = of.signed_distance(pts)
    # f32 path; SURVEY H4's bar is |delta| <= 1e-6 and identical sign.  The device runs the reference's f32 operations in
    # the reference's order, unfused, and the acosf of the host's libm (test_device_acosf_is_the_host_libms): same bits
    assert np.array_equal(bits(got), bits(want.astype(np.float64)))
    assert set(np.unique(simp // 4)) == {0, 1, 2}  # vertex, edge and face regions all exercised
    r = np.linalg.norm(pts - np.array([0.05, -0.02, 0.01]), axis=1) - 0.35
    assert np.abs(got - r).max() < 0.02  # it is a sphere, to faceting error


def test_torus_mesh_field_and_create_match_oracle(H, O, ctx):
    """A genus-1 mesh with concave regions (the displaced torus that stands in for the north_star's 2 M-triangle mesh,
    here 48 x 32 x 2 triangles): point values against the naive scan, and the tree built from it against the oracle's."""
    verts, tris = displaced_torus(48, 32)
    mf, of = H.Field.mesh(ctx, verts, tris), O.MeshField(verts, tris)
    pts = O.splitmix64_points(20000, seed=21)
    got = mf.eval(ctx, pts)
    want, tri, simp = of.signed_distance(pts)
    assert np.array_equal(bits(got), bits(want.astype(np.float64)))
    assert (got < 0).mean() > 0.02  # the tube's inside is seen
    verts, tris = displaced_torus(16, 12)  # the oracle scans every triangle for every sample: keep it small
    blk, st = H.create_block(ctx, H.make_config(1e-4), H.Field.mesh(ctx, verts, tris), 1024)
    ot = O.Tree.create(O.default_config(1e-4), O.MeshField(verts, tris), 1024)
    assert blk == ot.to_block()  # byte for byte, as for analytic fields


def test_mesh_open_mesh_rejected(H, ctx):
    verts, tris = icosphere(1)
    with pytest.raises(H.HpsdfError) as e:
        H.Field.mesh(ctx, verts, tris[:-1])
    assert e.value.status == H.ERR_OPEN_MESH


def test_create_from_mesh_field_matches_oracle(H, O, ctx):
    verts, tris = icosphere(1, 0.3)
    cfg_root = ((-0.4, -0.4, -0.4), (0.4, 0.45, 0.4))  # anisotropic root = "mesh AABB"-like
    blk, st = H.create_block(ctx, H.make_config(1e-4, *cfg_root), H.Field.mesh(ctx, verts, tris), 1024)
    ot = O.Tree.create(O.default_config(1e-4, *cfg_root), O.MeshField(verts, tris), 1024)
    assert blk == ot.to_block()


def _reference_mesh():
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "halfedge_fail_mesh.npz"))
    return d["verts"], d["tris"].astype(np.uint64)


def test_reference_mesh_field_matches_naive_oracle(H, O, ctx):
    """Resources/halfedge_fail.obj (the one mesh the reference ships; 22 840 triangles, a thin wire: the closest
    feature is almost always an edge or a vertex, i.e. the pseudo-normal paths)."""
    verts, tris = _reference_mesh()
    mf, of = H.Field.mesh(ctx, verts, tris), O.MeshField(verts, tris)
    lo, hi = verts.min(0).astype(np.float64), verts.max(0).astype(np.float64)
    pts = np.random.default_rng(17).uniform(lo - 0.05, hi + 0.05, (3000, 3))
    got = mf.eval(ctx, pts)
    want, tri, simp = of.signed_distance(pts)
    assert np.array_equal(bits(got), bits(want.astype(np.float64)))
    assert (simp // 4 == 0).any() and (simp // 4 == 1).any()  # vertex fans (the acosf-weighted normals) and edges decide signs here


def test_create_on_reference_mesh(H, ctx):
    """BASELINE config[2] shape with the mesh that exists: root = mesh AABB (anisotropic), targetError 1e-5."""
    verts, tris = _reference_mesh()
    lo, hi = verts.min(0) - 0.01, verts.max(0) + 0.01
    f = H.Field.mesh(ctx, verts, tris)
    t = H.Octree(jobs_per_round=1024)
    t.Create(H.make_config(1e-5, tuple(lo), tuple(hi)), f)
    assert t.stats["n_nodes"] >= 4681 and t.stats["samples"] >= 2985984
    pts = np.random.default_rng(3).uniform(lo, hi, (20000, 3))
    q, v = t.Query(pts), f.eval(ctx, pts)
    # the field is only C0 across the wire's medial axis; the reference's own bar for smooth fields is 1e-2
    assert np.median(np.abs(q - v)) < 2e-3 and np.mean(np.abs(q - v) < 2e-2) > 0.97
    blk2, _ = H.create_block(ctx, H.make_config(1e-5, tuple(lo), tuple(hi)), f, 1024)
    assert blk2 == t.ToMemoryBlock()  # run-to-run deterministic


def test_mesh_sampling_paths_build_identical_trees(H, ctx, monkeypatch):
    """Mesh fields are sampled by mesh_sample_kernel and fitted from the samples; past 2^30 samples per round the fit
    kernel samples for itself (HPSDF_MESH_FUSED=1 forces that).  Same samples, same fit: byte-identical blocks,
    with refinement rounds (1e-7: 11 rounds, H and P fits of degrees 2-3) and with a csg wrapper on top."""
    verts, tris = icosphere(3, 0.3)
    f = H.Field.mesh(ctx, verts, tris)
    cfg = H.make_config(1e-7)
    blk_a, st_a = H.create_block(ctx, cfg, f, 256)
    base = H.DeviceTree(ctx, H.create_block(ctx, H.make_config(1e-4), H.Field.union3(), 1024)[0])
    csg = H.Field.tree_csg(base, H.OP_UNION, f)
    blk_c, _ = H.create_block(ctx, H.make_config(1e-5), csg, 1024)
    monkeypatch.setenv("HPSDF_MESH_FUSED", "1")
    blk_b, st_b = H.create_block(ctx, cfg, f, 256)
    blk_d, _ = H.create_block(ctx, H.make_config(1e-5), csg, 1024)
    assert st_a["rounds"] > 1 and st_a["jobs"] == st_b["jobs"]
    assert blk_a == blk_b and blk_c == blk_d


# ------------------------------------------------------------------ nearness weighting (SURVEY 8 a-W)
@pytest.mark.parametrize("wtype,target", [(1, 1e-8), (2, 1e-10)])
def test_weighted_create_matches_oracle(H, O, ctx, wtype, target):
    """Source/Tests/HPUnitTests.cpp:53-58 (Polynomial, strength 3, 1e-8) and HPBenchmarks.cpp:34-39 (Exponential, 3,
    1e-10).  The reference samples the weight with std::rand (unpinnable); oracle and GPU share a hashed sampler, so
    they must agree bit for bit with each other, and the reference's accuracy bar must hold."""
    cfg = H.make_config(target)
    cfg.nearnessWeighting_type, cfg.nearnessWeighting_strength = wtype, 3.0
    blk, st = H.create_block(ctx, cfg, H.Field.sphere(), 1024)
    ocfg = O.default_config(target)
    ocfg.weighting_type, ocfg.weighting_strength = wtype, 3.0
    ot = O.Tree.create(ocfg, O.sphere_field(), 1024)
    assert blk == ot.to_block()
    assert st["jobs"] == ot.stats["jobs"] and st["h_refines"] == ot.stats["h_refines"]
    p = O.splitmix64_points(100000, seed=8)
    assert np.abs(H.DeviceTree(ctx, blk).query(p) - O.sphere_field().eval(p)).max() <= 0.01


def test_query_with_gradient_bitwise(H, O, ctx, golden):
    """Octree::QueryWithGradient (Octree.cpp:749-789, 904-985; reference benchmark HPBenchmarks.cpp:169-203)."""
    g = golden["blocks"]["A1_union3_1e-7_K1024"]
    ot = O.Tree.create(O.default_config(g["target"]), O.union3_field(), g["K"])
    tree = H.DeviceTree(ctx, ot.to_block())
    pts = np.concatenate([O.splitmix64_points(50000, seed=31), edge_points(np.random.default_rng(2), 2000)])
    init = np.full((len(pts), 3), 7.0)
    gv, gg = tree.query_with_gradient(pts, init)
    wv, wg = ot.query_with_gradient(pts, init)
    assert np.array_equal(bits(gv), bits(wv)) and np.array_equal(bits(gg), bits(wg))
    outside = gv == DBL_MAX
    assert outside.sum() > 0 and np.all(gg[outside] == 7.0)  # untouched, like the reference's output argument
    assert np.allclose(np.linalg.norm(gg[~outside], axis=1), 1.0, atol=1e-12)
    assert np.array_equal(bits(gv), bits(tree.query(pts)))
    rng = np.random.default_rng(4)
    blk = synthetic_block(rng, [8, 9, 10, 11, 12, 6, 7, 2], depth=2)
    p2 = rng.uniform(-0.5, 0.5, (3000, 3))
    a, b = H.DeviceTree(ctx, blk).query_with_gradient(p2), O.Tree.from_block(blk).query_with_gradient(p2)
    assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(bits(a[1]), bits(b[1]))
    # degrees that stop at 5: the deferred points' second pass with the degree at compile time (degree 4 and degree 5 leaves, both depths)
    for degs in ([5, 4, 3, 5, 2, 4, 5, 1], [4, 4, 2, 3, 4, 0, 3, 4]):
        for depth in (1, 2):
            blk = synthetic_block(rng, degs, depth=depth)
            dt = H.DeviceTree(ctx, blk)
            assert dt.info()["max_degree"] == max(degs)
            a, b = dt.query_with_gradient(p2), O.Tree.from_block(blk).query_with_gradient(p2)
            assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(bits(a[1]), bits(b[1]))
            assert np.array_equal(bits(dt.query(p2)), bits(a[0]))
    # trees whose leaves all sit in the top table with degree <= 2 (the BASELINE thresholds' trees) take query_grad_kernel, one line a
    # point like Query: the two headline trees (top level at depth 4), and synthetic ones with the top level elsewhere and degrees 0..2
    cases = [O.Tree.create(O.default_config(golden["blocks"][c]["target"]), oracle_field(O, golden["blocks"][c]["field"]), 1024).to_block()
             for c in ("C1_sphere_1e-4", "C2_union3_1e-5")]
    cases += [synthetic_block(rng, [2, 1, 0, 2, 2, 1, 2, 0], depth=1), synthetic_block(rng, [2] * 8, depth=1)]
    for blk in cases:
        dt, ot = H.DeviceTree(ctx, blk), O.Tree.from_block(blk)
        gv, gg = dt.query_with_gradient(pts, init)
        wv, wg = ot.query_with_gradient(pts, init)
        assert np.array_equal(bits(gv), bits(wv)) and np.array_equal(bits(gg), bits(wg))
        assert np.array_equal(bits(gv), bits(dt.query(pts)))
        assert np.all(gg[gv == DBL_MAX] == 7.0)
        os.environ["HPSDF_QUERY_GRAD_GENERAL"] = "1"   # the any-tree kernel on the same tree: same bits
        try:
            ov, og = dt.query_with_gradient(pts, init)
        finally:
            del os.environ["HPSDF_QUERY_GRAD_GENERAL"]
        assert np.array_equal(bits(ov), bits(gv)) and np.array_equal(bits(og), bits(gg))


def test_csg_with_host_callback_inner_field(H, O, ctx):
    """UnionSDF(std::function): F' = min(old.Query, F_) with F_ sampled on the host and old.Query on the GPU."""
    import math
    target, K = 1e-5, 1024

    def other(pt, thread_idx):
        dx, dy, dz = pt[0] + 0.25, pt[1], pt[2]
        return math.sqrt(dx * dx + (dy * dy + dz * dz)) - 0.5

    t = H.Octree(jobs_per_round=K)
    t.Create(H.make_config(target, threads=2), H.Field.sphere((0.25, 0, 0), 0.5))
    t.UnionSDF(other)
    ocfg = O.default_config(target)
    old = O.Tree.create(ocfg, O.sphere_field((0.25, 0, 0), 0.5), K)
    want = O.Tree.create(ocfg, O.TreeCsgField(old, O.sphere_field((-0.25, 0, 0), 0.5), O.OP_UNION), K)
    got = bytearray(t.ToMemoryBlock())
    got[-80 + 48:-80 + 56] = np.array([1], np.uint64).tobytes()  # threadCount differs (2 vs 1), nothing else may
    assert bytes(got) == want.to_block()


def test_polynomial_field_is_represented_exactly(H, O, ctx):
    """A plane is inside the degree-2 space: every leaf stays at degree 2 and Query reproduces it to rounding."""
    spec = [(H.PRIM_PLANE, H.OP_UNION, [0.3, -0.2, 0.5, 0.1])]
    blk, st = H.create_block(ctx, H.make_config(1e-12), H.Field.analytic(spec), 1024)
    assert st["n_nodes"] == 4681 and st["rounds"] == 1
    assert blk == O.Tree.create(O.default_config(1e-12), O.AnalyticField(spec), 1024).to_block()
    p = O.splitmix64_points(100000, seed=6)
    q = H.DeviceTree(ctx, blk).query(p)
    assert np.abs(q - (0.3 * p[:, 0] - 0.2 * p[:, 1] + 0.5 * p[:, 2] + 0.1)).max() < 1e-13


def test_capi_argument_checks(H, ctx):
    L = H.lib()
    import ctypes as C
    assert L.hpsdf_query_device(None, None, None, 0, None) == H.ERR_NO_DEVICE
    assert L.hpsdf_query_device(ctx.handle, None, None, 5, None) == 1  # HPSDF_ERR_INVALID_ARGUMENT
    assert L.hpsdf_tree_upload(ctx.handle, None, 0, C.byref(C.c_void_p())) == 4  # HPSDF_ERR_BAD_BLOCK
    assert L.hpsdf_field_eval_host(ctx.handle, None, None, 0, None) == 1
    assert b"" != L.hpsdf_last_error()
    tmp = H.Context(0)   # context settings: ranges checked, nothing silently clamped
    for bad in (1, 13, -3):
        assert L.hpsdf_ctx_set_split_min_degree(tmp.handle, bad) == 1
    for ok in (2, 6, 12):
        assert L.hpsdf_ctx_set_split_min_degree(tmp.handle, ok) == 0
    assert L.hpsdf_ctx_set_fit_mode(tmp.handle, 3) == 1 and L.hpsdf_ctx_set_fit_mode(tmp.handle, H.FIT_EXACT) == 0 and tmp.fit_mode() == H.FIT_EXACT
    assert L.hpsdf_ctx_set_split_min_degree(None, 6) == 1
    tmp.close()
    assert H.reduction_order() == 0   # (the process-wide default: a . (b . c))
    blk, _ = H.create_block(ctx, H.make_config(1e-4), H.Field.sphere(), 0)  # K = 0 -> default
    tree = H.DeviceTree(ctx, blk)
    assert tree.query(np.zeros((0, 3))).shape == (0,)
    assert tree.query(np.array([[0.1, 0.2, 0.3]])).shape == (1,)
    f = H.Field.callback(lambda p, t: 0.0)
    with pytest.raises(H.HpsdfError) as e:
        f.eval(ctx, np.zeros((4, 3)))
    assert e.value.status == H.ERR_UNSUPPORTED


# ------------------------------------------------------------------ QueryRay / OutputFunctionSlice
def _ray_set(rng, n):
    o = rng.uniform(-0.6, 0.6, (n, 3))           # most inside the root, some outside
    o[: n // 8] = rng.uniform(-3.0, 3.0, (n // 8, 3))  # far outside: IntersectAABB path
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[n // 8: n // 8 + 16, 1] = 0.0              # axis-parallel components: infinite slab parameters
    d[n // 8 + 16: n // 8 + 32] = [1.0, 0.0, 0.0]
    tmax = rng.uniform(0.05, 3.0, n)
    return o, d, tmax


@pytest.mark.parametrize("case", ["C1_sphere_1e-4", "A1_union3_1e-7_K1024", "D1_sphere075_customroot_1e-6"])
def test_query_ray_bitwise(H, O, ctx, golden, case):
    g = golden["blocks"][case]
    blk = O.Tree.create(O.default_config(g["target"], g["root_min"], g["root_max"]), oracle_field(O, g["field"]), g["K"]).to_block()
    o, d, tmax = _ray_set(np.random.default_rng(11), 20000)
    lo, hi = np.array(g["root_min"]), np.array(g["root_max"])
    o = (o + 0.5) * (hi - lo) + lo
    init = np.full(len(o), -123.0)
    hit, t = H.DeviceTree(ctx, blk).query_ray(o, d, tmax, t_init=init)
    whit, wt = O.Tree.from_block(blk).query_ray(o, d, tmax, t_init=init)
    assert np.array_equal(hit, whit)
    assert np.array_equal(bits(t), bits(wt))
    assert 0 < hit.sum() < len(hit)
    assert np.all(t[hit == 0] == -123.0)  # t_ untouched on a miss (Octree.cpp:705-746)
    # a scalar QueryRay(ray, tMax, t) -- calls of up to 32 rays -- is stepped on the calling thread (csrc/host_query.cpp): same answers
    tree = H.DeviceTree(ctx, blk)
    sel = np.concatenate([np.arange(0, 400), np.arange(2490, 2540), np.arange(5000, 6500)])  # outside origins, axis-parallel, inside
    so, sd, sm, si = o[sel], d[sel], tmax[sel], init[sel]
    one = [tree.query_ray(so[i:i + 1], sd[i:i + 1], sm[i:i + 1], t_init=si[i:i + 1]) for i in range(len(sel))]
    assert np.array_equal(np.concatenate([h for h, _ in one]), whit[sel]) and np.array_equal(bits(np.concatenate([v for _, v in one])), bits(wt[sel]))
    some = [tree.query_ray(so[i:i + 32], sd[i:i + 32], sm[i:i + 32], t_init=si[i:i + 32]) for i in range(0, len(sel), 32)]
    assert np.array_equal(np.concatenate([h for h, _ in some]), whit[sel]) and np.array_equal(bits(np.concatenate([v for _, v in some])), bits(wt[sel]))
    assert 0 < whit[sel].sum() < len(sel)


def test_function_slice_bitwise(H, O, ctx, golden, tmp_path):
    g = golden["blocks"]["C2_union3_1e-5"]
    blk = O.Tree.create(O.default_config(g["target"]), oracle_field(O, g["field"]), g["K"]).to_block()
    dt, ot = H.DeviceTree(ctx, blk), O.Tree.from_block(blk)
    for c, vmin, vmax, n in ((0.1, (-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), 256),
                             (-0.2, (-0.7, -0.3, 0.0), (0.4, 0.9, 0.0), 200)):  # partly outside the root
        rgb, vals = dt.function_slice(c, vmin, vmax, n)
        wrgb, wvals = ot.function_slice(c, vmin, vmax, n)
        assert np.array_equal(bits(vals), bits(wvals))
        assert np.array_equal(rgb, wrgb)
    # the reference's 2048^2 image through the Octree mirror, written as a BMP
    oc = H.Octree()
    oc.FromMemoryBlock(blk)
    rgb = oc.OutputFunctionSlice(str(tmp_path / "slice"), 0.1, (-0.5, -0.5, -0.5), (0.5, 0.5, 0.5))
    raw = open(str(tmp_path / "slice.bmp"), "rb").read()
    assert raw[:2] == b"BM" and len(raw) == 54 + 2048 * 2048 * 3
    assert rgb.shape == (2048, 2048, 3) and rgb[..., 1].max() >= 250 and rgb[..., 2].max() >= 250
    # bottom-up BGR rows: the file's first pixel is image row 2047, column 0
    assert raw[54:57] == bytes(rgb[2047, 0, ::-1])


# ------------------------------------------------------------------ Create with continuity.enforce
@pytest.mark.parametrize("field,target,rmin,rmax", [("sphere", 1e-8, (-0.5,) * 3, (0.5,) * 3),
                                                    ("union3", 1e-7, (-0.5,) * 3, (0.5,) * 3)])
def test_create_with_continuity_matches_oracle(H, O, ctx, field, target, rmin, rmax):
    """GPU build + host post-process (Octree.cpp:341-344) == oracle build + oracle post-process: topology
    identical, coefficients within the north_star's 1e-6 (both CG runs stop at the reference's relative 1e-6)."""
    cfg = H.make_config(target, rmin, rmax, continuity=True)
    blk, st = H.create_block(ctx, cfg, product_field(H, field), 1024)
    cs = H.continuity_last_stats()
    assert cs["n_pairs"] > 0 and cs["iterations"] > 0 and cs["residual"] < 1e-6 and cs["jump_after"] < cs["jump_before"]
    ocfg = O.default_config(target, rmin, rmax, continuity=True)
    t = O.Tree.create(ocfg, oracle_field(O, field), 1024)
    raw = t.to_block()
    so = t.continuity_post_process(1e-6)
    a, b = O.parse_block(blk), O.parse_block(t.to_block())
    assert blk[8 + 8 * a["n_coeffs"]:] == raw[8 + 8 * a["n_coeffs"]:]           # nodes + config: bit-identical
    assert np.abs(a["coeffs"] - b["coeffs"]).max() <= TOL
    assert cs["n_pairs"] == so["n_pairs"] and cs["nnz"] == so["nnz"]
    # the pre-continuity GPU block is the oracle's, so post-processing it on the host reproduces Create's output exactly
    cfg0 = H.make_config(target, rmin, rmax, continuity=False)
    blk0, _ = H.create_block(ctx, cfg0, product_field(H, field), 1024)
    b0 = bytearray(blk0)
    b0[-80 + 16] = 1
    assert H.continuity_post_process(bytes(b0))[0] == blk
    # and the reference's own acceptance test for this path: |Query - true| <= 1e-2 (HPUnitTests.cpp:80-112)
    if field == "sphere":
        pts = O.splitmix64_points(300000, seed=9)
        true = np.linalg.norm(pts - np.array([0.25, 0, 0]), axis=1) - 0.5
        assert np.abs(H.DeviceTree(ctx, blk).query(pts) - true).max() <= 1e-2


@pytest.mark.parametrize("case", ["A1_union3_1e-7_K1024", "A2_sphere_1e-8_K1024"])
def test_query_refined_trees_many_points_bitwise(H, O, ctx, golden, case):
    """The wave-cooperative general kernel over many workgroups and a ragged last tile: leaves of degree 2 and 3 at
    depths 4-6 fetched by the wave, leaves of degree 4 through the per-workgroup deferred lists (no global atomics)."""
    g = golden["blocks"][case]
    ot = O.Tree.create(O.default_config(g["target"]), oracle_field(O, g["field"]), g["K"])
    tree = H.DeviceTree(ctx, ot.to_block())
    assert tree.info()["max_degree"] == 4
    n = 2_100_003
    pts = O.splitmix64_points(n, seed=77)
    pts[::1000] *= 2.5  # some outside the root
    got, want = tree.query(pts), ot.query(pts)
    assert np.array_equal(bits(got), bits(want))
    assert (got == DBL_MAX).sum() > 100
    # (that was one launch: degrees 4-5 are finished by the workgroup that met them; the scan + second-pass route, which trees with
    # higher degrees take, on the same tree)
    os.environ["HPSDF_QUERY_TWO_PASS"] = "1"
    try:
        assert np.array_equal(bits(tree.query(pts)), bits(want))
    finally:
        del os.environ["HPSDF_QUERY_TWO_PASS"]
    gv, gg = tree.query_with_gradient(pts[:700_001])
    wv, wg = ot.query_with_gradient(pts[:700_001])
    assert np.array_equal(bits(gv), bits(wv)) and np.array_equal(bits(gg), bits(wg))


@pytest.mark.parametrize("case,root", [("C2_union3_1e-5", None), ("D1_sphere075_customroot_1e-6", "custom")])
def test_query_ordered_point_sets_bitwise(H, O, ctx, golden, case, root):
    """query_kernel's paths for ORDERED input (round 6): a wave whose 64 points lie in one cell takes its row through the scalar cache; a
    wave with 2..32 runs of equal cells fetches one row a run; everything else goes the way random points go.  All of them return
    the tree's values bit for bit: whole waves in one cell (cell-sorted), z-fastest grids (a handful of runs a wave), runs of exactly
    2 (32 runs: the run path's four steps) and of 1-2 (33+ runs: the other path), runs that straddle tiles, points outside the root
    and NaNs inside runs, a ragged tail; Query and QueryWithGradient (the same body) alike."""
    g = golden["blocks"][case]
    ot = O.Tree.create(O.default_config(g["target"], g["root_min"], g["root_max"]), oracle_field(O, g["field"]), g["K"])
    tree = H.DeviceTree(ctx, ot.to_block())
    lo, hi = np.array(g["root_min"], np.float64), np.array(g["root_max"], np.float64)
    rng = np.random.default_rng(21)
    rnd = lo + (O.splitmix64_points(300_000, seed=9) + 0.5) * (hi - lo)
    cell = np.clip(np.floor((rnd - lo) / (hi - lo) * 16.0), 0, 15).astype(np.int64)
    key = cell[:, 0] + 16 * (cell[:, 1] + 16 * cell[:, 2])
    srt = rnd[np.argsort(key, kind="stable")]                                   # ~73 points a cell: most waves lie in one cell
    m = 97
    ax = [lo[a] + (np.arange(m) + 0.5) / m * (hi[a] - lo[a]) for a in range(3)]
    grid = np.stack(np.meshgrid(ax[0], ax[1], ax[2], indexing="ij"), -1).reshape(-1, 3)   # z fastest: ~6 points a cell and run
    pairs = np.repeat(rnd[:100_000], 2, axis=0)                                 # runs of exactly 2: 32 runs a wave
    mixed = np.repeat(rnd[100_000:160_000], rng.integers(1, 3, 60_000), axis=0)  # runs of 1 or 2: ~43 runs a wave
    long_runs = np.repeat(rnd[:3000], rng.integers(1, 200, 3000), axis=0)       # runs that straddle waves and tiles
    sets = {"cell-sorted": srt, "grid": grid, "pairs": pairs, "mixed": mixed, "long runs": long_runs, "random": rnd}
    for name, pts in sets.items():
        pts = pts.copy()
        pts[5::997] = lo - 0.25 * (hi - lo)          # outside the root, inside runs
        pts[11::4999, 1] = np.nan
        pts = pts[:len(pts) - 37]                    # a ragged last tile
        got, want = tree.query(pts), ot.query(pts)
        assert np.array_equal(bits(got), bits(want)), name
        init = np.full((len(pts), 3), 7.0)
        gv, gg = tree.query_with_gradient(pts, init)
        wv, wg = ot.query_with_gradient(pts, init)
        assert np.array_equal(bits(gv), bits(wv)) and np.array_equal(bits(gg), bits(wg)), name
    assert (tree.info()["n_nodes"] == 4681) == (root is None)  # (C2: every leaf inline, query_kernel; D1: a refined tree, the any-tree kernel)


def test_query_down_to_max_depth_bitwise(H, O, ctx):
    """TREE_MAX_DEPTH = 10 (Consts.h:8): a chain of splits towards the (+,+,+) corner, leaves of degree 0-5 at every
    depth 1..10 -- the walk below the (here 1-level) top table, cooperative and deferred leaves side by side."""
    from helpers import deep_chain_block
    rng = np.random.default_rng(10)
    blk = deep_chain_block(rng)
    tree, ot = H.DeviceTree(ctx, blk), O.Tree.from_block(blk)
    assert tree.info()["max_depth"] == 10 and tree.info()["max_degree"] == 5
    pts = rng.uniform(-0.5, 0.5, (300000, 3))
    pts[:200000] = 0.5 - rng.uniform(0, 1, (200000, 3)) * 2.0 ** -rng.integers(0, 11, (200000, 1))  # crowd the deep corner
    pts[:64] = 0.5                                                                                  # the corner itself
    got, want = tree.query(pts), ot.query(pts)
    assert np.array_equal(bits(got), bits(want))
    gv, gg = tree.query_with_gradient(pts[:100000])
    wv, wg = ot.query_with_gradient(pts[:100000])
    assert np.array_equal(bits(gv), bits(wv)) and np.array_equal(bits(gg), bits(wg))
    o = rng.uniform(-0.5, 0.5, (4000, 3))
    d = rng.standard_normal((4000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    h1, t1 = tree.query_ray(o, d, 1.0)
    h2, t2 = ot.query_ray(o, d, 1.0)
    assert np.array_equal(h1, h2) and np.array_equal(bits(t1), bits(t2))


def test_capi_argument_checks_query_side(H, ctx):
    import ctypes as C
    L = H.lib()
    blk, _ = H.create_block(ctx, H.make_config(1e-4), H.Field.sphere(), 0)
    tree = H.DeviceTree(ctx, blk)
    assert L.hpsdf_query_ray_host(None, tree.handle, None, None, None, 0, None, None) == H.ERR_NO_DEVICE
    assert L.hpsdf_query_ray_host(ctx.handle, tree.handle, None, None, None, 3, None, None) == 1
    assert L.hpsdf_query_ray_host(ctx.handle, tree.handle, None, None, None, 0, None, None) == 0  # empty batch
    assert L.hpsdf_query_gradient_host(ctx.handle, tree.handle, None, 0, None, None) == 0
    f3 = (C.c_float * 3)(0, 0, 0)
    assert L.hpsdf_function_slice(ctx.handle, tree.handle, 0.0, f3, f3, 0, None, None) == 1
    buf = (C.c_uint8 * 12)()
    assert L.hpsdf_function_slice(ctx.handle, tree.handle, 0.0, f3, f3, 0, buf, None) == 1       # n_samples = 0
    hit, t = tree.query_ray(np.zeros((0, 3)), np.zeros((0, 3)), 1.0)
    assert hit.shape == (0,) and t.shape == (0,)
    rgb, vals = tree.function_slice(0.0, (-0.5,) * 3, (0.5,) * 3, 1)
    assert rgb.shape == (1, 1, 3) and vals.shape == (1, 1)


@pytest.mark.parametrize("tol,max_iter", [(0.0, 0), (1e-12, 0), (0.0, 3), (1e-30, 40)])
def test_continuity_device_solve_equals_host_solve(H, ctx, tol, max_iter):
    """cg.hip runs continuity.cpp's conjugate-gradient loop with the same sums in the same order: same block, same
    iteration count, same residual -- also when the loop ends on the iteration cap instead of the tolerance."""
    cfg0 = H.make_config(1e-8, continuity=False)
    cfg0.continuity_strength = 8.0
    b0, _ = H.create_block(ctx, cfg0, H.Field.sphere((0.25, 0.0, 0.0), 0.5), 1024)
    host, sh = H.continuity_post_process(bytes(b0), tol, max_iter)
    dev, sd = H.continuity_post_process(bytes(b0), tol, max_iter, ctx=ctx)
    assert host == dev
    assert sh["iterations"] == sd["iterations"] and sh["residual"] == sd["residual"] and sh["jump_after"] == sd["jump_after"]
    assert sd["iterations"] > 0 and (max_iter == 0 or sd["iterations"] <= max_iter)


@pytest.mark.parametrize("case", ["sphere@1e-8", "union3@1e-7 K=256", "offset sphere, custom root", "deep chain", "mixed degrees",
                                  "eight leaves"])
def test_continuity_matrix_assembled_on_device_equals_host(H, ctx, case):
    """continuity_asm.hip against continuity.cpp: row pointer, columns and values of the jump-energy matrix bit for bit --
    conforming faces (analytic integrals) and non-conforming ones (quadrature across depth differences of 1 and more),
    own blocks that sum several faces, degrees 0..12."""
    rng = np.random.default_rng(11)
    if case == "sphere@1e-8":
        blk, _ = H.create_block(ctx, H.make_config(1e-8), H.Field.sphere(), 1024)
    elif case == "union3@1e-7 K=256":
        blk, _ = H.create_block(ctx, H.make_config(1e-7), H.Field.union3(), 256)
    elif case == "offset sphere, custom root":
        blk, _ = H.create_block(ctx, H.make_config(3e-8, (-0.25, -0.3, -0.2), (0.6, 0.5, 0.7)), H.Field.sphere((0.25, 0.0, 0.0), 0.3), 1024)
    elif case == "deep chain":
        from helpers import deep_chain_block
        blk = deep_chain_block(rng, max_depth=6)
    elif case == "eight leaves":
        blk = synthetic_block(rng, [4, 1, 0, 6, 2, 2, 9, 3], depth=1)
    else:
        blk = synthetic_block(rng, [12, 0, 7, 3, 11, 2, 5, 9], depth=2)
    rp, col, val, st = H.continuity_matrix(blk, 4)
    drp, dcol, dval, dst = H.continuity_matrix_device(ctx, blk)
    assert np.array_equal(rp, drp) and np.array_equal(col, dcol)
    assert np.array_equal(bits(val), bits(dval))
    for k in ("n_pairs", "n_pairs_analytic", "n_pairs_numeric", "nnz"):
        assert st[k] == dst[k], k
    if case in ("union3@1e-7 K=256", "deep chain"):
        assert st["n_pairs_numeric"] > 0


def test_continuity_device_assembly_equals_host_assembly_end_to_end(H, ctx, monkeypatch):
    """hpsdf_create with continuity: matrix assembled on the device vs HPSDF_CONTINUITY_HOST_ASSEMBLY=1 -- same block."""
    cfg = H.make_config(1e-7, continuity=True)
    a, _ = H.create_block(ctx, cfg, H.Field.union3(), 1024)
    monkeypatch.setenv("HPSDF_CONTINUITY_HOST_ASSEMBLY", "1")
    b, _ = H.create_block(ctx, cfg, H.Field.union3(), 1024)
    assert a == b


def test_mesh_create_with_continuity_config5_shape(H, O, ctx):
    """BASELINE config 5 in miniature: mesh field, root = mesh box, targetError 1e-5, continuity.enforce -- GPU build,
    host post-process.  The result must equal the host post-process of the continuity-free build (bit for bit) and stay
    close to the mesh's own signed distance."""
    verts, tris = icosphere(4, 0.35)
    lo, hi = verts.min(0) - 0.03, verts.max(0) + 0.03
    f = H.Field.mesh(ctx, verts, tris)
    cfg = H.make_config(1e-5, tuple(lo), tuple(hi), continuity=True)
    blk, st = H.create_block(ctx, cfg, f, 1024)
    cs = H.continuity_last_stats()
    assert cs["n_pairs"] >= 11520 and cs["residual"] < 1e-6 and cs["jump_after"] < cs["jump_before"]
    cfg0 = H.make_config(1e-5, tuple(lo), tuple(hi), continuity=False)
    blk0, _ = H.create_block(ctx, cfg0, f, 1024)
    b0 = bytearray(blk0)
    b0[-80 + 16] = 1
    assert H.continuity_post_process(bytes(b0))[0] == blk
    pts = np.random.default_rng(5).uniform(lo, hi, (50000, 3))
    q = H.DeviceTree(ctx, blk).query(pts)
    d = f.eval(ctx, pts)
    # the reference's continuity tests run at 1e-8 with a 1e-2 bar (HPUnitTests.cpp:80-112); this config's 1e-5 is coarser
    assert np.abs(q - d).max() <= 2e-2 and np.median(np.abs(q - d)) <= 1e-3
