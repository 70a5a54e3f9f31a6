"""One cloud over several "ranks" (csrc/route.hip; BASELINE config 5) on ONE GPU: the partition
kernel against its host restatement, the whole sharded call with two / three contexts in lock-step (dist.ThreadComm stands
in for RCCL), and the C5-sized cloud (10 M points) at full size."""
import threading

import numpy as np
import pytest

from wildcat_slam_amd import dist as wdist
from wildcat_slam_amd import lib
from wildcat_slam_amd import records as R
from wildcat_slam_amd import synth

pytestmark = pytest.mark.gpu


def _mixed_cloud(n_roots=300, n_room=120_000, seed=5):
    a, _ = synth.g2_lattice(n_roots, m=32, seed=seed)
    b = synth.g1_room(n_room, seed=seed + 1, t_start=float(a["time"][-1]) + 1e-3)
    return synth.concat_points(a, b)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_route_partition_is_stable_and_complete(gpu, oracle, world):
    pts = _mixed_cloud()
    d_pts = gpu.to_device(pts)
    d_send, counts = gpu.route_partition(d_pts, len(pts), world)
    got = d_send.download(R.ROUTE_POINT, len(pts))
    segs = wdist.route_partition_host(pts, oracle.voxel_keys(pts), world)  # library hash + oracle voxel index
    assert [int(c) for c in counts] == [len(s) for s in segs] and int(counts.sum()) == len(pts)
    o = 0
    for s in segs:
        g = got[o : o + len(s)]
        for f, h in (("x", "x"), ("y", "y"), ("z", "z"), ("t", "time")):
            assert np.array_equal(g[f], s[h]), f
        o += len(s)


def _run_ranks(world, pts, want_gather=True, exact=False):
    ctxs = [lib.Context(0) for _ in range(world)]
    for c in ctxs:
        c.set_exact_sums(exact)
    shared = wdist.ThreadComm.shared(world)
    t_lo, t_hi = float(pts["time"][0]), float(pts["time"][-1])
    out, errors = [None] * world, []

    def run(r):
        try:
            ctx = ctxs[r]
            ctx.set_comm(wdist.ThreadComm(shared, r, ctx))
            lo, cnt = wdist.shard_range(len(pts), r, world)
            d_slice = ctx.to_device(pts[lo : lo + cnt])
            d_s, d_i, m, owned = ctx.extract_surfels_sharded(d_slice, cnt, t_lo, t_hi)
            local = (d_s.download(R.SURFEL, m), d_i.download(R.SURFEL_ID, m), owned, ctx.extract_path_info()["fast"])
            merged = ctx.gather_surfels(d_s, d_i, m, cap=len(pts) // 4 + 1024) if want_gather else None
            out[r] = (local, merged)
        except Exception as e:  # pragma: no cover
            errors.append(e)
            shared["bar"].abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errors, errors
    for c in ctxs:
        c.close()
    return out


@pytest.mark.parametrize("world,exact", [(2, True), (3, True), (2, False), (3, False)])
def test_sharded_extraction_equals_unsharded(gpu, world, exact):
    """every rank extracts the voxels it owns from the routed points; the lists are disjoint by voxel, each rank owns work, and
    the gathered + merged list is the unsharded call's output byte for byte (surfels AND ids, in order) - in the exact
    arithmetic always, in the default (integer-moment) arithmetic whenever no rank had to hand its share to the exact path
    (then the two arithmetics meet in one list: ids / counts still identical, geometry within 1e-6)"""
    import helpers

    pts = _mixed_cloud()
    gpu.set_exact_sums(exact)
    try:
        s_ref, i_ref = gpu.extract_surfels(pts)
        ref_fast = gpu.extract_path_info()["fast"]
    finally:
        gpu.set_exact_sums(False)
    assert len(s_ref) > 2000
    out = _run_ranks(world, pts, exact=exact)
    owned_pts = sum(o[0][2] for o in out)
    assert owned_pts == len(pts)
    keys = [set(map(tuple, np.stack([o[0][1]["kx"], o[0][1]["ky"], o[0][1]["kz"]], 1).tolist())) for o in out]
    for a in range(world):
        assert len(out[a][0][0]) > len(s_ref) // (4 * world)
        assert np.all(np.diff(out[a][0][0]["t"]) >= 0)
        own = lib.route_owner(np.array(sorted(keys[a])), world)
        assert np.all(own == a)  # a rank only emits surfels of voxels it owns
        for b in range(a + 1, world):
            assert not (keys[a] & keys[b])
    assert sum(len(o[0][0]) for o in out) == len(s_ref)
    same_arithmetic = exact or (ref_fast and all(o[0][3] for o in out))
    for r in range(world):
        ms, mi = out[r][1]
        if same_arithmetic:
            assert mi.tobytes() == i_ref.tobytes() and ms.tobytes() == s_ref.tobytes()
        else:
            helpers.check_surfels(ms, mi, s_ref, i_ref, tol=1e-6, t_tol=1e-4)
        assert ms.tobytes() == out[0][1][0].tobytes()  # every rank holds the same merged list


def test_merge_surfels_kway(gpu):
    pts = _mixed_cloud(200, 60_000, seed=11)
    s, i = gpu.extract_surfels(pts)
    rng = np.random.default_rng(0)
    lab = rng.integers(0, 5, len(s))
    lists, ids = [s[lab == r] for r in range(5)], [i[lab == r] for r in range(5)]
    lists.insert(2, s[:0]), ids.insert(2, i[:0])  # an empty list in the middle
    ms, mi = gpu.merge_surfels(lists, ids)
    assert ms.tobytes() == s.tobytes() and mi.tobytes() == i.tobytes()


def test_c5_cloud_10m_points_full_size(gpu, oracle):
    """BASELINE config 5's cloud (G2, R = 39 062 roots, 9 999 872 points) at FULL size on one GPU: size-independent properties
    (8 surfels per root, time sorted, every root present) and, for every 32nd root, byte equality with the oracle run on just
    those roots' points (a root voxel's surfels depend on its own points only)."""
    n_roots = 39_062
    pts, info = synth.g2_lattice(n_roots, m=32, seed=synth.SEED + 50)
    assert len(pts) == 9_999_872
    s, ids = gpu.extract_surfels(pts)
    assert gpu.extract_path_info()["fast"]  # the default (integer-moment) path handles the 10 M-point cloud itself
    assert len(s) == 8 * n_roots
    assert np.all(np.diff(s["t"]) >= 0)
    assert np.all(s["resolution"] == np.float64(np.float32(0.2) * 2))  # all at layer 1 (0.4 m)
    got_roots = np.unique(np.stack([ids["kx"], ids["ky"], ids["kz"]], 1), axis=0)
    assert len(got_roots) == n_roots
    # 1/32 sub-sample of the roots: their points are the blocks [r * 256, (r + 1) * 256) of the root-major cloud
    pick = np.arange(0, n_roots, 32)
    sel = (pick[:, None] * 256 + np.arange(256)[None, :]).reshape(-1)
    s_ref, i_ref, _ = oracle.extract_surfels(pts[sel])
    assert len(s_ref) == 8 * len(pick)
    want = set(map(tuple, info["root_keys"][pick].tolist()))
    import helpers

    def sub(ids_):
        return np.array([(a, b, c) in want for a, b, c in zip(ids_["kx"].tolist(), ids_["ky"].tolist(), ids_["kz"].tolist())])

    mask = sub(ids)
    helpers.check_surfels(s[mask], ids[mask], s_ref, i_ref, tol=1e-6, t_tol=1e-4)  # fast path: ids / counts exact, geometry 1e-6
    gpu.set_exact_sums(True)
    try:
        s, ids = gpu.extract_surfels(pts)
    finally:
        gpu.set_exact_sums(False)
    mask = sub(ids)
    assert len(s) == 8 * n_roots
    assert ids[mask].tobytes() == i_ref.tobytes() and s[mask].tobytes() == s_ref.tobytes()  # exact path: the oracle's bytes


def test_c5_cloud_routed_two_ranks(gpu):
    """the same kind of cloud at 2 M points through the routed two-rank path: merged result = unsharded result"""
    pts, _ = synth.g2_lattice(7_812, m=32, seed=synth.SEED + 51)
    s_ref, i_ref = gpu.extract_surfels(pts)
    assert len(s_ref) == 8 * 7_812 and gpu.extract_path_info()["fast"]
    out = _run_ranks(2, pts)
    assert all(o[0][3] for o in out)  # both ranks on the default (integer-moment) path
    share = [o[0][2] / len(pts) for o in out]
    assert abs(share[0] - 0.5) < 0.05  # balanced ownership
    for r in range(2):
        ms, mi = out[r][1]
        assert mi.tobytes() == i_ref.tobytes() and ms.tobytes() == s_ref.tobytes()


def test_c5_cloud_routed_full_size_four_ranks(gpu):
    """BASELINE config 5 at FULL size through the routed path: the 9 999 872-point cloud as four time-contiguous slices on four ranks
    (threads + contexts on the one GPU; dist.ThreadComm stands in for RCCL), ONE all-to-all of 24-byte records by root voxel, every
    rank extracts its voxels (above 2 M points per rank: the two-kernel node stage), wc_gather_surfels merges: every rank ends with
    the unsharded call's 312 496 surfels, byte for byte."""
    n_roots = 39_062
    pts, _ = synth.g2_lattice(n_roots, m=32, seed=synth.SEED + 50)
    s_ref, i_ref = gpu.extract_surfels(pts)
    assert len(pts) == 9_999_872 and len(s_ref) == 8 * n_roots and gpu.extract_path_info()["fast"]
    out = _run_ranks(4, pts)
    assert all(o[0][3] for o in out)  # every rank on the default (integer-moment) path
    share = [o[0][2] / len(pts) for o in out]
    assert abs(sum(share) - 1.0) < 1e-12 and max(share) < 0.27  # every point owned once, balanced ownership
    for r in range(4):
        ms, mi = out[r][1]
        assert mi.tobytes() == i_ref.tobytes() and ms.tobytes() == s_ref.tobytes()


def test_in_library_rccl_communicator_world_of_one(gpu):
    """csrc/comm.hip on the one GPU of this box: librccl.so is dlopen()ed, ncclCommInitRank with one rank, and the sharded
    extraction + gather run their collectives (grouped ncclSend / ncclRecv to self on the ctx stream) through it - the result is
    the plain call's, byte for byte.  (Two ranks need two GPUs: the driver's multi-GPU bench covers that leg.)"""
    ctx = lib.Context(0)
    try:
        uid = lib.rccl_unique_id()
        assert len(uid) == 128
        ctx.comm_rccl_init(0, 1, uid)
        pts = _mixed_cloud(150, 60_000, seed=21)
        s_ref, i_ref = gpu.extract_surfels(pts)
        d_pts = ctx.to_device(pts)
        d_s, d_i, m, owned = ctx.extract_surfels_sharded(d_pts, len(pts), float(pts["time"][0]), float(pts["time"][-1]))
        assert owned == len(pts) and m == len(s_ref)
        ms, mi = ctx.gather_surfels(d_s, d_i, m, cap=m + 16)
        assert mi.tobytes() == i_ref.tobytes() and ms.tobytes() == s_ref.tobytes()
        ctx.comm_rccl_destroy()
    finally:
        ctx.close()
