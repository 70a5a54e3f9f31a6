"""GPU parity of wc_extract_surfels against the CPU oracle (BuildSurfels, surfel_extraction.cc:316-337).
Bar (north_star): voxel indices and surfel counts bit-exact, normals within 1e-6 relative."""
import ctypes as C

import numpy as np
import pytest

import helpers
from wildcat_slam_amd import records as R
from wildcat_slam_amd import synth

pytestmark = pytest.mark.gpu


def _run(gpu, oracle, pts, expect_fast=None, **kw):
    """both arithmetic modes of the library against the oracle (helpers.check_fast_and_exact)"""
    info, st = helpers.check_fast_and_exact(gpu, oracle, pts, expect_fast=expect_fast, **kw)
    res = dict(info["exact"])
    res["fast"] = info["fast"]
    return res, st


def test_voxel_keys_bit_exact(gpu, oracle):
    pts = synth.g1_room(50_000)
    assert np.array_equal(gpu.voxel_keys(pts), oracle.voxel_keys(pts))
    pts, _ = synth.g2_lattice(300, m=32)
    assert np.array_equal(gpu.voxel_keys(pts), oracle.voxel_keys(pts))


def test_g2_lattice_small(gpu, oracle):
    pts, info = synth.g2_lattice(400, m=32)
    res, st = _run(gpu, oracle, pts, expect_fast=True)
    assert res["n"] == 8 * 400
    print(res)


def test_g2_sparse_overlap_q4(gpu, oracle):
    # one patch per root: the root is a plane AND is force-split, so root (0.8 m) and child (0.4 m) surfels overlap (Q4)
    pts, info = synth.g2_lattice(500, m=48, patches_per_root=1)
    res, st = _run(gpu, oracle, pts, expect_fast=True)
    assert st.nodes_plane[0] > 0 and st.nodes_plane[1] > 0
    print(res)


def test_g1_room_multires(gpu, oracle):
    pts = synth.g1_room(300_000)
    res, st = _run(gpu, oracle, pts, expect_fast=True)
    assert res["n"] > 1000
    print(res, list(st.nodes_plane))


def test_sparse_firing_order_sweep_stays_on_the_default_path(gpu, oracle):
    # what the host facade hands over every 0.5 s: ~75 k points of a room in firing order, nearly every point of a 1024-point tile
    # in a cell of its own.  Three quarters of the tile's partial sums overflow its 256-cell LDS hash into the spill pool; round 2
    # sized that pool at n / 32 per bank, so every such sweep silently fell back to the exact path (and the facade never ran the
    # kernels the bench measures).  The sweep must be completed by the default path on a FRESH context, first call.
    from wildcat_slam_amd import lib

    msgs, _, _ = synth.raw_stream(1.0, pts_per_s=150_000, t_start=1000.0)
    ctx = lib.Context(0)
    try:
        for k in (0, 5):
            pts = synth.concat_points(*msgs[k : k + 5])
            res, st = helpers.check_fast_and_exact(ctx, oracle, pts, expect_fast=True)
            assert res["fast"]["n"] > 300 and ctx.extract_path_info()["fallbacks"] == 0
    finally:
        ctx.close()


def test_g1_firing_order_full_size(gpu, oracle):
    # the bench's second extraction entry: a 1 M-point sweep in firing order (five revolutions of the room: every node holds
    # several temporal clusters, the busiest voxels have long record lists, octree layer 2 is in use) - counts and ids equal
    # to the oracle's, geometry within 1e-6, completed by the default (integer-moment) path itself
    pts = synth.g1_room(1_000_000, seed=synth.SEED + 3)
    res, st = _run(gpu, oracle, pts, expect_fast=True)
    assert res["n"] > 10_000 and st.nodes_plane[2] > 0
    print(res, list(st.nodes_plane))


def test_long_record_lists_are_merged_from_the_second_sweep_on(gpu, oracle):
    # sweeps in firing order leave several records per (node, time slot) list; a sweep that met such lists makes the next one run
    # k_fx_merge (one lane per list) in front of k_fx_nodes: every sweep of the series must equal the oracle, the flag must be up
    # after the first one, and a run-structured sweep must take it down again
    gpu.set_exact_sums(False)
    room = synth.g1_room(500_000, seed=31)
    s_ref, i_ref, st = oracle.extract_surfels(room)
    for k in range(3):
        s, i = gpu.extract_surfels(room)
        info = gpu.extract_path_info()
        assert info["fast"] and info["long_lists"], (k, info)
        helpers.check_surfels(s, i, s_ref, i_ref, tol=1e-6, t_tol=1e-4)
    lat, _ = synth.g2_lattice(600, m=32)
    l_ref, li_ref, _ = oracle.extract_surfels(lat)
    for k in range(2):
        s, i = gpu.extract_surfels(lat)
        helpers.check_surfels(s, i, l_ref, li_ref, tol=1e-6, t_tol=1e-4)
    assert not gpu.extract_path_info()["long_lists"]


def test_g1_no_time_hint(gpu, oracle):
    pts = synth.g1_room(100_000, seed=7)
    _run(gpu, oracle, pts, hint=False)


def test_small_timestamps_and_temporal_clusters(gpu, oracle):
    # re-observe the same lattice twice, 0.2 s apart, inside one sweep: every node holds two temporal clusters
    a, _ = synth.g2_lattice(200, m=32, t_start=0.0, duration=0.1)
    b, _ = synth.g2_lattice(200, m=32, t_start=0.3, duration=0.1)
    b["x"] += np.float32(0.001)
    pts = synth.concat_points(a, b)
    res, st = _run(gpu, oracle, pts, expect_fast=True)
    assert res["n"] == 2 * 8 * 200
    assert st.clusters_total == 2 * 8 * 200  # two temporal clusters per layer-1 plane node


@pytest.mark.parametrize("revisits,fast", [(6, True), (9, None)])
def test_many_temporal_clusters_per_node(gpu, oracle, revisits, fast):
    # the same lattice observed 6 / 9 times inside one sweep, 0.08 s apart: every layer-1 node is a plane with that many
    # temporal clusters.  k_fx_nodes puts every closed cluster aside as a job (512 per wavefront of 64 nodes): 6 x 64 fit - the
    # default path completes the sweep itself; 9 x 64 do not - the sweep must be handed to the exact path, same result
    parts = []
    for r in range(revisits):
        a, _ = synth.g2_lattice(150, m=32, t_start=0.08 * r, duration=0.02)
        a["x"] += np.float32(0.0005 * r)
        parts.append(a)
    pts = synth.concat_points(*parts)
    res, st = _run(gpu, oracle, pts, expect_fast=fast)
    assert res["n"] == revisits * 8 * 150 and st.clusters_total == revisits * 8 * 150
    if fast is None:
        assert not res["fast"]["fast_path"]


def test_empty_and_tiny_inputs(gpu, oracle):
    s, i = gpu.extract_surfels(np.zeros(0, R.POINT))
    assert len(s) == 0
    pts, _ = synth.g2_lattice(1, m=5)  # 40 points in one root, no node reaches 21 points at layer 1
    res, st = _run(gpu, oracle, pts)
    pts, _ = synth.g2_lattice(3, m=2)  # 16 points per root: below the root threshold
    s, i = gpu.extract_surfels(pts)
    assert len(s) == 0


def test_wide_extent_falls_back_to_wide_keys(gpu, oracle):
    pts, _ = synth.g2_lattice(64, m=32, span=8)
    far, _ = synth.g2_lattice(64, m=32, span=8, seed=5, t_start=synth.T0 + 0.6)
    far["x"] += np.float32(2000.0)  # > 512 root voxels away from the first point
    pts = synth.concat_points(pts, far)
    _run(gpu, oracle, pts)


def test_capacity_error(gpu):
    from wildcat_slam_amd import lib

    pts, _ = synth.g2_lattice(100, m=32)
    with pytest.raises(lib.WildcatError) as e:
        gpu.extract_surfels(pts, cap=10)
    assert e.value.code == lib.WC_ERR_CAPACITY


def test_c2_full_size_properties(gpu, oracle):
    """BASELINE config C2 (999 936 points): count is known by construction (8 surfels per root), output sorted,
    and a 1/16 sub-sample of roots agrees with the oracle."""
    pts, info = synth.g2_lattice(3906, m=32)
    s_gpu, id_gpu = gpu.extract_surfels(pts)
    assert gpu.extract_path_info()["fast"]  # the headline workload runs on the fast path, no fall-back
    assert len(pts) == 999_936 and len(s_gpu) == 8 * 3906
    assert np.all(np.diff(s_gpu["t"]) >= 0)
    assert np.all(s_gpu["resolution"] == np.float64(np.float32(0.4)))
    res, st = _run(gpu, oracle, pts, expect_fast=True)
    print(res["fast"])


def test_c5_cloud_full_size_on_one_gpu(gpu, oracle):
    """BASELINE config C5's cloud (9 999 872 points) on ONE GPU, both arithmetic modes against the oracle: the sizes at which
    k_fx_nodes hands its parents out sub-list after sub-list (more than 2 M points) and a time bucket of the slot order holds
    more than 64 surfels (k_slot_emit's LDS ranking)."""
    pts, info = synth.g2_lattice(39062, m=32)
    assert len(pts) == 9_999_872
    res, st = _run(gpu, oracle, pts, expect_fast=True)
    assert st.surfels == 8 * 39062
    print(res["fast"])


def test_dense_voxels_overflow_fast_sort_and_fall_back(gpu, oracle):
    # 3 root voxels with ~9 000 points each: a bucket of the fast (bucket) sort overflows and the general radix path
    # must take over transparently; huge roots also exercise the multi-chunk streaming of k_roots
    pts, _ = synth.g2_lattice(3, m=1100, span=4, seed=17)
    assert len(pts) == 3 * 8 * 1100
    _run(gpu, oracle, pts)


def test_layer2_pass_after_a_sweep_without_splits(gpu, oracle):
    # the layer-2 launch is skipped when the previous sweep queued no root for it; a sweep that does need it must then be
    # completed by the finish() side (late layer-2 pass + second ordering), and the next regular sweep must be unaffected
    regular, _ = synth.g2_lattice(300, m=32)
    room = synth.g1_room(300_000)
    _run(gpu, oracle, regular)
    res, st = _run(gpu, oracle, room)
    assert st.nodes_tested[2] > 0  # the room does reach layer 2
    _run(gpu, oracle, regular)
    _run(gpu, oracle, room)


def test_wide_time_hint_overflows_time_bins_and_falls_back(gpu, oracle):
    # a time hint a thousand times wider than the sweep puts every surfel into one of the 4096 time bins: the bin overflows and
    # the call is completed with the radix sort of the slot keys
    pts, _ = synth.g2_lattice(300, m=32)
    t0, t1 = float(pts["time"][0]), float(pts["time"][-1])
    _run(gpu, oracle, pts, hint=(t0 - 1.0, t1 + 2000.0))
    _run(gpu, oracle, pts)


def test_unordered_sweep_uses_radix_path_and_stays_there(gpu, oracle):
    # spinning multi-beam order (consecutive points come from different beams): no run structure, bins overflow; the calls that
    # follow start on the radix-sort path directly and must give the same result
    room = synth.g1_room(400_000, seed=11)
    for _ in range(3):
        _run(gpu, oracle, room)
    regular, _ = synth.g2_lattice(200, m=32)
    for _ in range(18):  # long enough for the fast path to be tried again
        _run(gpu, oracle, regular)


def test_unordered_small_sweep_stays_on_the_run_binned_path(gpu, oracle):
    # firing-order sweep small enough for the run bins (one run per point, buckets of a few hundred runs: in-wave bitonic sort,
    # LDS capacity grown on demand); the first call streams with k_roots (binary search in the run offsets), the following
    # ones with k_roots_banks (run statistics of the previous sweep) - all of them must agree with the oracle
    room = synth.g1_room(60_000, seed=5)
    for _ in range(3):
        _run(gpu, oracle, room)
    regular, _ = synth.g2_lattice(150, m=32)
    for _ in range(2):  # back to a sweep with run structure
        _run(gpu, oracle, regular)
    _run(gpu, oracle, synth.g1_room(120_000, seed=6))


@pytest.mark.parametrize("override", [
    dict(max_layer=1),                                  # no layer-2 pass at all
    dict(max_layer=0),                                  # root voxels only
    dict(min_points=8, cluster_min_points=8),           # denser head table / more candidate slots per point
    dict(cluster_gap=2e-4),                             # many temporal clusters per node (every scan line its own)
    dict(voxel_size=0.5, planer_threshold=0.02),        # other grid, other gate
    dict(voxel_size=0.3),                               # a voxel size that is no short binary fraction (round 5: the moments are taken
    dict(voxel_size=0.95),                              # about the voxel centre itself, exact in fp64 for ANY float voxel size;
    dict(voxel_size=0.125, planer_threshold=0.002),     # 0.95: the largest grid the default arithmetic takes, |p - centre| 2^32 < 2^31)
    dict(view_point=(3.0, -2.0, 1.0), min_plane_likeness=0.3),
])
def test_non_default_parameters(gpu, oracle, override):
    """the reference hard-codes BuildVoxelMap(..., 0.8, 2, {20,...}, 0.01, 0.1) (surfel_extraction.cc:327); the kernels take
    them as parameters, so other values must agree with the oracle as well - on a sweep with run structure and on one in
    firing order"""
    params = oracle.default_params()
    for k, v in override.items():
        if k == "view_point":
            for i in range(3):
                params.view_point[i] = v[i]
        else:
            setattr(params, k, v)
    gpu.set_params(params)
    gpu.params = params
    try:
        for pts in (synth.g2_lattice(120, m=40)[0], synth.g1_room(90_000, seed=13)):
            helpers.check_fast_and_exact(gpu, oracle, pts, params=params)
    finally:
        gpu.params = oracle.default_params()
        gpu.set_params(gpu.params)


@pytest.mark.parametrize("xyz_stride", [12, 16])
def test_soa_point_layouts_match_the_aos_record(gpu, oracle, xyz_stride):
    """wc_points takes any (xyz pointer + stride, time pointer + stride): packed 12-byte xyz / padded 16-byte xyz with a
    separate time array must give bit-for-bit what the 48-byte hilti_ros::Point record gives (common.h:12-28)"""
    for pts in (synth.g2_lattice(300, m=32)[0], synth.g1_room(60_000)):
        for exact in (True, False):
            gpu.set_exact_sums(exact)  # (also resets the adaptive path choice: both calls below start from the same state)
            try:
                n = len(pts)
                s_ref, id_ref = gpu.extract_surfels(pts)
                ref_fast = gpu.extract_path_info()["fast"]
                xyz = np.zeros((n, xyz_stride // 4), np.float32)
                xyz[:, 0], xyz[:, 1], xyz[:, 2] = pts["x"], pts["y"], pts["z"]
                t = np.ascontiguousarray(pts["time"], np.float64)
                d_xyz, d_t = gpu.to_device(xyz), gpu.to_device(t)
                cap = max(1024, (3 * n) // 20 + 1)
                d_out, d_ids = gpu.alloc(cap * 144), gpu.alloc(cap * 16)
                desc = R.Points(d_xyz.ptr, d_t.ptr, xyz_stride, 8, n)
                gpu.set_exact_sums(exact)
                gpu.extract_enqueue(desc, d_out, d_ids, cap, float(t[0]), float(t[-1]))
                m = gpu.extract_finish()
                assert m == len(s_ref) > 0
                s, ids = d_out.download(R.SURFEL, m), d_ids.download(R.SURFEL_ID, m)
                if exact or gpu.extract_path_info()["fast"] == ref_fast:  # same arithmetic on both layouts: same bytes
                    assert s.tobytes() == s_ref.tobytes() and ids.tobytes() == id_ref.tobytes()
                else:
                    helpers.check_surfels(s, ids, s_ref, id_ref, tol=1e-6, t_tol=1e-4)
            finally:
                gpu.set_exact_sums(False)

def test_default_path_backs_off_after_repeated_fall_backs(gpu, oracle):
    """a sweep the default (integer-moment) path cannot finish - here: every node observed again after one second, more than its 16 time bins of 0.05 s -
    is repeated on the exact path; when that keeps happening the library goes to the exact path directly for an exponentially
    growing number of sweeps, and comes back to the default path afterwards.  Results are the oracle's all along."""
    a, _ = synth.g2_lattice(60, m=32, t_start=synth.T0, duration=0.1)
    b, _ = synth.g2_lattice(60, m=32, t_start=synth.T0 + 1.0, duration=0.1)  # the same voxels again, one second later
    b["x"] += np.float32(0.001)
    both = synth.concat_points(a, b)
    s_ref, i_ref, _ = oracle.extract_surfels(both)
    gpu.set_exact_sums(False)  # (also resets the adaptive state)
    fast_flags = []
    for _ in range(8):
        s, i = gpu.extract_surfels(both)
        helpers.check_surfels(s, i, s_ref, i_ref, tol=1e-6, t_tol=1e-4)
        fast_flags.append(gpu.extract_path_info()["fast"])
    info = gpu.extract_path_info()
    assert not any(fast_flags)          # this cloud is never finished by the default path ...
    assert 1 <= info["fallbacks"]       # ... which was tried ...
    regular, _ = synth.g2_lattice(100, m=32)
    gpu.set_exact_sums(False)
    s, i = gpu.extract_surfels(regular)
    assert gpu.extract_path_info()["fast"]  # ... and is back for a sweep it can handle


def test_batched_sweeps_equal_single_sweeps(gpu, oracle):
    """wc_extract_surfels_batch_*: K sweeps through ONE launch chain (every sweep its own tables, the kernels run once over all of
    them) give, sweep by sweep, the bytes K wc_extract_surfels calls give - for sweeps with run structure, a sweep in firing order
    (long record lists, layer 2 in use: its flags join the shared layer-2 launches from the second round on), a short and an EMPTY
    sweep in one batch, twice in a row (the tables are clean again behind a batch)."""
    from wildcat_slam_amd import lib

    sweeps = [synth.g2_lattice(300, m=32, seed=synth.SEED + 40)[0], synth.g1_room(120_000, seed=synth.SEED + 41),
              synth.g2_lattice(64, m=32, seed=synth.SEED + 42)[0], synth.g2_lattice(4, m=32, seed=3)[0][:0],
              synth.g2_lattice(500, m=24, seed=synth.SEED + 43)[0]]
    single = [gpu.extract_surfels(p) if len(p) else (np.zeros(0, R.SURFEL), np.zeros(0, R.SURFEL_ID)) for p in sweeps]
    ctx = lib.Context(0)
    try:
        jobs, keep = [], []
        for p in sweeps:
            n = len(p)
            cap = max(1024, (3 * n) // 20 + 1)
            d_p = ctx.to_device(p) if n else ctx.alloc(48)
            d_o, d_i = ctx.alloc(144 * cap), ctx.alloc(16 * cap)
            keep.append((d_p, d_o, d_i))
            t_lo, t_hi = (float(p["time"][0]), float(p["time"][-1])) if n else (1.0, 0.0)
            jobs.append((ctx.points_desc(d_p, n), d_o, d_i, cap, t_lo, t_hi))
        enq, fin = ctx.extract_batch_prepare(jobs)
        for rnd in range(3):
            enq()
            counts = fin()
            for k, (s_ref, id_ref) in enumerate(single):
                assert counts[k] == len(s_ref), (rnd, k, counts[k], len(s_ref))
                s_b, id_b = keep[k][1].download(R.SURFEL, counts[k]), keep[k][2].download(R.SURFEL_ID, counts[k])
                assert s_b.tobytes() == s_ref.tobytes() and id_b.tobytes() == id_ref.tobytes(), (rnd, k)
        # and against the oracle, for the sweep in firing order
        s_o, id_o, _ = oracle.extract_surfels(sweeps[1])
        helpers.check_surfels(keep[1][1].download(R.SURFEL, counts[1]), keep[1][2].download(R.SURFEL_ID, counts[1]), s_o, id_o, tol=1e-6, t_tol=1e-4)
    finally:
        ctx.close()


def test_development_options_and_warmup(gpu, oracle):
    """wc_ctx_set_dev_option replaces the environment knobs of rounds 2 - 4 (nothing in the release library reads the environment to
    decide what to execute): an unknown name is an argument error; both forms of the default path's node stage (fx_split 0 / 1) give the
    oracle's surfels on the same cloud; wc_ctx_warmup can be called on a used context (and twice)"""
    from wildcat_slam_amd import lib

    with pytest.raises(lib.WildcatError):
        gpu.set_dev_option("no_such_option", 1)
    pts, _ = synth.g2_lattice(300, m=32)
    s_ref, id_ref, _ = oracle.extract_surfels(pts)
    try:
        for form in (0, 1):
            gpu.set_dev_option("fx_split", form)
            s_gpu, id_gpu = gpu.extract_surfels(pts)
            assert gpu.extract_path_info()["fast"]
            helpers.check_surfels(s_gpu, id_gpu, s_ref, id_ref, tol=1e-6, t_tol=1e-5)
    finally:
        gpu.set_dev_option("fx_split", -1)
    for _ in range(2):
        gpu._ck(gpu.lib.wc_ctx_warmup(gpu.h, C.c_size_t(64 << 20)))
    s_gpu, id_gpu = gpu.extract_surfels(pts)
    helpers.check_surfels(s_gpu, id_gpu, s_ref, id_ref, tol=1e-6, t_tol=1e-5)


def test_completion_ticket_and_stream_wait_agree(gpu, oracle):
    """wc_extract_surfels_finish waits for the sweep's completion ticket in the pinned mailbox (one thread behind the sweep's last kernel;
    development option ex_sync = 1: the stream wait of rounds 1 - 4).  Sweeps of different clouds alternate 150 times under each form: a
    count or a flag read before it has arrived would show as the other cloud's count, a record read early as other bytes."""
    clouds = [synth.g2_lattice(300, m=32)[0], synth.g2_lattice(77, m=24)[0], synth.g1_room(60000, seed=5)]
    want = []
    gpu.set_dev_option("ex_sync", 1)
    try:
        for pts in clouds:
            s_ref, id_ref, _ = oracle.extract_surfels(pts)
            s_gpu, id_gpu = gpu.extract_surfels(pts)
            helpers.check_surfels(s_gpu, id_gpu, s_ref, id_ref, tol=1e-6, t_tol=1e-5)
            want.append((len(s_gpu), s_gpu.tobytes(), id_gpu.tobytes()))
        assert len({w[0] for w in want}) == len(clouds)
        for form in (0, 1):
            gpu.set_dev_option("ex_sync", form)
            for rep in range(150):
                i = (rep * 7 + rep // 3) % len(clouds)
                s_gpu, id_gpu = gpu.extract_surfels(clouds[i])
                assert i == 2 or gpu.extract_path_info()["fast"]  # (the lattices are completed by the default path itself)
                assert (len(s_gpu), s_gpu.tobytes(), id_gpu.tobytes()) == want[i], (form, rep, i, len(s_gpu), want[i][0])
    finally:
        gpu.set_dev_option("ex_sync", 0)


def test_displaced_root_keeps_its_layer2_nodes(gpu, oracle):
    """regression (round 5): two root voxels with one home slot in the default path's hash; k_fx_nodes<1> clears the entry of the one
    that queued nothing for layer 2, and the layer-2 pass's look-up of the other - displaced behind it - stopped at the now empty slot:
    its points were skipped and a layer-2 surfel went missing, silently, in two runs of three (which tile inserts first is a race).
    The lattice at voxel size 0.95 has such a pair; every repetition, in both forms of the node stage, must give the oracle's ids."""
    params = oracle.default_params()
    params.voxel_size = 0.95
    gpu.set_params(params)
    gpu.params = params
    try:
        pts = synth.g2_lattice(120, m=40)[0]
        s_ref, id_ref, _ = oracle.extract_surfels(pts, params)
        want = set(helpers.id_tuples(id_ref))
        assert sum(1 for t in want if (t[3] & 3) == 2) >= 40  # layer-2 surfels are in play
        for form in (0, 1):
            gpu.set_dev_option("fx_split", form)
            for _ in range(6):
                s_gpu, id_gpu = gpu.extract_surfels(pts)
                assert gpu.extract_path_info()["fast"]
                assert set(helpers.id_tuples(id_gpu)) == want
    finally:
        gpu.set_dev_option("fx_split", -1)
        gpu.params = oracle.default_params()
        gpu.set_params(gpu.params)


def test_default_moments_against_extended_precision(gpu, oracle):
    """which side of the surfel comparison carries the difference?  The surfels of a G2 sweep at epoch-sized
    stamps and +-20 m coordinates from (a) the default arithmetic (exact integer moments about the voxel centre), (b) exact_sums = 1 (fp64
    sums in the reference's order) and (c) the CPU oracle (the reference's un-centred fp64 sums, surfel_extraction.cc:36-51), each against
    the same moments formed in EXTENDED precision (numpy longdouble, centred two-pass).  The default path must be at least as close to
    that as the reference's own sums are - the 1e-6 (surfels) / 1e-5 (step corrections) between the default path and the oracle is then
    the oracle's distance from the exact moments, not the default path's."""
    pts, _ = synth.g2_lattice(600, m=32, seed=synth.SEED + 77)
    s_def, id_def = gpu.extract_surfels(pts)
    gpu.set_exact_sums(True)
    try:
        s_ex, id_ex = gpu.extract_surfels(pts)
    finally:
        gpu.set_exact_sums(False)
    s_ref, id_ref, _ = oracle.extract_surfels(pts)
    # extended-precision moments of every (root voxel, layer-1 octant) cell (G2: one surfel per cell, all at layer 1)
    vs = np.float64(np.float32(0.8))
    p = np.stack([pts["x"], pts["y"], pts["z"]], 1).astype(np.float64)
    k = np.floor(p / vs).astype(np.int64)
    cen = (k + 0.5) * vs
    octant = (4 * (p[:, 0] > cen[:, 0]) + 2 * (p[:, 1] > cen[:, 1]) + (p[:, 2] > cen[:, 2])).astype(np.int64)
    key = np.concatenate([k, octant[:, None]], 1)
    uniq, inv = np.unique(key, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    pl = p.astype(np.longdouble)
    n = np.bincount(inv).astype(np.longdouble)
    c = np.stack([np.bincount(inv, weights=None, minlength=len(uniq)) * 0] * 3, 1).astype(np.longdouble)
    for a in range(3):
        acc = np.zeros(len(uniq), np.longdouble)
        np.add.at(acc, inv, pl[:, a])
        c[:, a] = acc / n
    d = pl - c[inv]
    cov = np.zeros((len(uniq), 9), np.longdouble)
    for a in range(3):
        for b in range(3):
            acc = np.zeros(len(uniq), np.longdouble)
            np.add.at(acc, inv, d[:, a] * d[:, b])
            cov[:, 3 * a + b] = acc / n
    tl = pts["time"].astype(np.longdouble)
    tacc = np.zeros(len(uniq), np.longdouble)
    np.add.at(tacc, inv, tl - tl[0])
    tmean = tl[0] + tacc / n  # (mean of the stamps, formed about the first stamp in extended precision)
    truth = {(int(u[0]), int(u[1]), int(u[2]), int(u[3])): i for i, u in enumerate(uniq)}

    def errs(s, ids):
        assert len(s) == len(uniq)
        rows = np.array([truth[(int(i["kx"]), int(i["ky"]), int(i["kz"]), int((i["node"] >> 2) & 7))] for i in ids])
        assert np.all((ids["node"] & 3) == 1)  # layer 1
        ec = np.abs(s["center"].astype(np.longdouble) - c[rows]).max() / max(float(np.abs(c).max()), 1.0)
        scale = np.abs(cov[rows]).max(axis=1, keepdims=True)
        ev = (np.abs(s["cov"].astype(np.longdouble) - cov[rows]) / scale).max()
        et = np.abs(s["t"].astype(np.longdouble) - tmean[rows]).max()  # seconds
        return float(ec), float(ev), float(et)

    e_def, e_ex, e_ref = errs(s_def, id_def), errs(s_ex, id_ex), errs(s_ref, id_ref)
    print("\ncentre / covariance / stamp [s] against extended precision: default %.1e / %.1e / %.1e, exact_sums %.1e / %.1e / %.1e, oracle %.1e / %.1e / %.1e"
          % (e_def + e_ex + e_ref))
    # the stamps are where the two arithmetics part: the reference adds epoch-sized doubles one by one (surfel_extraction.cc:36-47: ~n ulp of
    # n x 1.6e9 s), the default path averages integer ticks relative to the sweep's first stamp - the correctly rounded mean
    assert e_def[2] <= 3e-7 and e_def[2] <= e_ref[2]
    assert e_def[0] <= 1e-15 and e_def[1] <= 1e-9  # the integer moments are exact: what is left is the final division and the 2^-44 m^2 grid
    assert e_def[1] <= e_ref[1] * 1.001 and e_def[0] <= e_ref[0] * 1.001 + 1e-16  # never further from the exact moments than the reference's sums
    assert e_ex[1] <= 10 * e_ref[1] + 1e-12  # (the reference's order on the device: the same kind of noise as the oracle's)
