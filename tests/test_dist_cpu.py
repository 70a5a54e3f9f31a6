"""N > 1 path of the window solve on CPU: two gloo ranks shard the correspondences and the IMU factors, linearise their
shard with the oracle, all-reduce the packed {upper block pairs of H, g, cost} buffer (the layout of the device path and
the shard helpers bench.py uses with RCCL) and must end up with the normal equations of the unsharded problem."""
import os
import sys

import numpy as np
import pytest

from wildcat_slam_amd import dist as wdist


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(here, "..", "wildcat-slam_amd", "python"), os.path.join(here, "..", "oracle")):
        sys.path.insert(0, os.path.abspath(p))
    import pyoracle as O
    from wildcat_slam_amd import dist as wd
    from wildcat_slam_amd import synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = synth.surfel_window(3, 200, seed=21, fixed_patches=80)  # replicated, deterministic
    pairs = O.match(w["surf"], w["pose"], w["surf"], w["pose"], True)
    pf = O.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False)
    lo_b, n_b = wd.shard_range(len(pairs), rank, world)
    lo_u, n_u = wd.shard_range(len(pf), rank, world)
    W = O.Window(w["sample_times"], w["grav"], True)
    W.add_binary(w["surf"], w["pose"], pairs[lo_b : lo_b + n_b])
    W.add_unary(w["fix_surf"], w["fix_pose"], w["surf"], w["pose"], pf[lo_u : lo_u + n_u])
    imu_r = wd.shard_imu(w["imu"], rank, world)  # every IMU factor (triple of consecutive states) on exactly one rank
    if len(imu_r) >= 3:
        W.add_imu(imu_r)
    x = 1e-3 * np.random.default_rng(5).normal(size=12 * W.ns)
    H, g, cost = W.linearize(x)
    buf = torch.from_numpy(wd.pack(H, g, cost))
    assert buf.numel() == wd.packed_count(W.ns)
    dist.all_reduce(buf)  # the ONE collective of a linearisation
    c = torch.tensor([W.evaluate(x)], dtype=torch.float64)
    dist.all_reduce(c)  # candidate-cost evaluation: one scalar
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.concatenate([buf.numpy(), c.numpy()]))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_linearization_allreduce_gloo(tmp_path, oracle):
    import torch.multiprocessing as mp

    from wildcat_slam_amd import synth

    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(r0, r1)  # every rank holds bitwise the same reduced system => LM stays in lock-step
    w = synth.surfel_window(3, 200, seed=21, fixed_patches=80)
    pairs = oracle.match(w["surf"], w["pose"], w["surf"], w["pose"], True)
    pf = oracle.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False)
    W = oracle.Window(w["sample_times"], w["grav"], True)
    W.add_binary(w["surf"], w["pose"], pairs)
    W.add_unary(w["fix_surf"], w["fix_pose"], w["surf"], w["pose"], pf)
    W.add_imu(w["imu"])
    x = 1e-3 * np.random.default_rng(5).normal(size=12 * W.ns)
    H, g, cost = W.linearize(x)
    Hr, gr, cr = wdist.unpack(r0[:-1], W.ns)
    assert np.abs(Hr - H).max() <= 1e-12 * np.abs(H).max()
    assert np.abs(gr - g).max() <= 1e-12 * np.abs(g).max()
    assert abs(cr - cost) <= 1e-12 * cost and abs(r0[-1] - W.evaluate(x)) <= 1e-12 * cost


def _worker_two(rank, world, port, out_dir):
    """the two-collective form (round 6): surfel factors sharded, IMU factors on every rank; {surfel cost} first, then the pose
    corners + the pose half of g; every rank adds the sums onto its own IMU part"""
    import torch
    import torch.distributed as dist

    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(here, "..", "wildcat-slam_amd", "python"), os.path.join(here, "..", "oracle")):
        sys.path.insert(0, os.path.abspath(p))
    import pyoracle as O
    from wildcat_slam_amd import dist as wd
    from wildcat_slam_amd import synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = synth.surfel_window(3, 200, seed=21, fixed_patches=80)
    pairs = O.match(w["surf"], w["pose"], w["surf"], w["pose"], True)
    pf = O.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False)
    lo_b, n_b = wd.shard_range(len(pairs), rank, world)
    lo_u, n_u = wd.shard_range(len(pf), rank, world)
    Ws = O.Window(w["sample_times"], w["grav"], True)  # this rank's surfel factors
    Ws.add_binary(w["surf"], w["pose"], pairs[lo_b : lo_b + n_b])
    Ws.add_unary(w["fix_surf"], w["fix_pose"], w["surf"], w["pose"], pf[lo_u : lo_u + n_u])
    Wi = O.Window(w["sample_times"], w["grav"], True)  # ALL IMU factors, on every rank
    Wi.add_imu(w["imu"])
    x = 1e-3 * np.random.default_rng(5).normal(size=12 * Ws.ns)
    Hs, gs, cs = Ws.linearize(x)
    Hi, gi, ci = Wi.linearize(x)
    small = torch.tensor([cs, 0.0], dtype=torch.float64)
    dist.all_reduce(small)  # collective 1: 16 bytes - the trust-region decision waits for this one only
    cost = ci + float(small[0])
    buf = torch.from_numpy(wd.pack_corners(Hs, gs))
    assert buf.numel() + 2 == wd.corner_count(Ws.ns)
    dist.all_reduce(buf)  # collective 2: pose corners + the pose half of g (needed by the reduced system only)
    H, g = wd.add_corners(buf.numpy(), Hi, gi)
    np.save(os.path.join(out_dir, f"two{rank}.npy"), np.concatenate([H.reshape(-1), g, [cost]]))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_collective_linearization_gloo(tmp_path, oracle):
    import torch.multiprocessing as mp

    from wildcat_slam_amd import synth

    world, port = 2, 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_two, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "two0.npy"), np.load(tmp_path / "two1.npy")
    assert np.array_equal(r0, r1)  # the IMU part is formed identically on every rank, the rest is a sum all ranks receive
    w = synth.surfel_window(3, 200, seed=21, fixed_patches=80)
    pairs = oracle.match(w["surf"], w["pose"], w["surf"], w["pose"], True)
    pf = oracle.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False)
    W = oracle.Window(w["sample_times"], w["grav"], True)
    W.add_binary(w["surf"], w["pose"], pairs)
    W.add_unary(w["fix_surf"], w["fix_pose"], w["surf"], w["pose"], pf)
    W.add_imu(w["imu"])
    x = 1e-3 * np.random.default_rng(5).normal(size=12 * W.ns)
    H, g, cost = W.linearize(x)
    n = 12 * W.ns
    Hr, gr, cr = r0[: n * n].reshape(n, n), r0[n * n : n * n + n], r0[-1]
    assert np.abs(Hr - H).max() <= 1e-12 * np.abs(H).max()
    assert np.abs(gr - g).max() <= 1e-12 * np.abs(g).max()
    assert abs(cr - cost) <= 1e-12 * cost
    # the payload: 36 doubles per pair for ALL pairs instead of 144 for the near ones
    assert wdist.corner_count(W.ns) < wdist.packed_count(W.ns)


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 1000, 999_937):
        for world in (1, 2, 3, 8):
            spans = [wdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (lo, c), (lo2, _) in zip(spans, spans[1:]):
                assert lo + c == lo2
