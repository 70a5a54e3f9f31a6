"""N > 1 path of the sharded extraction (csrc/route.hip) on CPU: two gloo ranks each hold a
time-contiguous half of one cloud, partition it by the owner of each point's root voxel (wc_route_owner - the library's own
hash, callable without a GPU), exchange the points with the SAME communicator class bench.py uses with RCCL (dist.TorchComm,
here on host buffers), extract the voxels they own with the oracle, all-gather and merge.  The result must be the unsharded
oracle extraction, byte for byte: a root voxel is complete on its owner and its points are still in time order."""
import os
import sys

import numpy as np
import pytest

from wildcat_slam_amd import dist as wdist
from wildcat_slam_amd import lib
from wildcat_slam_amd import records as R
from wildcat_slam_amd import synth


def _cloud():
    a, _ = synth.g2_lattice(60, m=32, seed=5)
    b = synth.g1_room(30_000, seed=6, t_start=float(a["time"][-1]) + 1e-3)  # firing order: its voxels are spread over time
    return synth.concat_points(a, b)


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
    comm = wd.TorchComm(torch, dist, "cpu")
    pts = _cloud()
    lo, cnt = wd.shard_range(len(pts), rank, world)
    mine = pts[lo : lo + cnt]  # this rank's time-contiguous slice
    segs = wd.route_partition_host(mine, O.voxel_keys(mine), world)
    # counts, then the points: the two all-to-alls of wc_extract_surfels_sharded
    send_cnt = np.array([len(s) for s in segs], np.uint64)
    recv_cnt = np.zeros(world, np.uint64)
    comm.alltoallv(send_cnt.ctypes.data, [8] * world, recv_cnt.ctypes.data, [8] * world)
    send = np.ascontiguousarray(synth.concat_points(*segs))  # (np.concatenate would drop the padding of the 48-byte record)
    recv = np.zeros(int(recv_cnt.sum()), R.POINT)
    comm.alltoallv(send.ctypes.data, [int(c) * 48 for c in send_cnt], recv.ctypes.data, [int(c) * 48 for c in recv_cnt])
    assert np.all(np.diff(recv["time"]) >= 0), "segments in source-rank order must be time ordered"
    s, ids, _ = O.extract_surfels(recv)
    # all-gather + merge (wc_gather_surfels)
    n_all = np.zeros(world, np.uint64)
    n_me = np.array([len(s)], np.uint64)
    comm.allgatherv(n_me.ctypes.data, 8, n_all.ctypes.data, [8] * world)
    s, ids = np.ascontiguousarray(s), np.ascontiguousarray(ids)
    all_s, all_i = np.zeros(int(n_all.sum()), R.SURFEL), np.zeros(int(n_all.sum()), R.SURFEL_ID)
    comm.allgatherv(s.ctypes.data, len(s) * 144, all_s.ctypes.data, [int(c) * 144 for c in n_all])
    comm.allgatherv(ids.ctypes.data, len(ids) * 16, all_i.ctypes.data, [int(c) * 16 for c in n_all])
    offs = np.concatenate([[0], np.cumsum(n_all)]).astype(int)
    ms, mi = wd.merge_surfels_host([all_s[offs[r] : offs[r + 1]] for r in range(world)], [all_i[offs[r] : offs[r + 1]] for r in range(world)])
    np.save(os.path.join(out_dir, f"s{rank}.npy"), ms)
    np.save(os.path.join(out_dir, f"i{rank}.npy"), mi)
    np.save(os.path.join(out_dir, f"n{rank}.npy"), np.array([len(recv), len(s)]))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_routed_extraction_two_gloo_ranks_equals_unsharded(tmp_path, oracle):
    import torch.multiprocessing as mp

    world, port = 2, 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    s_ref, i_ref, _ = oracle.extract_surfels(_cloud())
    assert len(s_ref) > 400
    owned = [np.load(tmp_path / f"n{r}.npy") for r in range(world)]
    assert sum(int(o[0]) for o in owned) == len(_cloud()) and all(o[1] > 50 for o in owned)  # both ranks own work
    for r in range(world):
        s, i = np.load(tmp_path / f"s{r}.npy"), np.load(tmp_path / f"i{r}.npy")
        assert i.tobytes() == i_ref.tobytes() and s.tobytes() == s_ref.tobytes()


def test_route_owner_is_a_balanced_function_of_the_voxel():
    rng = np.random.default_rng(3)
    keys = rng.integers(-300, 300, size=(20_000, 3)).astype(np.int32)
    for world in (1, 2, 3, 8):
        o = lib.route_owner(keys, world)
        assert o.min() >= 0 and o.max() < world
        assert np.array_equal(o, lib.route_owner(keys.copy(), world))  # pure function of (kx, ky, kz)
        share = np.bincount(o, minlength=world) / len(o)
        assert np.abs(share - 1 / world).max() < 0.02
    # neighbouring voxels do not all land on one rank (a wall is spread over the ranks)
    line = np.stack([np.arange(64), np.zeros(64, int), np.zeros(64, int)], 1)
    assert len(set(lib.route_owner(line, 8).tolist())) == 8


def test_host_merge_is_the_canonical_order(oracle):
    pts, _ = synth.g2_lattice(40, m=32, seed=9)
    s, i, _ = oracle.extract_surfels(pts)
    parts = [np.arange(len(s)) % 3 == r for r in range(3)]
    ms, mi = wdist.merge_surfels_host([s[p] for p in parts], [i[p] for p in parts])
    assert ms.tobytes() == s.tobytes() and mi.tobytes() == i.tobytes()
