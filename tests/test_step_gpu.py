"""The whole odometry step (hot block of LidarOdometry::AddLidarScan, lidar_odometry.cc:523-566) for the N ranks of a multi-GPU job
against the 1-rank step (every stage sharded, the window state replicated).  Ranks = threads with one context each on
the one GPU of the box, dist.ThreadComm standing in for RCCL; the code path between the collectives is the one `bench.py --gpus N`
runs under torch.distributed.run (wildcat_slam_amd/step.py)."""
import threading

import numpy as np
import pytest

from wildcat_slam_amd import dist as wdist
from wildcat_slam_amd import synth
from wildcat_slam_amd.step import StepWindow

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_n_rank_odometry_step_equals_the_one_rank_step(gpu, world):
    from wildcat_slam_amd import lib

    w = synth.g2_scan_sequence(6, 400, m=32, seed=synth.SEED + 5)
    one = StepWindow(gpu, w)
    _, info1, x1 = one.step()
    assert info1["new_surfels"] == 8 * 400 and info1["binary"] > 3000 and info1["unary"] > 500 and info1["iters"] >= 2
    assert info1["allreduce_bytes"] == 0
    ctxs = [lib.Context(0) for _ in range(world)]
    shared = wdist.ThreadComm.shared(world)
    res, errors = [None] * world, []

    def run(r):
        try:
            c = ctxs[r]
            c.set_comm(wdist.ThreadComm(shared, r, c))
            sw = StepWindow(c, w, rank=r, world=world)
            res[r] = sw.step() + (sw,)
        except Exception as e:  # pragma: no cover
            errors.append(e)
            shared["bar"].abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errors, errors
    assert len(set(shared["calls"])) == 1 and shared["calls"][0] > 8  # every rank made the same collectives
    ns = len(w["sample_times"])
    for r in range(world):
        _, info, x, _ = res[r]
        for key in ("new_surfels", "sld", "fix", "binary", "unary", "iters", "term"):
            assert info[key] == info1[key], (r, key, info[key], info1[key])
        assert info["allreduce_bytes"] == 8 * wdist.packed_count(ns) < 2_000_000
        assert np.array_equal(x, res[0][2]), "ranks diverged"
        # north_star: pose increments within 1e-6 relative of the single-GPU step
        assert np.abs(x - x1).max() <= 1e-6 * max(np.abs(x1).max(), 1e-12), np.abs(x - x1).max()
        assert abs(info["cost"][1] - info1["cost"][1]) <= 1e-9 * info1["cost"][1]
    for c in ctxs:
        c.close()


def test_two_processes_on_one_gpu_run_the_n_rank_step(gpu, tmp_path):
    """process-level plumbing of the multi-GPU step: TWO PROCESSES, each with its own context on GPU 0 and a
    torch.distributed (gloo) communicator - rendezvous on 127.0.0.1, no shared Python state, the collectives through the ctx's
    callbacks (dist.StagedTorchComm) - run wildcat_slam_amd/step.py's 2-rank step; both end bitwise equal, with the counts,
    iterations and termination of the 1-rank step and its corrections within 1e-6."""
    import os
    import socket
    import subprocess
    import sys

    scans, roots = 6, 400
    w = synth.g2_scan_sequence(scans, roots, m=32, seed=synth.SEED + 5)
    _, info1, x1 = StepWindow(gpu, w).step()
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    out = str(tmp_path / "step")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_step_worker.py")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, worker, out, str(scans), str(roots)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:  # pragma: no cover
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace")[-2000:])
    assert all(p.returncode == 0 for p in procs), logs
    res = [np.load(out + ".rank%d.npz" % r) for r in range(2)]
    assert np.array_equal(res[0]["x"], res[1]["x"]), "ranks diverged"
    for r in range(2):
        for key in ("new_surfels", "sld", "fix", "binary", "unary", "iters", "term"):
            assert int(res[r][key]) == info1[key], (r, key, res[r][key], info1[key])
        assert int(res[r]["allreduce_bytes"]) == 8 * wdist.packed_count(len(w["sample_times"]))
        assert np.abs(res[r]["x"] - x1).max() <= 1e-6 * max(np.abs(x1).max(), 1e-12)
