"""`python bench.py --gpus N` as the driver types it: the script becomes its own launcher (torch.distributed.run, rendezvous on
127.0.0.1), one rank per GPU, the library's RCCL communicator on the ctx stream, ONE JSON line on stdout.  The box has one GPU, so
the route is taken as a world of one (WC_BENCH_FORCE_DIST=1 sends `--gpus 1` through the same launcher and the same communicator
set-up); sizes are cut down so the run takes seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_bench_launches_its_own_ranks_and_prints_one_json_line(gpu):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WC_BENCH_FORCE_DIST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--roots", "400", "--no-cpu-baseline",
           "--window-scans", "4", "--window-patches", "2000", "--no-clouds", "--in-flight", "1"]
    p = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # stdout carries the JSON line and nothing else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["value"] > 0 and d["higher_is_better"] is True
    assert d["communicator"] is not None, "the distributed route was not taken"
    assert "roofline" in d and d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert d["window"]["lm_iterations"] >= 1 and d["lm_iters_per_s"] > 0
    assert d["odometry_step"]["ms_per_step"] > 0
