"""Randomised differential test of the extraction against the CPU oracle: point clouds the hand-made generators do not
produce - random planes visited in random order, repeated time stamps, bursts separated by gaps around the cluster threshold,
one voxel holding most of the sweep, points exactly on voxel and octant boundaries, very small inputs."""
import os

import numpy as np
import pytest

import helpers
from wildcat_slam_amd import synth

pytestmark = pytest.mark.gpu


def _cloud(seed):
    rng = np.random.default_rng(seed)
    kind = seed % 6
    n = int(rng.integers(30, 60_000))
    if kind == 5:
        n = int(rng.integers(1, 200))
    nplanes = int(rng.integers(1, 40))
    centre = rng.uniform(-20, 20, (nplanes, 3))
    nrm = rng.normal(size=(nplanes, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    which = rng.integers(0, nplanes, n)
    if kind == 1:  # one plane (and with it a few voxels) gets 80 % of the points
        which = np.where(rng.random(n) < 0.8, 0, which)
    a = rng.normal(size=(n, 3)) * rng.choice([0.15, 0.6, 2.5])
    a -= (a * nrm[which]).sum(1, keepdims=True) * nrm[which]  # in-plane offsets
    xyz = centre[which] + a + nrm[which] * rng.normal(size=(n, 1)) * rng.choice([0.0, 0.004, 0.03])
    if kind == 2:  # snap a third of the points onto the voxel / octant lattice (strict > comparisons, floor at boundaries)
        sel = rng.random(n) < 0.33
        xyz[sel] = np.round(xyz[sel] / 0.2) * 0.2
    if kind == 3:  # visit order: sorted by plane in blocks (run structure) instead of random
        order = np.argsort(which + rng.integers(0, 3, n) * nplanes, kind="stable")
        xyz = xyz[order]
    # time stamps: non-decreasing, with repeats and with gaps around the 0.05 s cluster threshold
    dt = rng.choice([0.0, 1e-6, 2e-5, 3e-4], n, p=[0.1, 0.5, 0.3, 0.1])
    gaps = rng.random(n) < (8.0 / max(n, 8))
    dt[gaps] = rng.choice([0.049, 0.0500001, 0.06, 0.2], gaps.sum())
    t = synth.T0 + np.cumsum(dt)
    return synth.make_points(xyz.astype(np.float32), t)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("WC_FUZZ_SEEDS", "24")))))  # WC_FUZZ_SEEDS=N for a longer run
def test_random_clouds_match_oracle(gpu, oracle, seed):
    pts = _cloud(1000 + seed)
    for hint in (True, False):  # both arithmetic modes each: exact = the oracle's bytes and order, fast = ids / counts + 1e-6
        info, _ = helpers.check_fast_and_exact(gpu, oracle, pts, hint=hint)
    print(seed, "fast path" if info["fast"].get("fast_path") else "fell back to the exact path", info["fast"].get("n"))
