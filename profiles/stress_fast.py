"""Stress of the DEFAULT (integer-moment) extraction path against the CPU oracle on randomised sweeps larger and odder than the
unit tests use (python profiles/stress_fast.py on the GPU box): counts and ids identical, geometry within 1e-6, in both
arithmetic modes (tests/helpers.check_fast_and_exact)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "wildcat-slam_amd", "python"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import helpers  # noqa: E402
import pyoracle  # noqa: E402
from wildcat_slam_amd import lib, synth  # noqa: E402


class Oracle:
    extract_surfels = staticmethod(pyoracle.extract_surfels)


ctx = lib.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 77)
n_fast = n_all = 0


def check(name, pts):
    global n_fast, n_all
    info, st = helpers.check_fast_and_exact(ctx, Oracle, pts)
    n_all += 1
    n_fast += bool(info["fast"].get("fast_path"))
    print(f"{name:56s} {len(pts):8d} pts -> {info['exact']['n']:6d} surfels  fast path: {info['fast'].get('fast_path')}  planes per layer {list(st.nodes_plane)}", flush=True)


for _ in range(8):
    n = int(rng.integers(50_000, 1_200_000))
    check("room, firing order", synth.g1_room(n, seed=int(rng.integers(1 << 30))))
for _ in range(6):
    roots, m, ppr = int(rng.integers(50, 3000)), int(rng.integers(21, 120)), int(rng.integers(1, 9))
    check(f"lattice {roots} roots x {ppr} patches x {m}", synth.g2_lattice(roots, m=m, patches_per_root=ppr, seed=int(rng.integers(1 << 30)))[0])
for _ in range(6):  # the same lattice seen several times in one sweep (temporal clusters), small offsets between the visits
    roots, m, rev = int(rng.integers(50, 800)), int(rng.integers(21, 60)), int(rng.integers(2, 8))
    gap = float(rng.uniform(0.052, 0.09))
    parts = []
    for r in range(rev):
        a, _ = synth.g2_lattice(roots, m=m, seed=99, t_start=gap * r, duration=0.02)
        a["x"] += np.float32(0.0007 * r)
        parts.append(a)
    check(f"lattice {roots} roots x {rev} visits, {gap:.3f} s apart", synth.concat_points(*parts))
for _ in range(3):  # room + lattice mixed in one sweep (time-interleaved halves)
    a = synth.g1_room(int(rng.integers(100_000, 400_000)), seed=int(rng.integers(1 << 30)))
    b, _ = synth.g2_lattice(int(rng.integers(100, 1500)), m=32, seed=int(rng.integers(1 << 30)), t_start=float(a["time"][-1]) + 0.001, duration=0.2)
    check("room followed by a lattice", synth.concat_points(a, b))
print(f"all {n_all} cases agree with the oracle ({n_fast} completed by the default path itself)")
