"""Ad-hoc stress of the extraction against the CPU oracle on larger and odder sweeps than the unit tests use (run on the GPU
box: python profiles/stress_extract.py), in the EXACT arithmetic (wc_params.exact_sums = 1: every sum in the reference's order; the
default integer-moment path has profiles/stress_fast.py).  Every case must be bit-identical."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wildcat-slam_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
import pyoracle  # noqa: E402
from wildcat_slam_amd import lib, synth  # noqa: E402

ctx = lib.Context(0)
ctx.set_exact_sums(True)
rng = np.random.default_rng(2024)


def check(name, pts):
    s_ref, id_ref, st = pyoracle.extract_surfels(pts)
    for rep in range(2):  # twice: the second call runs on whatever path the first one made sticky
        s, ids = ctx.extract_surfels(pts)
        assert len(s) == len(s_ref), (name, len(s), len(s_ref))
        # surfels with EQUAL time stamps may come in another order than the oracle's (the reference's std::sort on the stamp
        # leaves ties unspecified, Q7): records are matched by their id before they are compared bit for bit
        info = helpers.check_surfels(s, ids, s_ref, id_ref, tol=1e-6, t_tol=1e-5)
        exact = bool(info["bit_exact"])
        assert exact, (name, info)
        same_order = s.tobytes() == s_ref.tobytes() and ids.tobytes() == id_ref.tobytes()
        print(f"{name:48s} {len(pts):8d} pts -> {len(s):6d} surfels  bit-exact={exact} (same order: {same_order})  layer-2 nodes tested={st.nodes_tested[2]}")


for n in (120_000, 300_000, 700_000):
    check(f"room, firing order, {n}", synth.g1_room(n, seed=int(rng.integers(1 << 30))))
for roots, m, ppr in ((1500, 64, 8), (3000, 24, 8), (2500, 40, 3), (900, 200, 8)):
    check(f"lattice {roots} roots x {ppr} patches x {m}", synth.g2_lattice(roots, m=m, patches_per_root=ppr, seed=int(rng.integers(1 << 30)))[0])
# a room sweep re-ordered so that every voxel's points are consecutive in time per beam: long runs, many events
p = synth.g1_room(400_000, seed=5)
order = np.lexsort((p["time"], p["ring"]))
q = p[order].copy()
q["time"] = np.sort(p["time"])
check("room, beam-major order (long runs)", q)
print("all cases agree with the oracle")
