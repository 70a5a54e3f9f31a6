"""Ad-hoc stress of the matcher against the CPU oracle's kd-tree on windows larger than the unit tests use (run on the GPU
box: python profiles/stress_match.py; the matcher takes the eight-lanes-per-query walk at these sizes, WC_KNN_GROUP=0 python
profiles/stress_match.py runs the same windows through the lane-per-query walk).  k-NN tables and pair lists must be identical."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wildcat-slam_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402
from wildcat_slam_amd import lib, synth  # noqa: E402

ctx = lib.Context(0)
for scans, patches, fixed, seed in ((20, 2500, 2500, 7), (10, 8000, 20000, 8), (5, 20000, 3000, 9), (8, 31248, 62496, 10), (10, 70000, 30000, 11)):
    w = synth.surfel_window(scans, patches, seed=seed, fixed_patches=fixed)
    t0 = time.perf_counter()
    ref_s = pyoracle.match(w["surf"], w["pose"], w["surf"], w["pose"], True)
    ref_f = pyoracle.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False)
    t_cpu = time.perf_counter() - t0
    got_s, idx, d2 = ctx.match(w["surf"], w["pose"], w["surf"], w["pose"], True, want_knn=True)
    got_f = ctx.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False)
    assert np.array_equal(got_s, ref_s) and np.array_equal(got_f, ref_f), (scans, patches, fixed)
    assert (np.diff(d2, axis=1) >= 0).all() and (idx[:, 0] == np.arange(len(idx))).all()
    # both searches side by side (wc_match_pair: helper context + host thread), three times in a row
    from wildcat_slam_amd import records as R
    ns, nf = len(w["surf"]), len(w["fix_surf"])
    d_s, d_p, d_fs, d_fp = ctx.to_device(w["surf"]), ctx.to_device(w["pose"]), ctx.to_device(w["fix_surf"]), ctx.to_device(w["fix_pose"])
    d_b, d_u = ctx.alloc(8 * ns), ctx.alloc(8 * ns)
    for _ in range(3):
        nb, nu = ctx.match_pair_device(d_s, d_p, ns, d_fs, d_fp, nf, d_b, ns, d_u, ns)
        assert np.array_equal(d_b.download(R.PAIR, nb), ref_s) and np.array_equal(d_u.download(R.PAIR, nu), ref_f), (scans, patches, fixed, "pair")
    print(f"{scans:3d} sweeps x {patches:6d} patches, {fixed:6d} fixed: {len(ref_s):7d} + {len(ref_f):7d} pairs identical (oracle {t_cpu:.1f} s)")
print("all windows agree with the oracle")
