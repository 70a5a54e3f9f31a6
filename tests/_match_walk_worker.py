"""Both walks of the matcher's tree (csrc/match_tree.inc: one lane per query / eight lanes per query) pinned by the development
option knn_group (wc_ctx_set_dev_option), each in a process of its own (tests/test_match_gpu.py starts it twice): every instantiated k, same-set and
other-set searches, k-NN tables and pairs against the CPU oracle.  Usage: python _match_walk_worker.py 0|1"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "wildcat-slam_amd", "python"), os.path.join(ROOT, "oracle")]

import numpy as np  # noqa: E402

import pyoracle  # noqa: E402
from wildcat_slam_amd import lib, synth  # noqa: E402

AS = 5.0 * np.pi / 180.0


def feat(s):
    return np.concatenate([s["center"], s["normal"] / AS], 1)


def main():
    group = int(sys.argv[1])
    ctx = lib.Context(0)
    ctx.set_dev_option("knn_group", group)
    w = synth.surfel_window(4, 2500, seed=77, fixed_patches=3000)  # identity-free poses: features through the oracle's own path
    for k in (1, 2, 3, 5, 8, 10, 12, 16):
        prm = pyoracle.default_params()
        prm.knn_k = k
        ctx.set_params(prm)
        got = ctx.match(w["surf"], w["pose"], w["surf"], w["pose"], True)
        assert np.array_equal(got, pyoracle.match(w["surf"], w["pose"], w["surf"], w["pose"], True, prm)), ("same set", k)
        got = ctx.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False)
        assert np.array_equal(got, pyoracle.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False, prm)), ("other set", k)
    ctx.set_params(pyoracle.default_params())
    rng = np.random.default_rng(3)
    from wildcat_slam_amd import records as R

    for nt in (7, 300, 5000, 40000):
        s = np.zeros(nt, R.SURFEL)
        s["center"] = rng.uniform(-8, 8, (nt, 3))
        nr = rng.normal(size=(nt, 3))
        s["normal"] = nr / np.linalg.norm(nr, axis=1, keepdims=True)
        s["t"] = np.sort(rng.uniform(0, 5, nt))
        p = np.zeros(nt, R.POSE)
        p["quat"][:, 0] = 1.0
        _, idx, d2 = ctx.match(s, p, s, p, True, want_knn=True)
        ridx, rd2 = pyoracle.knn6(feat(s), feat(s), 10)
        assert np.array_equal(d2, rd2) and np.array_equal(idx.astype(np.int64), ridx.astype(np.int64)), nt
    print("walk", group, "ok")


if __name__ == "__main__":
    main()
