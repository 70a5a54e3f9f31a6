import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "wildcat-slam_amd/python"); sys.path.insert(0, "oracle")
import numpy as np
import pyoracle
from wildcat_slam_amd import lib
import test_window_gpu as T
gpu = lib.Context(0)
class O:  # the tests' oracle fixture is the module
    pass
for n_scans in (30, 50):
    w, W, keep = T._setup(gpu, pyoracle, n_scans=n_scans, patches=60, fixed=30, seed=3)
    x = np.zeros(12 * W.ns)
    H, g, c = gpu.window_linearize(x)
    Ho, go, co = W.linearize(x)
    print("ns", W.ns, "rel H", T._rel(H, Ho), "rel g", T._rel(g, go), "cost", abs(c - co) / co)
    xs, s, _ = gpu.window_solve(x)
    xr, sr, _ = W.solve(x)
    print("   solve iters", s.iterations, sr.iterations, "term", s.termination, sr.termination, "rel x", T._rel(xs, xr))
