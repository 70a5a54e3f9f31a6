"""the damped solve's diagonal-block kernel on its own against numpy, with its shader clocks (wc_selftest_factor32); round 5 ran two
single-wavefront forms through it (note in csrc/window.hip)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", os.environ.get("AB_LM_TREE", "."), "wildcat-slam_amd", "python"))
import numpy as np
from wildcat_slam_amd import lib
ctx = lib.Context(0)
rng = np.random.default_rng(5)
for trial in range(3):
    b = rng.normal(size=(32, 40))
    a = b @ b.T + (0.5 if trial else 1e-3) * np.eye(32)
    Lr = np.linalg.cholesky(a)
    Xr = np.linalg.inv(Lr)
    for v in (0, 1):
        L, X, clk, ok = ctx.selftest_factor32(a, v, 9)
        eL = np.abs(np.tril(L) - Lr).max() / np.abs(Lr).max()
        eX = np.abs(np.tril(X) - Xr).max() / np.abs(Xr).max()
        print("trial %d variant %d: %6d clocks, ok %s, |L - L_ref| %.1e, |X - X_ref| %.1e, cond %.1e" % (trial, v, clk, ok, eL, eX, np.linalg.cond(a)))
bad = -np.eye(32)
print("not SPD:", [ctx.selftest_factor32(bad, v, 1)[3] for v in (0, 1)])
