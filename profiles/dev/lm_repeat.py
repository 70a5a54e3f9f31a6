"""The odometry step repeated from the same window state: iterations, costs and the solve's x must be the same bits in every repetition
(a race in the linearisation's mailbox - k_gather's gather_post - would show as a changed iteration count or a changed x).
python profiles/dev/lm_repeat.py [reps]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "wildcat-slam_amd", "python"))
import zlib
import numpy as np
from wildcat_slam_amd import lib, synth
from wildcat_slam_amd.step import StepWindow
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = lib.Context(0)
bad = 0
for name, w in (("step", synth.g2_scan_sequence(10, 3906, m=32, seed=synth.SEED + 21)), ("small", synth.g2_scan_sequence(6, 400, m=32, seed=synth.SEED + 5))):
    sw = StepWindow(ctx, w, rank=0, world=1)
    _, info0, x0 = sw.step()
    for r in range(reps):
        _, info, x = sw.step()
        if info["iters"] != info0["iters"] or info["cost"] != info0["cost"] or not np.array_equal(x, x0):
            bad += 1
            print(name, "repetition", r, "differs:", info["iters"], info["cost"], float(np.abs(x - x0).max()))
    print(name, "iters", info0["iters"], "cost", info0["cost"], "reps", reps, "differing", bad, "crc32 of x %08x" % zlib.crc32(x0.tobytes()))
print("LM_REPEAT_OK" if bad == 0 else "LM_REPEAT_BAD")
