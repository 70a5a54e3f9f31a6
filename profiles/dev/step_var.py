"""Repetition-to-repetition spread of the odometry step's stages (one box): python profiles/dev/step_var.py [reps] [torch]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "wildcat-slam_amd", "python"))
import numpy as np
from wildcat_slam_amd import lib, synth
from wildcat_slam_amd.step import StepWindow
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
if len(sys.argv) > 2 and sys.argv[2] == "torch":  # (bench.py imports torch: does the runtime it initialises change the host waits?)
    import torch
    torch.cuda.init(); _x = torch.zeros(4, device="cuda:0"); torch.cuda.synchronize()
ctx = lib.Context(0)
for kv in os.environ.get("STEP_VAR_OPTS", "").split(","):  # development options: STEP_VAR_OPTS="match_pair_hold=0,knn_sort=1"
    if kv:
        ctx.set_dev_option(kv.split("=")[0], int(kv.split("=")[1]))
w = synth.g2_scan_sequence(10, 3906, m=32, seed=synth.SEED + 21)
sw = StepWindow(ctx, w, rank=0, world=1)
sw.step(); sw.step()
rows = []
for _ in range(reps):
    T, info, _ = sw.step()
    rows.append([T[k] * 1e3 for k in ("match", "build", "solve", "total")])
a = np.array(rows)
for i, k in enumerate(("match", "build", "solve", "total")):
    print(k, "min %.3f p50 %.3f max %.3f" % (a[:, i].min(), np.median(a[:, i]), a[:, i].max()), " ".join("%.2f" % v for v in a[:, i]))
