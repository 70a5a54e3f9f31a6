import sys, os
sys.path[:0] = ["wildcat-slam_amd/python", "oracle", "tests"]
import numpy as np
from wildcat_slam_amd import lib, synth
msgs, imu, _ = synth.raw_stream(0.6, pts_per_s=150_000, gyro_bias=(0.0, 0.0, 0.02), t_start=1000.0)
ctx = lib.Context(0)
pts = np.concatenate(msgs[0:5])
n = int(sys.argv[1]) if len(sys.argv) > 1 else len(pts)
pts = pts[:n]
print("n", len(pts), "extent", [(float(pts[a].min()), float(pts[a].max())) for a in "xyz"], "t", pts["time"][0], pts["time"][-1])
s, ids = ctx.extract_surfels(pts)
print(len(s), ctx.extract_path_info())
