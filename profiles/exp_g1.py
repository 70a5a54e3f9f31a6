# profiling helper: N extractions of one cloud (argv: room|g2 n_points [steps]) - run under rocprofv3 --kernel-trace --stats
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + "/wildcat-slam_amd/python"]
import numpy as np
from wildcat_slam_amd import lib, synth
which, n = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
pts = synth.g1_room(n, seed=synth.SEED + 3) if which == "room" else synth.g2_lattice(n // 256, m=32)[0]
ctx = lib.Context(0)
for _ in range(steps):
    s, i = ctx.extract_surfels(pts)
print(len(pts), len(s), ctx.extract_path_info())
