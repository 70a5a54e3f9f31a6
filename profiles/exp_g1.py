# profiling helper: N extractions of one cloud (argv: room|g2 n_points [steps] [soa]) - run under rocprofv3 --kernel-trace --stats
# soa: the points as float32 xyz (stride 12) + float64 time (stride 8), the layout wc_undistort_sweep_packed leaves
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + "/wildcat-slam_amd/python"]
import numpy as np
from wildcat_slam_amd import lib, synth
from wildcat_slam_amd import records as Rec
which, n = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
soa = len(sys.argv) > 4 and sys.argv[4] == "soa"
pts = synth.g1_room(n, seed=synth.SEED + 3) if which == "room" else synth.g2_lattice(n // 256, m=32)[0]
ctx = lib.Context(0)
if soa:
    n = len(pts)
    cap = (3 * n) // 20 + 1
    d_xyz = ctx.to_device(np.stack([pts["x"], pts["y"], pts["z"]], axis=1).astype(np.float32).reshape(-1))
    d_t = ctx.to_device(np.ascontiguousarray(pts["time"], np.float64))
    d_out, d_ids = ctx.alloc(cap * 144), ctx.alloc(cap * 16)
    desc = Rec.Points(d_xyz.ptr, d_t.ptr, 12, 8, n)
    for _ in range(steps):
        ctx.extract_enqueue(desc, d_out, d_ids, cap, float(pts["time"][0]), float(pts["time"][-1]))
        m = ctx.extract_finish()
    print(len(pts), m, ctx.extract_path_info())
else:
    for _ in range(steps):
        s, i = ctx.extract_surfels(pts)
    print(len(pts), len(s), ctx.extract_path_info())
