"""knock-out timing of k_fx_acc (development option debug_skip; results are wrong, only the timing of what is left means something):
python profiles/dev/knock_acc.py [c2|c5|room] [soa]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "wildcat-slam_amd", "python"))
import numpy as np
from wildcat_slam_amd import lib, synth
from wildcat_slam_amd import records as Rec
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
soa = len(sys.argv) > 2 and sys.argv[2] == "soa"
pts = {"c2": lambda: synth.g2_lattice(3906, m=32)[0], "c5": lambda: synth.g2_lattice(39062, m=32)[0], "room": lambda: synth.g1_room(1_000_000, seed=synth.SEED + 3)}[which]()
n = len(pts)
cap = (3 * n) // 20 + 1
for bits in (0, 4, 2, 1, 2 | 64, 16, 32, 8, 64):
    ctx = lib.Context(0)
    ctx.set_dev_option("debug_skip", bits)
    d_out, d_ids = ctx.alloc(cap * 144), ctx.alloc(cap * 16)
    if soa:
        d_xyz = ctx.to_device(np.stack([pts["x"], pts["y"], pts["z"]], axis=1).astype(np.float32).reshape(-1))
        d_t = ctx.to_device(np.ascontiguousarray(pts["time"], np.float64))
        desc = Rec.Points(d_xyz.ptr, d_t.ptr, 12, 8, n)
    else:
        d = ctx.to_device(pts)
        desc = ctx.points_desc(d, n)
    t_lo, t_hi = float(pts["time"][0]), float(pts["time"][-1])
    ctx.extract_profile(True)
    acc = []
    for i in range(24):
        try:
            ctx.extract_enqueue(desc, d_out, d_ids, cap, t_lo, t_hi)
            ctx.extract_finish()
        except Exception as e:
            pass
        if i >= 4:
            acc.append(ctx.extract_stage_ms()["point_sort"])
    print("skip %5d: k_fx_acc %.1f us (min %.1f)" % (bits, 1e3 * float(np.median(acc)), 1e3 * min(acc)))
    ctx.close()
