"""extraction: wc_extract_surfels_finish waiting for the sweep's completion ticket (default) against the stream wait (development option
ex_sync = 1), alternating on one context: wall ms per sweep (enqueue + finish).  python profiles/dev/ab_ticket.py [steps]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "wildcat-slam_amd", "python"))
import numpy as np
from wildcat_slam_amd import lib, synth
from wildcat_slam_amd import records as Rec
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ctx = lib.Context(0)
c2 = synth.g2_lattice(3906, m=32)[0]
room = synth.g1_room(1_000_000, seed=synth.SEED + 3)
for name, pts, soa in (("c2", c2, False), ("c2 packed", c2, True), ("room", room, False)):
    n = len(pts)
    cap = (3 * n) // 20 + 1
    d_out, d_ids = ctx.alloc(cap * 144), ctx.alloc(cap * 16)
    if soa:
        d_xyz = ctx.to_device(np.stack([pts["x"], pts["y"], pts["z"]], axis=1).astype(np.float32).reshape(-1))
        d_t = ctx.to_device(np.ascontiguousarray(pts["time"], np.float64))
        desc = Rec.Points(d_xyz.ptr, d_t.ptr, 12, 8, n)
    else:
        d = ctx.to_device(pts)
        desc = ctx.points_desc(d, n)
    t_lo, t_hi = float(pts["time"][0]), float(pts["time"][-1])
    res = {0: [], 1: []}
    counts = set()
    for rep in range(4):
        for sync in (1, 0):
            ctx.set_dev_option("ex_sync", sync)
            for _ in range(10):
                ctx.extract_enqueue(desc, d_out, d_ids, cap, t_lo, t_hi)
                counts.add(ctx.extract_finish())
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                ctx.extract_enqueue(desc, d_out, d_ids, cap, t_lo, t_hi)
                m = ctx.extract_finish()
            ctx.sync()
            res[sync].append((time.perf_counter() - t0) / steps * 1e3)
            counts.add(m)
    print("%-10s stream wait %s   ticket %s   surfel counts %s" % (name, ["%.4f" % v for v in res[1]], ["%.4f" % v for v in res[0]], sorted(counts)), flush=True)
