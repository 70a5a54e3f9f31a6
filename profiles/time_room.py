"""Extraction time on the G1 'room raycast' sweep (spinning-lidar point order: many short runs per voxel, layer-2 work),
next to the regular G2 lattice the headline is quoted on.  python profiles/time_room.py [n_points]"""
import sys, time
sys.path.insert(0, "wildcat-slam_amd/python")
import numpy as np
from wildcat_slam_amd import lib, synth, records as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ctx = lib.Context(0)
for name, pts in (("g1_room", synth.g1_room(n)), ("g2_lattice", synth.g2_lattice(n // 256, m=32)[0])):
    cap = (3 * len(pts)) // 20 + 1
    d_pts = ctx.to_device(pts)
    d_out, d_ids = ctx.alloc(cap * 144), ctx.alloc(cap * 16)
    desc = ctx.points_desc(d_pts, len(pts))
    t_lo, t_hi = float(pts["time"][0]), float(pts["time"][-1])
    for _ in range(5):
        ctx.extract_enqueue(desc, d_out, d_ids, cap, t_lo, t_hi); ns = ctx.extract_finish()
    ctx.sync(); t0 = time.perf_counter()
    K = 50
    for _ in range(K):
        ctx.extract_enqueue(desc, d_out, d_ids, cap, t_lo, t_hi); ns = ctx.extract_finish()
    ctx.sync(); dt = (time.perf_counter() - t0) / K
    ctx.extract_profile(True)
    ctx.extract_enqueue(desc, d_out, d_ids, cap, t_lo, t_hi); ctx.extract_finish()
    st = ctx.extract_stage_ms(); ctx.extract_profile(False)
    out = (lib.C.c_uint32 * 64)(); ctx.lib.wc_debug_status(ctx.h, out)
    print(f"{name}: {len(pts)} pts -> {ns} surfels, {dt*1e3:.4f} ms/sweep, {len(pts)/dt/1e6:.0f} Mpts/s, flags={out[1]}, layer-2 roots={out[4]}, stages(ms)={ {k: round(v,4) for k,v in st.items()} }")
