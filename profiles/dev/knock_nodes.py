"""knock-out timing of k_fx_nodes (development option debug_skip): python profiles/dev/knock_nodes.py [c2|room]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "wildcat-slam_amd", "python"))
import numpy as np
from wildcat_slam_amd import lib, synth
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
pts = {"c2": lambda: synth.g2_lattice(3906, m=32)[0], "room": lambda: synth.g1_room(1_000_000, seed=synth.SEED + 3)}[which]()
n = len(pts)
cap = (3 * n) // 20 + 1
for bits in (0, 2048, 512, 256, 1024, 256 | 1024, 8192, 16384):
    ctx = lib.Context(0)
    ctx.set_dev_option("debug_skip", bits)
    d_out, d_ids = ctx.alloc(cap * 144), ctx.alloc(cap * 16)
    d = ctx.to_device(pts)
    desc = ctx.points_desc(d, n)
    t_lo, t_hi = float(pts["time"][0]), float(pts["time"][-1])
    ctx.extract_profile(True)
    acc, emit = [], []
    for i in range(24):
        try:
            ctx.extract_enqueue(desc, d_out, d_ids, cap, t_lo, t_hi)
            ctx.extract_finish()
        except Exception as e:
            pass
        if i >= 4:
            st = ctx.extract_stage_ms()
            acc.append(st["roots_stream"]); emit.append(st["slot_order"])
    print("skip %5d: k_fx_nodes %.1f us (min %.1f)  slot_emit %.1f" % (bits, 1e3 * float(np.median(acc)), 1e3 * min(acc), 1e3 * float(np.median(emit))))
    ctx.close()
