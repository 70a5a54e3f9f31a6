# N batched extractions of K C2 sweeps (run under rocprofv3 --kernel-trace --stats): python profiles/dev/prof_batch.py [K] [reps]
import os, sys
R0 = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R0 + "/wildcat-slam_amd/python"]
import numpy as np
from wildcat_slam_amd import lib, synth, records as R
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = lib.Context(0)
jobs, keep = [], []
for k in range(K):
    p = synth.g2_lattice(3906, m=32, seed=synth.SEED + 300 + k)[0]
    n = len(p); cap = (3 * n) // 20 + 1
    d = ctx.to_device(p); o, i = ctx.alloc(144 * cap), ctx.alloc(16 * cap)
    keep.append((d, o, i))
    jobs.append((ctx.points_desc(d, n), o, i, cap, float(p["time"][0]), float(p["time"][-1])))
enq, fin = ctx.extract_batch_prepare(jobs)
for _ in range(reps):
    enq(); c = fin()
print(c)
