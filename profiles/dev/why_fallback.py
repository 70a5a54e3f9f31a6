import sys
sys.path[:0] = ["/root/repo/wildcat-slam_amd/python"]
import numpy as np
from wildcat_slam_amd import lib, synth
ctx = lib.Context(0)
for n in (1_000_000, 2_100_000, 2_335_310, 4_000_000):
    pts = synth.g1_room(n, seed=5)
    for rep in range(4):
        s, i = ctx.extract_surfels(pts)
        print(n, rep, len(s), ctx.extract_path_info())
