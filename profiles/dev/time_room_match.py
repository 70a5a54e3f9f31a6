# the matcher on the room stream's surfels (bench.py: match_room_stream), a few repetitions: python profiles/dev/time_room_match.py
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, R + "/wildcat-slam_amd/python"]
import bench
from wildcat_slam_amd import lib
ctx = lib.Context(0)
for _ in range(3):
    r = bench.bench_match_room(ctx)
    print(r["workload"], r["ms_per_search"], r["ms_per_50k_queries"], ctx.match_stats())
