"""per-sweep stage split of the facade on the room stream (bench.py: facade_stream): python profiles/dev/facade_sweeps.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "wildcat-slam_amd", "python"))
import numpy as np
from wildcat_slam_amd import lib, synth

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.2
msgs, imu, _ = synth.raw_stream(secs, pts_per_s=640_000, gyro_bias=(0.0, 0.0, 0.02), t_start=1000.0)
odo = lib.Odometry(0)
k, before = 0, 0
for m in msgs:
    if len(m) == 0:
        continue
    t_end = m["time"][-1]
    while k < len(imu["t"]) and imu["t"][k] <= t_end + 0.02:
        odo.add_imu(imu["t"][k], imu["acc"][k], imu["gyr"][k])
        k += 1
    t0 = time.perf_counter()
    odo.add_scan(m)
    dt = time.perf_counter() - t0
    if odo.sweeps() > before:
        before = odo.sweeps()
        st, s = odo.stage_ms(), odo.stats()
        print("sweep %2d %6.2f ms | " % (before, dt * 1e3) + " ".join("%s %.2f" % (a, b) for a, b in st.items()) + " | sld %d fix %d bin %d un %d" % (
            s["sld_surfels"], s["fix_surfels"], s["binary"], s["unary"]))
