"""Wall time of the host facade (LidarOdometry::AddLidarScan, reference interface) per sweep on the synthetic raw stream of
tests/test_facade_gpu.py: prefilter + undistort + extraction + pose update + two matches + window build + LM solve + post-solve
bookkeeping.  python profiles/time_facade.py [pts_per_s]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wildcat-slam_amd", "python"))  # (the tree this script lies in: A/B runs of two trees)
import numpy as np
from wildcat_slam_amd import lib, synth

pps = int(sys.argv[1]) if len(sys.argv) > 1 else 640_000
msgs, imu, truth = synth.raw_stream(4.0, pts_per_s=pps, gyro_bias=(0.0, 0.0, 0.02), t_start=1000.0)
odo = lib.Odometry(0)
k = 0
times, sweeps_before = [], 0
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
    if odo.sweeps() > sweeps_before:  # this message completed a sweep
        sweeps_before = odo.sweeps()
        times.append((dt, len(m), dict(odo.stats())))
print("sweeps", len(times), "points/message", int(np.mean([n for _, n, _ in times])))
for dt, n, st in times[-6:]:
    print(f"  add_scan {dt*1e3:7.2f} ms  " + "  ".join(f"{k_}={v_:g}" for k_, v_ in st.items()))
ts = np.array([t for t, _, _ in times[2:]])
print(f"median {np.median(ts)*1e3:.2f} ms per sweep, max {ts.max()*1e3:.2f} ms")
