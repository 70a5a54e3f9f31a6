"""End-to-end run of the host facade (C++ LidarOdometry with the reference's public interface, lidar_odometry.h:11-25)
on a synthetic raw sensor stream: lidar messages in the lidar frame + 200 Hz IMU with a gyro bias.  Checks that the
window bookkeeping around the hot path holds together over several sweeps and that the solve pulls the trajectory
towards the truth (IMU dead reckoning alone drifts with the bias)."""
import numpy as np
import pytest

from wildcat_slam_amd import synth

pytestmark = pytest.mark.gpu


def _drive(odo, msgs, imu):
    k = 0
    for m in msgs:
        if len(m) == 0:
            continue
        t_end = m["time"][-1]
        while k < len(imu["t"]) and imu["t"][k] <= t_end + 0.02:
            odo.add_imu(imu["t"][k], imu["acc"][k], imu["gyr"][k])
            k += 1
        odo.add_scan(m)


def test_facade_multi_sweep(gpu):
    from wildcat_slam_amd import lib

    msgs, imu, truth = synth.raw_stream(3.2, pts_per_s=500_000, gyro_bias=(0.0, 0.0, 0.02), t_start=1000.0)
    odo = lib.Odometry(0)
    _drive(odo, msgs, imu)
    assert odo.sweeps() >= 5
    st = odo.stats()
    assert st["sld_surfels"] > 2000 and st["binary"] > 500 and st["lm_iters"] >= 1
    assert st["cost1"] < st["cost0"]
    s = odo.samples()
    p_true, R_true = truth(s[:, 0])
    err = np.linalg.norm(s[:, 1:4] - p_true, axis=1)
    # yaw error of the last sample state vs pure gyro integration of the biased gyro (0.02 rad/s * t)
    q = s[-1, 4:8]
    yaw_est = np.arctan2(2 * (q[0] * q[3] + q[1] * q[2]), 1 - 2 * (q[2] ** 2 + q[3] ** 2))
    yaw_true = np.arctan2(R_true[-1][1, 0], R_true[-1][0, 0])
    t_rel = s[-1, 0] - 1000.0
    print("pos err max", err.max(), "yaw err", yaw_est - yaw_true, "dead-reckoning yaw err would be", 0.02 * t_rel)
    assert abs(yaw_est - yaw_true) < 0.5 * 0.02 * t_rel
    assert err.max() < 0.25
    odo.close()
