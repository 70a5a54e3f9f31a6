// imu_resampler.h — host-side stand-in for the reference's ImuResampler (src/sensor/imu_resampler.h:11-54), the hook
// wildcat_slam_node.cc:30-44 puts in front of LidarOdometry::AddImuData: an irregular IMU stream becomes a fixed-rate one by
// linear interpolation between the two newest raw samples.  Same class surface (ctor(int freq), AddImuData,
// AdvanceGetResampledImuData returning a shared_ptr that is null when no sample is due); KAT: imu_resampler_test.cc:7-31
// (tests/test_host_kat.py drives this class through host/odom_c_api.cc).
#pragma once
#include <memory>

#ifndef WC_HAVE_REFERENCE_TYPES
#include "shim/common.h"
#endif

class ImuResampler {
 public:
  explicit ImuResampler(int freq) : period_(1.0 / freq) {}

  // keeps the two newest raw samples (imu_resampler.h:16-21)
  void AddImuData(const ImuData &imu_data) {
    if (count_ == 2) {
      pair_[0] = pair_[1];
      pair_[1] = imu_data;
    } else {
      pair_[count_++] = imu_data;
    }
  }

  // one resampled measurement per call while the next grid time lies inside [older, newer] (imu_resampler.h:23-45):
  // the very first call hands out the older raw sample itself and starts the grid at its timestamp
  std::shared_ptr<ImuData> AdvanceGetResampledImuData() {
    if (count_ < 2) return nullptr;
    if (!started_) {
      started_ = true;
      last_out_time_ = pair_[0].timestamp;
      return std::make_shared<ImuData>(pair_[0]);
    }
    const double target = last_out_time_ + period_;
    const ImuData &a = pair_[0], &b = pair_[1];
    if (!(a.timestamp <= target && target <= b.timestamp)) return nullptr;
    const double f = (target - a.timestamp) / (b.timestamp - a.timestamp);
    auto out = std::make_shared<ImuData>();
    out->timestamp = target;
    for (int d = 0; d < 3; ++d) {
      out->linear_acceleration[d] = (1 - f) * a.linear_acceleration[d] + f * b.linear_acceleration[d];
      out->angular_velocity[d] = (1 - f) * a.angular_velocity[d] + f * b.angular_velocity[d];
    }
    last_out_time_ = target;
    return out;
  }

 private:
  ImuData pair_[2];
  int count_ = 0;
  double period_;
  double last_out_time_ = 0.0;
  bool started_ = false;
};
