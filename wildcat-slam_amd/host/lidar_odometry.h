// lidar_odometry.h — host-side drop-in for the reference's odometry facade.
//
// Public surface identical to the reference's src/odometry/lidar_odometry.h:11-25
//     LidarOdometry();  void AddImuData(const ImuData&);  void AddLidarScan(const pcl::PointCloud<hilti_ros::Point>::Ptr&);
// so that src/wildcat_slam_node.cc (:42, :51, :66) compiles against it unchanged where ROS / PCL exist.  Everything the
// reference does between lidar_odometry.cc:520 and :566 (sweep undistortion, surfel extraction, surfel pose update, correspondence,
// factor construction, the Ceres solve) goes through the C-ABI of libwildcat_hip.so; the window bookkeeping around it
// (point pre-filter :489-496, SyncHeadingMsgs :457-485, PredictImuStatesAndSampleStates :365-455, BuildSweep :134-141,
// UpdateImuPoses + B-spline corrector :22-54,:187-215, UpdateSamplePoses :172-179,
// ShrinkToFit :228-250) is restated here as plain host C++.  All state is per instance (the reference keeps some of it
// in function-local statics).  ROS publishing is replaced by the accessors at the bottom.
#pragma once
#include <cstdint>
#include <deque>
#include <memory>
#include <string>
#include <vector>

#ifndef WC_HAVE_REFERENCE_TYPES
#include "shim/common.h"
#endif
#include "../../include/wildcat_hip.h"
#include "lio_config.h"
#include "wire_formats.h"

class LidarOdometry {
 public:
  LidarOdometry();
  explicit LidarOdometry(int device);
  ~LidarOdometry();
  LidarOdometry(const LidarOdometry &) = delete;
  LidarOdometry &operator=(const LidarOdometry &) = delete;

  /** Add raw imu measurements to queue (lidar_odometry.h:14-18) */
  void AddImuData(const ImuData &msg);
  /** Add raw lidar points with timestamp (lidar_odometry.h:20-25) */
  void AddLidarScan(const pcl::PointCloud<hilti_ros::Point>::Ptr &msg);

  // ---- not in the reference: read-outs instead of ROS topics / TF (lidar_odometry.cc:582-602) ----
  struct SampleStateView {
    double timestamp;
    double pos[3];
    double quat[4];  // w, x, y, z
    double bg[3], ba[3];
  };
  int sweeps_done() const { return sweep_id_; }
  bool latest_state(SampleStateView *out) const;
  size_t num_sample_states() const { return samples_.size(); }
  bool sample_state(size_t i, SampleStateView *out) const;
  size_t sliding_window_surfels() const { return n_surfels_ - sld_begin_; }
  size_t fixed_window_surfels() const { return fix_end_ - fix_start_; }
  // timestamps of the fixed window in its own order (newest first, like the reference's deque after push_front, Q11)
  const std::deque<double> &fixed_window_times() const { return fix_times_; }
  const wc_solve_summary &last_solve() const { return last_summary_; }
  // wall time [ms] of the last completed sweep's stages: predict + undistort, extract + poses, match, build, solve, update, shrink
  const double *last_stage_ms() const { return last_stage_ms_; }
  int last_lm_iterations() const { return last_lm_iterations_; }
  // wall time [ms] the message that completed the last sweep spent in front of the stages: upload + pre-filter of its points
  // (AppendScanOnDevice) and the heading synchronisation
  double last_append_ms() const { return last_append_ms_; }
  uint64_t last_correspondences(int which) const { return last_corr_[which]; }
  // test hook: keep, for the last completed sweep's last outer iteration, the two surfel timestamps of every correspondence
  // (which = 0: sliding window, 1: fixed window) - what a comparison with another run can match pairs on when surfel ORDER differs
  void set_keep_pair_stamps(bool on) { keep_pair_stamps_ = on; }
  const std::vector<double> &last_pair_stamps(int which) const { return pair_stamps_[which]; }
  // sweeps whose extraction was completed by the default (integer-moment) path / by the reference-order path (configured, or
  // fallen back to because a gate lay inside the reference's own rounding noise)
  int sweeps_fast_path() const { return sweeps_fast_; }
  int sweeps_exact_path() const { return sweeps_exact_; }
  // what the reference publishes after a sweep (lidar_odometry.cc:582-602), as plain data (config().fill_outputs)
  struct SweepOutputs {
    std::vector<wc_wire::SurfelMarker> markers;  // PubSurfels(surfels_sld_win_) (:582), one SPHERE per sliding-window surfel
    wc_wire::PointCloud2 scan_in_world;          // the sweep undistorted with the final IMU poses (:584-595), frame "world"
    double scan_stamp = 0;                       // msg.header.stamp = time of the sweep's first point (:592)
    wc_wire::StampedTransform tf{};              // world -> imu_link at the last sample state (:596-602)
  };
  const SweepOutputs &last_outputs() const { return outputs_; }
  // the residual histograms of the last completed sweep (config().log_residual_histograms; lidar_odometry.cc:56-94)
  const std::string &last_residual_log() const { return residual_log_; }
  LioConfig &config() { return config_; }
  void ApplyConfig();  // push config() changes (quirks, extraction arithmetic, iteration cap, extrinsics) into the device context
  bool ImportState(const double *samples23, size_t ns, const wc_imu_state *imu, size_t n_imu);  // test hook, see .cc

 private:
  struct Sample {  // reference SampleState (surfel.h:9-23)
    double timestamp;
    double cor[12];  // rot_cor, pos_cor, bg, ba
    double grav[3];
    double quat[4];
    double pos[3];
  };
  void PredictImuStatesAndSampleStates(double end_time);
  bool SyncHeadingMsgs();
  void UploadImuStates();
  void LogResiduals(const std::vector<double> &x, const char *when);
  void UpdateImuPoses();
  void UpdateSamplePoses();
  void UpdateSurfelPosesOnDevice();
  void ShrinkToFit();
  void EnsureSurfelCapacity(size_t n);
  void EnsureFixedCapacity(size_t extra);
  void AppendScanOnDevice(const pcl::PointCloud<hilti_ros::Point> &msg);
  void DropBufferedPoints(size_t k);
  void Fatal(const char *what, int rc) const;

  LioConfig config_;
  wc_ctx *ctx_ = nullptr;
 public:
  // the library context behind this object (tests and profiling scripts set development options on it: wc_ctx_set_dev_option)
  wc_ctx *gpu_context() const { return ctx_; }
 private:
  std::deque<ImuData> imu_buff_;
  // points_buff_ of the reference (lidar_odometry.h:56) lives in HBM: the pre-filtered points of the scans not yet
  // consumed are d_pts_[pts_cur_][pts_begin_ .. pts_end_); the host keeps only their timestamps
  std::deque<double> point_times_;
  void *d_pts_[2] = {nullptr, nullptr};
  void *d_scan_raw_ = nullptr;
  size_t cap_pts_[2] = {0, 0}, cap_scan_raw_ = 0, pts_begin_ = 0, pts_end_ = 0;
  int pts_cur_ = 0;
  std::deque<Sample> samples_;
  std::deque<wc_imu_state> imu_states_;
  // contiguous copy of imu_states_ (the C-ABI takes arrays) and whether the device copy follows it: rebuilt / uploaded only after a change
  // (prediction, post-solve correction, shrink, import) - a sweep flattened and uploaded the ~0.3 MB four to five times
  std::vector<wc_imu_state> imu_flat_;
  bool imu_flat_stale_ = true, imu_dev_stale_ = true;
  void TouchImu() { imu_flat_stale_ = imu_dev_stale_ = true; }
  const std::vector<wc_imu_state> &FlatImu();
  double ext_quat_[4];
  bool init_sld_win_ = false, sync_done_ = false, first_sample_known_ = false;
  double first_sample_time_ = 0.0;
  int sweep_id_ = 0, sweeps_fast_ = 0, sweeps_exact_ = 0;
  // sliding window in HBM, time ordered: d_surf_[sld_begin_ .. n_surfels_) ([0, sld_begin_) has moved to the fixed window and
  // is dropped at the next reallocation).  Fixed window: d_fix_surf_[fix_start_ .. fix_end_), NEWEST first - the array is
  // filled from the back because the reference push_front()s (lidar_odometry.cc:243-246, Q11)
  wc_surfel *d_surf_ = nullptr;
  wc_surfel *d_fix_surf_ = nullptr;
  wc_pose *d_fix_pose_ = nullptr;
  size_t fix_cap_ = 0, fix_start_ = 0, fix_end_ = 0;
  std::deque<double> fix_times_;
  wc_pose *d_pose_ = nullptr;
  uint8_t *d_inbody_ = nullptr;
  wc_pair *d_pairs_sld_ = nullptr, *d_pairs_fix_ = nullptr;
  wc_imu_state *d_imu_ = nullptr;
  void *d_sweep_xyz_ = nullptr, *d_sweep_t_ = nullptr;  // the undistorted sweep, packed: 3 floats | 1 double per point
  void *d_kept_t_ = nullptr;  // stamps of the points the pre-filter kept (scratch of AppendScanOnDevice)
  size_t cap_kept_t_ = 0;
  size_t cap_surfels_ = 0, n_surfels_ = 0, sld_begin_ = 0, cap_imu_ = 0, cap_sweep_ = 0;
  std::deque<double> surfel_times_;  // host copy of the sliding window's surfel timestamps (window bookkeeping only)
  SweepOutputs outputs_;
  void FillOutputs(const void *d_raw_sweep, size_t n_sweep);
  std::string residual_log_;
  void *d_res_ = nullptr;
  size_t cap_res_ = 0;
  wc_solve_summary last_summary_{};
  double last_stage_ms_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int last_lm_iterations_ = 0;  // over the sweep's outer iterations
  double last_append_ms_ = 0.0;
  uint64_t last_corr_[2] = {0, 0};
  bool keep_pair_stamps_ = false;
  std::vector<double> pair_stamps_[2];  // (first, second) stamps, pair after pair
};
