// lidar_odometry.cc — host orchestration of one odometry instance around the MI355X C-ABI (see lidar_odometry.h).
// Every block names the reference lines it stands in for (src/odometry/lidar_odometry.cc unless noted).
#include "lidar_odometry.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>

#include "../csrc/dmath.h"
#include "cubic_bspline.h"
#include "histogram.h"

using namespace wc;

namespace {

inline V3 v3(const double *p) { return mk3(p[0], p[1], p[2]); }
inline Q4 q4(const double *p) { return Q4{p[0], p[1], p[2], p[3]}; }
inline void st3(double *d, V3 v) { d[0] = v.x, d[1] = v.y, d[2] = v.z; }
inline void stq(double *d, Q4 q) { d[0] = q.w, d[1] = q.x, d[2] = q.y, d[3] = q.z; }

// rotation matrix (row-major) -> unit quaternion; stands in for Eigen::Quaterniond(Matrix3d) + Rigid3's normalisation
Q4 quat_from_matrix(const double *m) {
  const double tr = m[0] + m[4] + m[8];
  Q4 q;
  if (tr > 0) {
    double s = std::sqrt(tr + 1.0) * 2;
    q = {0.25 * s, (m[7] - m[5]) / s, (m[2] - m[6]) / s, (m[3] - m[1]) / s};
  } else if (m[0] > m[4] && m[0] > m[8]) {
    double s = std::sqrt(1.0 + m[0] - m[4] - m[8]) * 2;
    q = {(m[7] - m[5]) / s, 0.25 * s, (m[1] + m[3]) / s, (m[2] + m[6]) / s};
  } else if (m[4] > m[8]) {
    double s = std::sqrt(1.0 + m[4] - m[0] - m[8]) * 2;
    q = {(m[2] - m[6]) / s, (m[1] + m[3]) / s, 0.25 * s, (m[5] + m[7]) / s};
  } else {
    double s = std::sqrt(1.0 + m[8] - m[0] - m[4]) * 2;
    q = {(m[3] - m[1]) / s, (m[2] + m[6]) / s, (m[5] + m[7]) / s, 0.25 * s};
  }
  const double n = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  return {q.w / n, q.x / n, q.y / n, q.z / n};
}

// PredictPoseOfNewImuState (:106-123): constant-acceleration / mid-point gyro dead reckoning
void PredictPoseOfNewImuState(const wc_imu_state &i1, const wc_imu_state &i2, V3 ba, V3 bg, V3 grav, wc_imu_state &i3) {
  const double dt = i3.t - i2.t;
  stq(i3.quat, qmul(q4(i2.quat), so3_exp(((v3(i2.gyr) + v3(i3.gyr)) / 2 - bg) * dt)));
  st3(i3.pos, ((qrot(q4(i1.quat), v3(i1.acc) - ba) + grav) * dt) * dt + 2 * v3(i2.pos) - v3(i1.pos));
}

}  // namespace

void LidarOdometry::Fatal(const char *what, int rc) const {
  // the reference aborts through glog CHECK / LOG(FATAL); so does the facade
  std::fprintf(stderr, "[wildcat] FATAL %s (rc=%d): %s\n", what, rc, ctx_ ? wc_last_error(ctx_) : "");
  std::abort();
}
#define WC_CHECK(cond)                                                                           \
  do {                                                                                           \
    if (!(cond)) {                                                                               \
      std::fprintf(stderr, "[wildcat] CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__);   \
      std::abort();                                                                              \
    }                                                                                            \
  } while (0)
#define WC_CALL(expr)                  \
  do {                                 \
    int rc_ = (expr);                  \
    if (rc_ != WC_OK) Fatal(#expr, rc_); \
  } while (0)

LidarOdometry::LidarOdometry() : LidarOdometry(0) {}

// the library parameters that follow LioConfig: solver iterations, the quirk Jacobians (Q1 / Q3) and the extraction arithmetic
static void ParamsFromConfig(const LioConfig &c, wc_params *P) {
  wc_params_default(P);
  P->max_iterations = c.inner_iter_num_max;
  P->reference_quirks = c.reference_quirks ? 1 : 0;
  P->exact_sums = c.exact_sums ? 1 : 0;
  P->imu_dt = 1 / c.imu_rate;
}

LidarOdometry::LidarOdometry(int device) {
  wc_params P;
  ParamsFromConfig(config_, &P);
  int rc = wc_ctx_create(&P, device, &ctx_);
  if (rc != WC_OK) {
    std::fprintf(stderr, "[wildcat] FATAL: no MI355X context (rc=%d); there is no CPU fallback\n", rc);
    std::abort();
  }
  // one-time costs out of the first sweeps: code objects, the matcher's helper context and thread, 2 GB of HBM taken into the pool
  // the library's scratch buffers grow from (a 6.5 s window of a 640 k points/s scanner uses ~1 GB of them)
  rc = wc_ctx_warmup(ctx_, (size_t)2 << 30);
  if (rc != WC_OK) std::fprintf(stderr, "[wildcat] wc_ctx_warmup: %s\n", wc_last_error(ctx_));
  stq(ext_quat_, quat_from_matrix(config_.ext_rotation));
}

// config() hands out a mutable LioConfig; the values the library holds a copy of (wc_params) and the extrinsic quaternion are
// re-derived here.  Call after changing config() and before the next AddLidarScan.
void LidarOdometry::ApplyConfig() {
  wc_params P;
  ParamsFromConfig(config_, &P);
  WC_CALL(wc_ctx_set_params(ctx_, &P));
  stq(ext_quat_, quat_from_matrix(config_.ext_rotation));
}

// test hook (not in the reference): overwrite the window's sample states and IMU states with another run's (23 doubles per
// sample state: timestamp, cor[12], grav[3], quat[4], pos[3]) and re-attach the surfel poses, so that every sweep of a
// comparison starts from the same states.  The counts must equal the facade's own.
bool LidarOdometry::ImportState(const double *samples23, size_t ns, const wc_imu_state *imu, size_t n_imu) {
  if (ns != samples_.size() || n_imu != imu_states_.size()) return false;
  for (size_t i = 0; i < ns; ++i) {
    const double *p = samples23 + 23 * i;
    Sample &s = samples_[i];
    if (s.timestamp != p[0]) return false;
    std::memcpy(s.cor, p + 1, 96);
    std::memcpy(s.grav, p + 13, 24);
    std::memcpy(s.quat, p + 16, 32);
    std::memcpy(s.pos, p + 20, 24);
  }
  for (size_t i = 0; i < n_imu; ++i) imu_states_[i] = imu[i];
  TouchImu();
  UpdateSurfelPosesOnDevice();
  return true;
}

LidarOdometry::~LidarOdometry() {
  if (!ctx_) return;
  if (d_res_) wc_dev_free(ctx_, d_res_);
  void *bufs[] = {d_surf_, d_pose_, d_inbody_, d_pairs_sld_, d_pairs_fix_, d_imu_, d_sweep_xyz_, d_sweep_t_, d_kept_t_, d_scan_raw_, d_pts_[0], d_pts_[1], d_fix_surf_, d_fix_pose_};
  for (void *b : bufs)
    if (b) wc_dev_free(ctx_, b);
  wc_ctx_destroy(ctx_);
}

void LidarOdometry::AddImuData(const ImuData &msg) { imu_buff_.push_back(msg); }  // :607-611

bool LidarOdometry::latest_state(SampleStateView *out) const {
  return !samples_.empty() && sample_state(samples_.size() - 1, out);
}
bool LidarOdometry::sample_state(size_t i, SampleStateView *out) const {
  if (i >= samples_.size()) return false;
  const Sample &s = samples_[i];
  out->timestamp = s.timestamp;
  std::memcpy(out->pos, s.pos, 24);
  std::memcpy(out->quat, s.quat, 32);
  std::memcpy(out->bg, s.cor + 6, 24);
  std::memcpy(out->ba, s.cor + 9, 24);
  return true;
}

// room for n more sliding-window surfels behind n_surfels_; a reallocation drops the dead prefix [0, sld_begin_)
void LidarOdometry::EnsureSurfelCapacity(size_t n) {
  if (n_surfels_ + n <= cap_surfels_) return;
  const size_t live = n_surfels_ - sld_begin_;
  const size_t cap = std::max<size_t>((live + n) + (live + n) / 2, 1 << 18);  // (262 k surfels = 57 MB from the start: the room stream's window crossed 65 k in its fifth sweep - five allocations and three copies, 0.5 ms)
  void *ns = nullptr, *np = nullptr, *nb = nullptr, *p1 = nullptr, *p2 = nullptr;
  WC_CALL(wc_dev_alloc(ctx_, cap * sizeof(wc_surfel), &ns));
  WC_CALL(wc_dev_alloc(ctx_, cap * sizeof(wc_pose), &np));
  WC_CALL(wc_dev_alloc(ctx_, cap, &nb));
  WC_CALL(wc_dev_alloc(ctx_, cap * sizeof(wc_pair), &p1));
  WC_CALL(wc_dev_alloc(ctx_, cap * sizeof(wc_pair), &p2));
  if (live) {
    WC_CALL(wc_d2d(ctx_, ns, d_surf_ + sld_begin_, live * sizeof(wc_surfel)));
    WC_CALL(wc_d2d(ctx_, np, d_pose_ + sld_begin_, live * sizeof(wc_pose)));
    WC_CALL(wc_d2d(ctx_, nb, d_inbody_ + sld_begin_, live));
  }
  void *old[] = {d_surf_, d_pose_, d_inbody_, d_pairs_sld_, d_pairs_fix_};
  for (void *b : old)
    if (b) WC_CALL(wc_dev_free(ctx_, b));
  d_surf_ = (wc_surfel *)ns, d_pose_ = (wc_pose *)np, d_inbody_ = (uint8_t *)nb;
  d_pairs_sld_ = (wc_pair *)p1, d_pairs_fix_ = (wc_pair *)p2;
  cap_surfels_ = cap;
  n_surfels_ = live;
  sld_begin_ = 0;
}

// room for `extra` more surfels in FRONT of the fixed window (it is filled from the back of its array)
void LidarOdometry::EnsureFixedCapacity(size_t extra) {
  if (extra <= fix_start_) return;
  const size_t live = fix_end_ - fix_start_;
  const size_t cap = std::max<size_t>(2 * (live + extra), 1 << 16);
  void *ns = nullptr, *np = nullptr;
  WC_CALL(wc_dev_alloc(ctx_, cap * sizeof(wc_surfel), &ns));
  WC_CALL(wc_dev_alloc(ctx_, cap * sizeof(wc_pose), &np));
  if (live) {
    WC_CALL(wc_d2d(ctx_, (wc_surfel *)ns + (cap - live), d_fix_surf_ + fix_start_, live * sizeof(wc_surfel)));
    WC_CALL(wc_d2d(ctx_, (wc_pose *)np + (cap - live), d_fix_pose_ + fix_start_, live * sizeof(wc_pose)));
  }
  if (d_fix_surf_) WC_CALL(wc_dev_free(ctx_, d_fix_surf_));
  if (d_fix_pose_) WC_CALL(wc_dev_free(ctx_, d_fix_pose_));
  d_fix_surf_ = (wc_surfel *)ns, d_fix_pose_ = (wc_pose *)np;
  fix_cap_ = cap, fix_end_ = cap, fix_start_ = cap - live;
}

// the per-point loop of AddLidarScan (:489-496) on the device: upload the raw message, wc_prefilter_points (extrinsic,
// range / blind-box filter) appends the survivors to the device-resident points_buff_; only their timestamps come back
void LidarOdometry::AppendScanOnDevice(const pcl::PointCloud<hilti_ros::Point> &msg) {
  const size_t n = msg.size();
  if (n == 0) return;
  if (n > cap_scan_raw_) {
    if (d_scan_raw_) WC_CALL(wc_dev_free(ctx_, d_scan_raw_));
    cap_scan_raw_ = n + n / 2;
    WC_CALL(wc_dev_alloc(ctx_, cap_scan_raw_ * sizeof(hilti_ros::Point), &d_scan_raw_));
  }
  WC_CALL(wc_h2d(ctx_, d_scan_raw_, msg.points.data(), n * sizeof(hilti_ros::Point)));
  // make room behind pts_end_: consumed points in front are dropped by moving the live range to the other buffer
  const size_t live = pts_end_ - pts_begin_;
  if (pts_end_ + n > cap_pts_[pts_cur_]) {
    const int other = pts_cur_ ^ 1;
    if (live + n > cap_pts_[other]) {
      if (d_pts_[other]) WC_CALL(wc_dev_free(ctx_, d_pts_[other]));
      cap_pts_[other] = 2 * (live + n);
      WC_CALL(wc_dev_alloc(ctx_, cap_pts_[other] * sizeof(hilti_ros::Point), &d_pts_[other]));
    }
    if (live)
      WC_CALL(wc_d2d(ctx_, d_pts_[other], (const char *)d_pts_[pts_cur_] + pts_begin_ * sizeof(hilti_ros::Point), live * sizeof(hilti_ros::Point)));
    pts_cur_ = other, pts_begin_ = 0, pts_end_ = live;
  }
  uint64_t kept = 0;
  char *dst = (char *)d_pts_[pts_cur_] + pts_end_ * sizeof(hilti_ros::Point);
  if (n > cap_kept_t_) {
    if (d_kept_t_) WC_CALL(wc_dev_free(ctx_, d_kept_t_));
    cap_kept_t_ = n + n / 2;
    WC_CALL(wc_dev_alloc(ctx_, cap_kept_t_ * sizeof(double), &d_kept_t_));
  }
  // CHECK(points_buff_.empty() || pt.time >= points_buff_.back().time) (:491) runs on the device too: every incoming point,
  // filtered or not, against the last BUFFERED point at that moment; the survivors' stamps come back packed
  int monotonic = 1;
  WC_CALL(wc_prefilter_points_checked(ctx_, d_scan_raw_, n, ext_quat_, config_.ext_translation, config_.min_range, config_.max_range,
                                      config_.blind_min, config_.blind_max, dst, cap_pts_[pts_cur_] - pts_end_, &kept,
                                      point_times_.empty() ? -INFINITY : point_times_.back(), (double *)d_kept_t_, &monotonic));
  WC_CHECK(monotonic);
  std::vector<double> kt(kept);
  if (kept) WC_CALL(wc_d2h(ctx_, kt.data(), d_kept_t_, kept * sizeof(double)));
  point_times_.insert(point_times_.end(), kt.begin(), kt.end());
  pts_end_ += kept;
}

void LidarOdometry::DropBufferedPoints(size_t k) {
  pts_begin_ += k;
  point_times_.erase(point_times_.begin(), point_times_.begin() + (long)k);
}

// SyncHeadingMsgs (:457-485)
bool LidarOdometry::SyncHeadingMsgs() {
  if (sync_done_) return true;
  if (imu_buff_.empty() || point_times_.empty()) return false;
  if (imu_buff_.back().timestamp < point_times_.front()) return false;
  while (imu_buff_.front().timestamp < point_times_.front()) {
    imu_buff_.pop_front();
    WC_CHECK(!imu_buff_.empty());
  }
  while (point_times_.front() < imu_buff_.front().timestamp) {
    DropBufferedPoints(1);
    WC_CHECK(!point_times_.empty());
  }
  sync_done_ = true;
  return true;
}

// PredictImuStatesAndSampleStates (:365-455)
void LidarOdometry::PredictImuStatesAndSampleStates(double end_time) {
  WC_CHECK(imu_buff_.size() >= 2);
  TouchImu();
  const double dt = 1 / config_.imu_rate;
  if (!init_sld_win_) {
    for (int i = 0; i < 2; ++i) {
      const ImuData m = imu_buff_.front();
      imu_buff_.pop_front();
      wc_imu_state s{};
      s.t = m.timestamp;
      for (int d = 0; d < 3; ++d) s.acc[d] = m.linear_acceleration[d], s.gyr[d] = m.angular_velocity[d], s.pos[d] = 0;
      if (i == 0)
        stq(s.quat, Q4{1, 0, 0, 0});
      else
        stq(s.quat, so3_exp(((v3(imu_states_.back().gyr) + v3(s.gyr)) / 2) * dt));
      imu_states_.push_back(s);
    }
    Sample ss{};
    ss.timestamp = imu_states_.front().t;
    const V3 a0 = v3(imu_states_.front().acc);
    st3(ss.grav, (-config_.gravity_norm) * (a0 / norm(a0)));
    std::memcpy(ss.quat, imu_states_.front().quat, 32);
    std::memcpy(ss.pos, imu_states_.front().pos, 24);
    samples_.push_back(ss);
    first_sample_time_ = ss.timestamp;
    first_sample_known_ = true;
    init_sld_win_ = true;
  }
  const double old_last = samples_.back().timestamp;
  const int add_size = (int)((end_time - old_last) / config_.sample_dt);
  const double add_last = old_last + config_.sample_dt * add_size;
  const V3 ba = v3(samples_.back().cor + 9), bg = v3(samples_.back().cor + 6), grav = v3(samples_.back().grav);
  while (!imu_buff_.empty()) {
    const size_t size = imu_states_.size();
    const ImuData m = imu_buff_.front();
    imu_buff_.pop_front();
    wc_imu_state s{};
    s.t = m.timestamp;
    for (int d = 0; d < 3; ++d) s.acc[d] = m.linear_acceleration[d], s.gyr[d] = m.angular_velocity[d];
    WC_CHECK(std::fabs((s.t - imu_states_[size - 1].t) - (imu_states_[size - 1].t - imu_states_[size - 2].t)) <= 1e-6);
    PredictPoseOfNewImuState(imu_states_[size - 2], imu_states_[size - 1], ba, bg, grav, s);
    imu_states_.push_back(s);
    if (s.t >= add_last) break;  // enough imu states
  }
  for (int i = 1; i <= add_size; ++i) {
    const double t = old_last + i * config_.sample_dt;
    Sample ss{};
    ss.timestamp = t;
    st3(ss.cor + 9, ba), st3(ss.cor + 6, bg), st3(ss.grav, grav);
    size_t idx = std::lower_bound(imu_states_.begin(), imu_states_.end(), t,
                                  [](const wc_imu_state &a, double b) { return a.t < b; }) - imu_states_.begin();
    WC_CHECK(idx != 0 && idx != imu_states_.size());
    const wc_imu_state &a = imu_states_[idx - 1], &b = imu_states_[idx];
    const double f = (t - a.t) / (b.t - a.t);
    stq(ss.quat, qslerp(q4(a.quat), f, q4(b.quat)));
    st3(ss.pos, (1 - f) * v3(a.pos) + f * v3(b.pos));
    WC_CHECK(f >= 0 && f <= 1);
    samples_.push_back(ss);
  }
}

// PrintSurfelResiduals x 2 + PrintImuResiduals (:56-94) on the problem wc_window_build holds: ONE wc_window_evaluate (the
// reference's three problem.Evaluate calls, each on one family's blocks, apply_loss_function = true) gives the loss-corrected
// residuals in the reference's block order - binary, unary, 12 per IMU factor.  The cost the reference prints is problem.Evaluate's
// 0.5 * sum rho(s): for the IMU family (TrivialLoss) half the squared norm; for the surfel families (CauchyLoss(0.4), cc:273,311)
// the corrected residual is r' = sqrt(rho'(s)) r with rho(s) = b log(1 + s / b), b = 0.16, so s = r'^2 / (1 - r'^2 / b) is
// recovered and rho summed - half the squared norm of r' would understate it (0.08 against 0.111 at s = b).  The
// text goes to last_residual_log() (and to stderr with WC_ODOM_DEBUG): the reference's LOG(INFO) lines, glog prefix aside.
void LidarOdometry::LogResiduals(const std::vector<double> &x, const char *when) {
  uint64_t cnt[4] = {0, 0, 0, 0};
  WC_CALL(wc_window_counts(ctx_, cnt));
  const size_t nb = cnt[0], nu = cnt[1], ni = cnt[2], nres = nb + nu + 12 * ni;
  if (nres == 0) return;
  if (nres > cap_res_) {
    if (d_res_) WC_CALL(wc_dev_free(ctx_, d_res_));
    cap_res_ = nres + nres / 2;
    WC_CALL(wc_dev_alloc(ctx_, cap_res_ * sizeof(double), &d_res_));
  }
  double cost = 0;
  WC_CALL(wc_window_evaluate(ctx_, x.data(), &cost, (double *)d_res_));
  std::vector<double> r(nres);
  WC_CALL(wc_d2h(ctx_, r.data(), d_res_, nres * sizeof(double)));
  char buf[96];
  wc_params P;
  ParamsFromConfig(config_, &P);
  const double cb = P.cauchy_a * P.cauchy_a;
  auto surfel = [&](size_t first, size_t n, const char *window_type) {
    if (n == 0) return;  // (:57-59)
    Histogram hist;
    double c = 0;
    for (size_t i = first; i < first + n; ++i) {
      hist.Add(r[i]);
      // (r'^2 = s / (1 + s / b) < b; for a strong outlier r'^2 rounds to b itself: the denominator is clamped, the logged cost stays
      // finite where the reference prints a finite rho - ADVICE r4)
      const double q = r[i] * r[i], s = q / std::max(1.0 - q / cb, 2.220446049250313e-16);
      c += cb * std::log1p(s / cb);
    }
    std::snprintf(buf, sizeof(buf), " Surfel residuals, cost: %g, dist: ", 0.5 * c);
    residual_log_ += std::string(when) + window_type + buf + hist.ToString(10) + "\n";
  };
  surfel(0, nb, "Sliding Window");
  surfel(nb, nu, "Fixed Window");
  if (ni) {  // (:74-93)
    Histogram hist[4];
    const char *types[4] = {"gyro", "acc", "gyro_bias", "acc_bias"};
    double c = 0;
    for (size_t i = nb + nu; i < nres; i += 12)
      for (int j = 0; j < 4; ++j) {
        const double *p = &r[i + 3 * j];
        const double s2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
        hist[j].Add(std::sqrt(s2));
        c += s2;
      }
    for (int j = 0; j < 4; ++j) {
      std::snprintf(buf, sizeof(buf), ", cost: %g, dist: ", 0.5 * c);
      residual_log_ += std::string(when) + "Imu residuals with type " + types[j] + buf + hist[j].ToString(10) + "\n";
    }
  }
}

// what the reference hands to ROS at the end of AddLidarScan (:582-602), as plain data
void LidarOdometry::FillOutputs(const void *d_raw_sweep, size_t n_sweep) {
  // PubSurfels(surfels_sld_win_, ...) (:582, surfel_extraction.cc:360-434)
  const size_t n = n_surfels_ - sld_begin_;
  std::vector<wc_surfel> surf(n);
  std::vector<wc_pose> pose(n);
  if (n) {
    WC_CALL(wc_d2h(ctx_, surf.data(), d_surf_ + sld_begin_, n * sizeof(wc_surfel)));
    WC_CALL(wc_d2h(ctx_, pose.data(), d_pose_ + sld_begin_, n * sizeof(wc_pose)));
  }
  outputs_.markers.resize(n);
  for (size_t i = 0; i < n; ++i) outputs_.markers[i] = wc_wire::MarkerFromSurfel(surf[i], pose[i]);
  // the sweep, undistorted once more with the poses the solve left behind (:584-595)
  std::vector<hilti_ros::Point> pts(n_sweep);
  if (n_sweep) {
    void *d_und = nullptr;
    WC_CALL(wc_dev_alloc(ctx_, n_sweep * sizeof(hilti_ros::Point), &d_und));
    UploadImuStates();
    WC_CALL(wc_undistort_sweep(ctx_, d_raw_sweep, n_sweep, d_imu_, imu_states_.size(), d_und));
    WC_CALL(wc_d2h(ctx_, pts.data(), d_und, n_sweep * sizeof(hilti_ros::Point)));
    WC_CALL(wc_dev_free(ctx_, d_und));
  }
  wc_wire::Cloud2FromPoints(pts.data(), pts.size(), outputs_.scan_in_world);
  outputs_.scan_stamp = n_sweep ? pts[0].time : 0.0;
  // tf world -> imu_link (:596-602)
  const Sample &b = samples_.back();
  outputs_.tf.stamp = b.timestamp;
  std::memcpy(outputs_.tf.origin, b.pos, 24);
  outputs_.tf.rotation_xyzw[0] = b.quat[1], outputs_.tf.rotation_xyzw[1] = b.quat[2], outputs_.tf.rotation_xyzw[2] = b.quat[3], outputs_.tf.rotation_xyzw[3] = b.quat[0];
}

void LidarOdometry::UploadImuStates() {
  const size_t n_imu = imu_states_.size();
  if (n_imu > cap_imu_) {
    if (d_imu_) WC_CALL(wc_dev_free(ctx_, d_imu_));
    cap_imu_ = n_imu * 2;
    void *p = nullptr;
    WC_CALL(wc_dev_alloc(ctx_, cap_imu_ * sizeof(wc_imu_state), &p));
    d_imu_ = (wc_imu_state *)p;
    imu_dev_stale_ = true;
  }
  if (!imu_dev_stale_) return;
  const std::vector<wc_imu_state> &flat = FlatImu();
  WC_CALL(wc_h2d(ctx_, d_imu_, flat.data(), n_imu * sizeof(wc_imu_state)));
  imu_dev_stale_ = false;
}

const std::vector<wc_imu_state> &LidarOdometry::FlatImu() {
  if (imu_flat_stale_) {
    imu_flat_.assign(imu_states_.begin(), imu_states_.end());
    imu_flat_stale_ = false;
  }
  return imu_flat_;
}

void LidarOdometry::UpdateSurfelPosesOnDevice() {  // UpdateSurfelPoses (:160-170) over the sliding window
  UploadImuStates();
  const size_t n = n_surfels_ - sld_begin_;
  if (n)
    WC_CALL(wc_update_surfel_poses(ctx_, d_imu_, imu_states_.size(), d_surf_ + sld_begin_, d_pose_ + sld_begin_, d_inbody_ + sld_begin_, n));
}

// UpdateImuPoses (:187-215) with the CubicBSplineSampleCorrector (:22-54)
void LidarOdometry::UpdateImuPoses() {
  TouchImu();
  std::vector<double> ts;
  std::vector<V3> rc, pc;
  for (const Sample &s : samples_) {
    ts.push_back(s.timestamp);
    rc.push_back(v3(s.cor));
    pc.push_back(v3(s.cor + 3));
  }
  CubicBSpline rot_interp(ts, rc), pos_interp(ts, pc);
  long first = -1, last = -1;
  for (size_t i = 0; i < imu_states_.size(); ++i) {
    V3 r{0, 0, 0}, p{0, 0, 0};
    const bool ok = rot_interp.Interp(imu_states_[i].t, r);
    const bool ok2 = pos_interp.Interp(imu_states_[i].t, p);
    WC_CHECK(ok == ok2);
    if (!ok) continue;
    stq(imu_states_[i].quat, qmul(so3_exp(r), q4(imu_states_[i].quat)));
    st3(imu_states_[i].pos, p + v3(imu_states_[i].pos));
    if (first < 0) first = (long)i;
    last = (long)i;
  }
  if (first != -1) {
    WC_CHECK(first == 0);
    WC_CHECK(last == (long)imu_states_.size() - 2);
    const size_t n = imu_states_.size();
    const Sample &b = samples_.back();
    PredictPoseOfNewImuState(imu_states_[n - 3], imu_states_[n - 2], v3(b.cor + 9), v3(b.cor + 6), v3(b.grav), imu_states_[n - 1]);
  }
}

void LidarOdometry::UpdateSamplePoses() {  // :172-179
  for (Sample &s : samples_) {
    stq(s.quat, qmul(so3_exp(v3(s.cor)), q4(s.quat)));
    st3(s.pos, v3(s.cor + 3) + v3(s.pos));
    for (int d = 0; d < 6; ++d) s.cor[d] = 0;
  }
}

// ShrinkToFit (:228-250).  The oldest sliding-window surfels move to the FRONT of the fixed window, oldest first
// (push_front, :243-246), so the fixed window is newest-first: one reversed device copy per sweep.  Like the reference
// (Q11: :247-249 compares back() with itself) the fixed window is never trimmed while reference_quirks is set; without the
// quirk it is cut to fixed_window_duration from its old end.
void LidarOdometry::ShrinkToFit() {
  if (samples_.empty() || samples_.back().timestamp - samples_.front().timestamp <= config_.sliding_window_duration) return;
  while (samples_.back().timestamp - samples_.front().timestamp > config_.sliding_window_duration) samples_.pop_front();
  while (imu_states_.front().t < samples_.front().timestamp) imu_states_.pop_front();
  TouchImu();
  size_t k = 0;
  while (k < surfel_times_.size() && surfel_times_[k] < imu_states_.front().t) ++k;
  if (k) {
    EnsureFixedCapacity(k);
    WC_CALL(wc_reverse_copy_surfels(ctx_, d_surf_ + sld_begin_, d_pose_ + sld_begin_, k, d_fix_surf_ + (fix_start_ - k), d_fix_pose_ + (fix_start_ - k)));
    WC_CALL(wc_sync(ctx_));
    fix_start_ -= k;
    for (size_t i = 0; i < k; ++i) fix_times_.push_front(surfel_times_[i]);
    surfel_times_.erase(surfel_times_.begin(), surfel_times_.begin() + (long)k);
    sld_begin_ += k;
  }
  if (!config_.reference_quirks)
    while (!fix_times_.empty() && fix_times_.front() - fix_times_.back() > config_.fixed_window_duration) {
      fix_times_.pop_back();
      --fix_end_;
    }
}

void LidarOdometry::AddLidarScan(const pcl::PointCloud<hilti_ros::Point>::Ptr &msg) {
  // lidar frame -> imu frame, range / blind-box filter (:489-496): on the device, the points stay there
  const auto t_entry = std::chrono::steady_clock::now();
  AppendScanOnDevice(*msg);
  if (!SyncHeadingMsgs()) return;

  // 1. collect scan to sweep (:501-509)
  double sweep_endtime = point_times_.front() + config_.sweep_duration;
  if (point_times_.back() < sweep_endtime || imu_buff_.empty() || imu_buff_.back().timestamp < sweep_endtime) return;

  // wall time of the stages of a completed sweep (last_stage_ms(); WC_ODOM_DEBUG=1 also prints them on stderr)
  static const bool dbg_t = getenv("WC_ODOM_DEBUG") != nullptr;  // (prints only; the one variable the facade reads)
  auto t_prev = std::chrono::steady_clock::now();
  last_append_ms_ = std::chrono::duration<double, std::milli>(t_prev - t_entry).count();
  double (&t_stage)[8] = last_stage_ms_;
  for (double &v : t_stage) v = 0.0;
  last_lm_iterations_ = 0;
  auto lap = [&](int i) {
    const auto now = std::chrono::steady_clock::now();
    t_stage[i] += std::chrono::duration<double, std::milli>(now - t_prev).count();
    t_prev = now;
  };
  // 2. integrate IMU poses in windows (:512-513)
  PredictImuStatesAndSampleStates(sweep_endtime);
  sweep_endtime = samples_.back().timestamp;
  size_t n_sweep = 0;  // BuildSweep (:134-141): the leading points with time < sweep_endtime
  // (the stamps ascend - the CHECK of :491 holds on the device, AppendScanOnDevice -, so the first one that is not earlier is found by
  // bisection; counted one by one through the deque, 300 k stamps were 0.25 ms of every completed sweep)
  n_sweep = (size_t)(std::lower_bound(point_times_.begin(), point_times_.end(), sweep_endtime) - point_times_.begin());
  WC_CHECK(n_sweep > 0);
  const double sweep_t0 = point_times_.front(), sweep_t1 = point_times_[n_sweep - 1];

  // 3. undistort sweep by IMU poses (:519-520) — wc_undistort_sweep_packed replaces UndistortSweep :143-158 and leaves the
  //    sweep as the 20 bytes per point BuildSurfels reads (x, y, z, time: surfel_extraction.cc:317-324); neither the raw nor the
  //    undistorted sweep ever comes back to the host, and the undistorted 48-byte records are never formed
  if (n_sweep > cap_sweep_) {
    if (d_sweep_xyz_) WC_CALL(wc_dev_free(ctx_, d_sweep_xyz_));
    if (d_sweep_t_) WC_CALL(wc_dev_free(ctx_, d_sweep_t_));
    cap_sweep_ = n_sweep * 2;
    WC_CALL(wc_dev_alloc(ctx_, cap_sweep_ * 3 * sizeof(float), &d_sweep_xyz_));
    WC_CALL(wc_dev_alloc(ctx_, cap_sweep_ * sizeof(double), &d_sweep_t_));
  }
  UploadImuStates();
  WC_CALL(wc_undistort_sweep_packed(ctx_, (const char *)d_pts_[pts_cur_] + pts_begin_ * sizeof(hilti_ros::Point), n_sweep, d_imu_, imu_states_.size(),
                                    (float *)d_sweep_xyz_, (double *)d_sweep_t_));
  const void *d_raw_sweep = (const char *)d_pts_[pts_cur_] + pts_begin_ * sizeof(hilti_ros::Point);  // (stays where it is until the next scan arrives)
  lap(0);

  // 4. ---- hot path: extract surfels, attach poses (:523-527) ----
  const size_t max_new = (3 * n_sweep) / 20 + 1;
  EnsureSurfelCapacity(max_new);
  wc_points desc{d_sweep_xyz_, d_sweep_t_, 3 * sizeof(float), sizeof(double), n_sweep};
  uint64_t n_new = 0;
  // (the sweep's stamps leave the host's buffer - 300 k deque entries, ~0.1 ms - while the extraction runs)
  WC_CALL(wc_extract_surfels_enqueue(ctx_, &desc, sweep_t0, sweep_t1, d_surf_ + n_surfels_, nullptr, max_new));
  DropBufferedPoints(n_sweep);
  WC_CALL(wc_extract_surfels_finish(ctx_, &n_new));
  {
    uint32_t st[64];
    WC_CALL(wc_debug_status(ctx_, st));
    ++(st[61] ? sweeps_fast_ : sweeps_exact_);  // which arithmetic completed this sweep (read-out for tests / logs)
  }
  if (n_new) {
    WC_CALL(wc_memset(ctx_, d_inbody_ + n_surfels_, 0, n_new));
    std::vector<double> fresh(n_new);  // only the timestamps travel (8 of 144 bytes per surfel)
    WC_CALL(wc_d2h_strided(ctx_, fresh.data(), d_surf_ + n_surfels_, 8, sizeof(wc_surfel), n_new));
    surfel_times_.insert(surfel_times_.end(), fresh.begin(), fresh.end());
    n_surfels_ += n_new;
  }
  UpdateSurfelPosesOnDevice();
  lap(1);

  for (int iter = 0; iter < config_.outer_iter_num_max; ++iter) {
    const size_t n_sld = n_surfels_ - sld_begin_;
    // correspondences (:530-538)
    uint64_t n_b = 0, n_u = 0;
    const size_t n_fix = fix_end_ - fix_start_;
    WC_CALL(wc_match_pair(ctx_, d_surf_ + sld_begin_, d_pose_ + sld_begin_, n_sld, d_fix_surf_ + fix_start_, d_fix_pose_ + fix_start_, n_fix,
                          d_pairs_sld_, cap_surfels_, &n_b, d_pairs_fix_, cap_surfels_, &n_u));
    last_corr_[0] = n_b, last_corr_[1] = n_u;
    if (keep_pair_stamps_) {  // (test hook: outside the stage clocks' interest, a few strided read-backs)
      std::vector<double> ts_sld(n_sld), ts_fix(n_fix);
      if (n_sld) WC_CALL(wc_d2h_strided(ctx_, ts_sld.data(), d_surf_ + sld_begin_, 8, sizeof(wc_surfel), n_sld));
      if (n_fix) WC_CALL(wc_d2h_strided(ctx_, ts_fix.data(), d_fix_surf_ + fix_start_, 8, sizeof(wc_surfel), n_fix));
      for (int which = 0; which < 2; ++which) {
        const uint64_t n = which ? n_u : n_b;
        std::vector<wc_pair> pr(n);
        if (n) WC_CALL(wc_d2h(ctx_, pr.data(), which ? d_pairs_fix_ : d_pairs_sld_, n * sizeof(wc_pair)));
        pair_stamps_[which].clear();
        for (const wc_pair &q : pr) {
          pair_stamps_[which].push_back(which ? ts_fix[(size_t)q.first] : ts_sld[(size_t)q.first]);  // (fixed window: first = its surfel)
          pair_stamps_[which].push_back(ts_sld[(size_t)q.second]);
        }
      }
    }
    lap(2);
    // 5. solve poses in windows (:541-562)
    std::vector<double> ts, x;
    for (const Sample &s : samples_) {
      ts.push_back(s.timestamp);
      x.insert(x.end(), s.cor, s.cor + 12);
    }
    const std::vector<wc_imu_state> &flat = FlatImu();
    const bool fix_first = first_sample_known_ && samples_.front().timestamp == first_sample_time_;  // :556-560
    WC_CALL(wc_window_build(ctx_, d_surf_ + sld_begin_, d_pose_ + sld_begin_, d_pairs_sld_, n_b, d_fix_surf_ + fix_start_, d_fix_pose_ + fix_start_, d_pairs_fix_, n_u,
                            flat.data(), flat.size(), ts.data(), ts.size(), samples_.back().grav, fix_first ? 1 : 0));
    lap(3);
    if (config_.log_residual_histograms) {  // :547-549
      residual_log_.clear();
      LogResiduals(x, "[before solve] ");
    }
    WC_CALL(wc_window_solve(ctx_, x.data(), &last_summary_, nullptr));
    last_lm_iterations_ += (int)last_summary_.iterations;
    lap(4);
    for (size_t i = 0; i < samples_.size(); ++i) std::memcpy(samples_[i].cor, &x[12 * i], 96);
    // state update (:564-566)
    UpdateImuPoses();
    UpdateSurfelPosesOnDevice();
    UpdateSamplePoses();
    if (config_.log_residual_histograms) {
      // :568-570 - AFTER UpdateSamplePoses: the factors still hold the poses they were built with, the sample states' rotation
      // and position corrections have just been zeroed (:176-177), the biases stay: the reference's second log shows the
      // residuals at (0, 0, bg, ba), not at the optimum.  Reproduced as it is.
      std::vector<double> x_after;
      for (const Sample &s : samples_) x_after.insert(x_after.end(), s.cor, s.cor + 12);
      LogResiduals(x_after, "[after update] ");
      if (dbg_t) fputs(residual_log_.c_str(), stderr);
    }
    lap(5);
  }
  ShrinkToFit();  // :574-580
  if (config_.fill_outputs) FillOutputs(d_raw_sweep, n_sweep);  // :582-602
  lap(6);
  if (dbg_t)
    fprintf(stderr, "[odom] sweep %d: predict + undistort %.2f, extract + poses %.2f, match %.2f, build %.2f, solve %.2f (%d iterations), update %.2f, shrink %.2f ms\n",
            (int)sweep_id_, t_stage[0], t_stage[1], t_stage[2], t_stage[3], t_stage[4], (int)last_summary_.iterations, t_stage[5], t_stage[6]);
  ++sweep_id_;
}
