// odom_c_api.cc — a flat C wrapper around the LidarOdometry facade so that scripts (tests, bench) can drive the same
// object wildcat_slam_node.cc would drive through its C++ interface.
#include <cstring>

#include <algorithm>

#include "cubic_bspline.h"
#include "histogram.h"
#include "imu_resampler.h"
#include "lidar_odometry.h"

extern "C" {

void *wc_odom_create(int device) { return new LidarOdometry(device); }
void wc_odom_destroy(void *h) { delete (LidarOdometry *)h; }

void wc_odom_add_imu(void *h, double t, const double acc[3], const double gyr[3]) {
  ImuData d;
  d.timestamp = t;
  for (int i = 0; i < 3; ++i) d.linear_acceleration[i] = acc[i], d.angular_velocity[i] = gyr[i];
  ((LidarOdometry *)h)->AddImuData(d);
}

// points: n records of the 48-byte hilti_ros::Point layout, in the LIDAR frame, time ascending
void wc_odom_add_scan(void *h, const void *points, uint64_t n) {
  auto cloud = std::make_shared<pcl::PointCloud<hilti_ros::Point>>();
  cloud->points.resize(n);
  if (n) std::memcpy(cloud->points.data(), points, n * sizeof(hilti_ros::Point));
  ((LidarOdometry *)h)->AddLidarScan(cloud);
}

int wc_odom_sweeps(void *h) { return ((LidarOdometry *)h)->sweeps_done(); }
uint64_t wc_odom_num_samples(void *h) { return ((LidarOdometry *)h)->num_sample_states(); }

// out[15] = t, pos[3], quat[4] (w,x,y,z), bg[3], ba[3], spare
int wc_odom_sample(void *h, uint64_t i, double *out) {
  LidarOdometry::SampleStateView v;
  if (!((LidarOdometry *)h)->sample_state(i, &v)) return 1;
  out[0] = v.timestamp;
  std::memcpy(out + 1, v.pos, 24);
  std::memcpy(out + 4, v.quat, 32);
  std::memcpy(out + 8, v.bg, 24);
  std::memcpy(out + 11, v.ba, 24);
  return 0;
}

// stats[8] = sliding surfels, fixed surfels, binary corr, unary corr, LM iterations, initial cost, final cost, termination
void wc_odom_stats(void *h, double *stats) {
  LidarOdometry *o = (LidarOdometry *)h;
  stats[0] = (double)o->sliding_window_surfels();
  stats[1] = (double)o->fixed_window_surfels();
  stats[2] = (double)o->last_correspondences(0);
  stats[3] = (double)o->last_correspondences(1);
  stats[4] = o->last_solve().iterations;
  stats[5] = o->last_solve().initial_cost;
  stats[6] = o->last_solve().final_cost;
  stats[7] = o->last_solve().termination;
}

// out[2] = sweeps extracted by the default (integer-moment) arithmetic, sweeps extracted in the reference's summation order
void wc_odom_extract_paths(void *h, int out[2]) {
  out[0] = ((LidarOdometry *)h)->sweeps_fast_path();
  out[1] = ((LidarOdometry *)h)->sweeps_exact_path();
}

// timestamps of the fixed window in its stored order (newest first, Q11); returns the window size
uint64_t wc_odom_fixed_times(void *h, double *out, uint64_t cap) {
  const std::deque<double> &t = ((LidarOdometry *)h)->fixed_window_times();
  for (uint64_t i = 0; i < t.size() && i < cap; ++i) out[i] = t[i];
  return t.size();
}
// both setters re-derive the device context's parameters (Q1 / Q3 Jacobians, extraction arithmetic), not only the host flags
void wc_odom_set_quirks(void *h, int on) {
  ((LidarOdometry *)h)->config().reference_quirks = on != 0;
  ((LidarOdometry *)h)->ApplyConfig();
}
void wc_odom_set_exact_sums(void *h, int on) {
  ((LidarOdometry *)h)->config().exact_sums = on != 0;
  ((LidarOdometry *)h)->ApplyConfig();
}
// the reference's residual log (PrintSurfelResiduals / PrintImuResiduals, lidar_odometry.cc:56-94): switch + the last sweep's text
void wc_odom_set_residual_log(void *h, int on) { ((LidarOdometry *)h)->config().log_residual_histograms = on != 0; }
uint64_t wc_odom_residual_log(void *h, char *out, uint64_t cap) {
  const std::string &s = ((LidarOdometry *)h)->last_residual_log();
  if (out && cap) {
    const size_t n = std::min<size_t>(s.size(), cap - 1);
    std::memcpy(out, s.data(), n);
    out[n] = 0;
  }
  return s.size();
}
// Histogram::ToString (src/common/histogram.cc:27-76) through the facade's restatement, for a known-answer test
uint64_t wc_host_histogram(const double *values, uint64_t n, int buckets, char *out, uint64_t cap) {
  Histogram hst;
  for (uint64_t i = 0; i < n; ++i) hst.Add(values[i]);
  const std::string s = hst.ToString(buckets);
  if (out && cap) {
    const size_t m = std::min<size_t>(s.size(), cap - 1);
    std::memcpy(out, s.data(), m);
    out[m] = 0;
  }
  return s.size();
}

// test hook: start the next sweep from another run's states (LidarOdometry::ImportState); 0 = done, 1 = the counts differ
int wc_odom_import_state(void *h, const double *samples23, uint64_t ns, const wc_imu_state *imu, uint64_t n_imu) {
  return ((LidarOdometry *)h)->ImportState(samples23, ns, imu, n_imu) ? 0 : 1;
}

// ---- known-answer hooks for the host-side product code (the g++ instantiation of csrc/dmath.h, the facade's spline, the
// resampler): the reference's own unit tests are run against these in tests/test_host_kat.py -------------------------------

// utils_test.cc:5-21 inputs -> out52 = Exp(v) | Log(Exp(v)) | Jl | Jl_inv | Jr | Jr_inv | Hat (layout of wc_selftest_so3)
void wc_host_so3(const double v3[3], double out52[52]) {
  using namespace wc;
  const V3 v = mk3(v3[0], v3[1], v3[2]);
  const Q4 q = so3_exp(v);
  const V3 l = so3_log(q);
  double *o = out52;
  o[0] = q.w, o[1] = q.x, o[2] = q.y, o[3] = q.z, o[4] = l.x, o[5] = l.y, o[6] = l.z;
  const M3 ms[5] = {so3_Jl(v), so3_Jl_inv(v), so3_Jr(v), so3_Jr_inv(v), hat(v)};
  for (int m = 0; m < 5; ++m)
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) o[7 + 9 * m + 3 * i + j] = ms[m].m[i][j];
}

// CubicBSplineInterpolator(timestamps, points).Interp(t) (spline_interpolation.h:42-113) through the facade's class.
// ctrl3 (may be NULL) receives the np control points.
void wc_host_bspline_fit_eval(const double *ts, const double *pts3, uint64_t np, const double *query, uint64_t nq, double *out3,
                              uint8_t *valid, double *ctrl3) {
  using namespace wc;
  std::vector<double> t(ts, ts + np);
  std::vector<V3> p(np);
  for (uint64_t i = 0; i < np; ++i) p[i] = mk3(pts3[3 * i], pts3[3 * i + 1], pts3[3 * i + 2]);
  const CubicBSpline sp(t, p);
  for (uint64_t i = 0; i < nq; ++i) {
    V3 o = mk3(0, 0, 0);
    valid[i] = sp.Interp(query[i], o) ? 1 : 0;
    out3[3 * i] = o.x, out3[3 * i + 1] = o.y, out3[3 * i + 2] = o.z;
  }
  if (ctrl3)
    for (uint64_t i = 0; i < np; ++i)
      ctrl3[3 * i] = sp.control_points()[i].x, ctrl3[3 * i + 1] = sp.control_points()[i].y, ctrl3[3 * i + 2] = sp.control_points()[i].z;
}

void *wc_host_resampler_create(int freq) { return new ImuResampler(freq); }
void wc_host_resampler_destroy(void *h) { delete (ImuResampler *)h; }
void wc_host_resampler_add(void *h, double t, const double acc[3], const double gyr[3]) {
  ImuData d;
  d.timestamp = t;
  for (int i = 0; i < 3; ++i) d.linear_acceleration[i] = acc[i], d.angular_velocity[i] = gyr[i];
  ((ImuResampler *)h)->AddImuData(d);
}
// out7 = t, acc, gyr; returns 1 when a resampled measurement was due (the shared_ptr was non-null)
int wc_host_resampler_advance(void *h, double out7[7]) {
  const std::shared_ptr<ImuData> r = ((ImuResampler *)h)->AdvanceGetResampledImuData();
  if (!r) return 0;
  out7[0] = r->timestamp;
  for (int i = 0; i < 3; ++i) out7[1 + i] = r->linear_acceleration[i], out7[4 + i] = r->angular_velocity[i];
  return 1;
}
}
