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
// a development option of the facade's library context (include/wildcat_hip.h: wc_ctx_set_dev_option) - stress scripts only
int wc_odom_set_dev_option(void *h, const char *name, int value) { return wc_ctx_set_dev_option(((LidarOdometry *)h)->gpu_context(), name, value); }

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

// wall time [ms] of the last completed sweep's stages (predict + undistort, extract + poses, match, build, solve, update, shrink)
// and, in out[7], the LM iterations of its outer iterations together
void wc_odom_stage_ms(void *h, double out[8]) {
  const LidarOdometry *o = (const LidarOdometry *)h;
  for (int i = 0; i < 7; ++i) out[i] = o->last_stage_ms()[i];
  out[7] = (double)o->last_lm_iterations();
}

// wall time [ms] of the completing message's part in front of those stages (upload + pre-filter of its points, heading synchronisation)
double wc_odom_append_ms(void *h) { return ((const LidarOdometry *)h)->last_append_ms(); }

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
// test hook: the surfel timestamps (first, second) of the last sweep's correspondences; which = 0 sliding, 1 fixed window
void wc_odom_set_keep_pair_stamps(void *h, int on) { ((LidarOdometry *)h)->set_keep_pair_stamps(on != 0); }
uint64_t wc_odom_pair_stamps(void *h, int which, double *out, uint64_t cap) {
  const std::vector<double> &v = ((LidarOdometry *)h)->last_pair_stamps(which ? 1 : 0);
  for (uint64_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
  return v.size();
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

// ---- the ROS-free wire formats (host/wire_formats.h) ------------------------------------------------------------------------
void wc_odom_set_fill_outputs(void *h, int on) { ((LidarOdometry *)h)->config().fill_outputs = on != 0; }
// markers of the last sweep: out = n x 14 doubles (position 3, orientation wxyz 4, scale 3, colour rgba 4); returns n
uint64_t wc_odom_markers(void *h, double *out, uint64_t cap) {
  const auto &m = ((LidarOdometry *)h)->last_outputs().markers;
  for (uint64_t i = 0; i < m.size() && i < cap; ++i) {
    double *o = out + 14 * i;
    std::memcpy(o, m[i].position, 24);
    std::memcpy(o + 3, m[i].orientation, 32);
    std::memcpy(o + 7, m[i].scale, 24);
    for (int c = 0; c < 4; ++c) o[10 + c] = m[i].color[c];
  }
  return m.size();
}
// the published sweep: its 48-byte point records (the PointCloud2 payload), header stamp; tf8 = stamp, origin, rotation xyzw
uint64_t wc_odom_scan_in_world(void *h, void *points48, uint64_t cap, double *stamp, double tf8[8]) {
  const LidarOdometry::SweepOutputs &o = ((LidarOdometry *)h)->last_outputs();
  const uint64_t n = o.scan_in_world.width;
  if (points48 && cap >= n && n) std::memcpy(points48, o.scan_in_world.data.data(), 48 * n);
  if (stamp) *stamp = o.scan_stamp;
  if (tf8) {
    tf8[0] = o.tf.stamp;
    std::memcpy(tf8 + 1, o.tf.origin, 24);
    std::memcpy(tf8 + 4, o.tf.rotation_xyzw, 32);
  }
  return n;
}
// pcl::fromROSMsg for hilti_ros::Point on a described payload: nf fields (names: zero-terminated strings back to back),
// returns the number of registered fields matched (-1: refused); out48 must hold width * height records
int wc_host_cloud2_to_points(const char *names, const uint32_t *offsets, const uint8_t *datatypes, const uint32_t *counts, int nf, uint32_t width,
                             uint32_t height, uint32_t point_step, uint32_t row_step, int is_bigendian, const uint8_t *data, uint64_t nbytes,
                             void *out48) {
  wc_wire::PointCloud2 msg;
  msg.width = width, msg.height = height, msg.point_step = point_step, msg.row_step = row_step, msg.is_bigendian = is_bigendian != 0;
  const char *p = names;
  for (int f = 0; f < nf; ++f) {
    msg.fields.push_back({std::string(p), offsets[f], datatypes[f], counts[f]});
    p += std::strlen(p) + 1;
  }
  msg.data.assign(data, data + nbytes);
  std::vector<hilti_ros::Point> pts;
  const int m = wc_wire::PointsFromCloud2(msg, pts);
  if (m >= 0 && !pts.empty()) std::memcpy(out48, pts.data(), 48 * pts.size());
  return m;
}
// pcl::toROSMsg: the field table (6 x {offset, datatype, count}) and point_step for hilti_ros::Point; names_out gets
// "x\0y\0z\0intensity\0timestamp\0ring\0"
int wc_host_points_to_cloud2_layout(uint32_t table18[18], char *names_out, uint64_t cap) {
  wc_wire::PointCloud2 msg;
  wc_wire::Cloud2FromPoints(nullptr, 0, msg);
  std::string names;
  for (size_t f = 0; f < msg.fields.size(); ++f) {
    table18[3 * f] = msg.fields[f].offset, table18[3 * f + 1] = msg.fields[f].datatype, table18[3 * f + 2] = msg.fields[f].count;
    names += msg.fields[f].name;
    names.push_back('\0');
  }
  if (names_out && cap >= names.size()) std::memcpy(names_out, names.data(), names.size());
  return (int)msg.point_step;
}
// makeRightHanded (surfel_extraction.cc:340-358) and the marker of one surfel, for known-answer tests
void wc_host_make_right_handed(double evec9[9], double eval3[3]) { wc_wire::MakeRightHanded(evec9, eval3); }
void wc_host_marker(const wc_surfel *s, const wc_pose *p, double out14[14]) {
  const wc_wire::SurfelMarker m = wc_wire::MarkerFromSurfel(*s, *p);
  std::memcpy(out14, m.position, 24);
  std::memcpy(out14 + 3, m.orientation, 32);
  std::memcpy(out14 + 7, m.scale, 24);
  for (int c = 0; c < 4; ++c) out14[10 + c] = m.color[c];
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
