// oracle/basics.cc — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// Default parameters, known-answer hooks for the SO(3) helpers, and the surfel pose update:
//   UpdateSurfelPoses   src/odometry/lidar_odometry.cc:160-170
//   Surfel::UpdatePose  src/odometry/surfel.h:48-58
#include <algorithm>
#include <cmath>
#include <cstring>

#include "math3.h"
#include "wc_oracle.h"

using namespace wco;

extern "C" void wco_params_default(wc_params *p) {
  // surfel_extraction.cc:327 — BuildVoxelMap(points, Zero, 0.8, 2, {20,20,20,20}, 0.01, 0.1, ...)
  p->voxel_size = 0.8f;
  p->max_layer = 2;
  p->min_points = 20;
  p->planer_threshold = 0.01f;
  p->min_plane_likeness = 0.1;
  p->view_point[0] = p->view_point[1] = p->view_point[2] = 0.0;
  p->cluster_gap = 0.05;       // surfel_extraction.cc:24
  p->cluster_min_points = 20;  // surfel_extraction.cc:33
  // knn_surfel_matcher.h:37-41
  p->center_scale = 1.0;
  p->angular_scale = 5.0 * M_PI / 180.0;
  p->surfel_dist_max = 0.1;
  p->knn_k = 10;
  p->time_diff_min = 0.06;
  // cost_functor.h:24, lidar_odometry.cc:270
  p->surfel_sigma0 = 0.05 / 6;
  p->cauchy_a = 0.4;
  // lio_config.h:10-14,32,42-45
  const double gn = 0.00015198973532354657, an = 0.006308226052016165;
  const double gw = 0.00011673723527962174, aw = 2.664506559330434e-06;
  const double rate = 200, k = 0.01;
  p->w_gyr = 1 / (gn * std::sqrt(rate)) * k;
  p->w_acc = 1 / (an * std::sqrt(rate)) * k;
  p->w_bg = 1 / (gw / std::sqrt(rate)) * k;
  p->w_ba = 1 / (aw / std::sqrt(rate)) * k;
  p->imu_dt = 1 / rate;
  p->max_iterations = 100;  // lio_config.h:41
  p->reference_quirks = 1;
  p->exact_sums = 0;  // (the oracle always sums in the reference's order; the field only steers the HIP library)
}

static void store(const M3 &m, double out9[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) out9[3 * i + j] = m.m[i][j];
}
extern "C" void wco_so3_exp(const double w[3], double q[4]) {
  Q4 r = so3_exp({w[0], w[1], w[2]});
  q[0] = r.w, q[1] = r.x, q[2] = r.y, q[3] = r.z;
}
extern "C" void wco_so3_log(const double q[4], double w[3]) {
  V3 r = so3_log({q[0], q[1], q[2], q[3]});
  w[0] = r.x, w[1] = r.y, w[2] = r.z;
}
extern "C" void wco_so3_jl(const double v[3], double o[9]) { store(so3_Jl({v[0], v[1], v[2]}), o); }
extern "C" void wco_so3_jl_inv(const double v[3], double o[9]) { store(so3_Jl_inv({v[0], v[1], v[2]}), o); }
extern "C" void wco_so3_jr(const double v[3], double o[9]) { store(so3_Jr({v[0], v[1], v[2]}), o); }
extern "C" void wco_so3_jr_inv(const double v[3], double o[9]) { store(so3_Jr_inv({v[0], v[1], v[2]}), o); }
extern "C" void wco_eig3(const double a9[9], double evals[3], double evecs9[9]) {
  M3 a, v;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a.m[i][j] = a9[3 * i + j];
  eig3_sym(a, evals, v);
  store(v, evecs9);
}

extern "C" int wco_update_surfel_poses(const wc_imu_state *imu, uint64_t n_imu, wc_surfel *surf, wc_pose *pose,
                                       uint8_t *in_body, uint64_t n) {
  for (uint64_t s = 0; s < n; ++s) {
    const double t = surf[s].t;
    // std::lower_bound on imu timestamps (lidar_odometry.cc:162)
    uint64_t lo = 0, hi = n_imu;
    while (lo < hi) {
      uint64_t mid = (lo + hi) / 2;
      if (imu[mid].t < t)
        lo = mid + 1;
      else
        hi = mid;
    }
    const uint64_t idx = lo;
    if (idx == 0 || idx == n_imu) return 2;  // CHECK at lidar_odometry.cc:164
    const wc_imu_state &a = imu[idx - 1], &b = imu[idx];
    const double f = (t - a.t) / (b.t - a.t);
    V3 pos = V3{a.pos[0], a.pos[1], a.pos[2]} * (1 - f) + V3{b.pos[0], b.pos[1], b.pos[2]} * f;
    Q4 rot = qslerp({a.quat[0], a.quat[1], a.quat[2], a.quat[3]}, f, {b.quat[0], b.quat[1], b.quat[2], b.quat[3]});
    pose[s].pos[0] = pos.x, pose[s].pos[1] = pos.y, pose[s].pos[2] = pos.z;
    pose[s].quat[0] = rot.w, pose[s].quat[1] = rot.x, pose[s].quat[2] = rot.y, pose[s].quat[3] = rot.z;
    if (!in_body[s]) {  // surfel.h:52-57: world -> body on the first update
      in_body[s] = 1;
      Q4 rc = qconj(rot);
      V3 c = qrot(rc, V3{surf[s].center[0], surf[s].center[1], surf[s].center[2]} - pos);
      V3 nn = qrot(rc, V3{surf[s].normal[0], surf[s].normal[1], surf[s].normal[2]});
      M3 C;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C.m[i][j] = surf[s].cov[3 * i + j];
      M3 R = qmat(rot);
      M3 Cb = (transpose(R) * C) * R;  // rot.conjugate() * covariance * rot
      for (int i = 0; i < 3; ++i) {
        surf[s].center[i] = c[i];
        surf[s].normal[i] = nn[i];
        for (int j = 0; j < 3; ++j) surf[s].cov[3 * i + j] = Cb.m[i][j];
      }
    }
  }
  return 0;
}

// ---- "next" row f-1: point pre-filter (lidar_odometry.cc:489-496) and UndistortSweep (lidar_odometry.cc:143-158) --------
// records are the reference's 48-byte hilti_ros::Point (x,y,z float @0, time double @24)
static inline float *pt_xyz(void *base, uint64_t i) { return (float *)((char *)base + 48 * i); }
static inline const float *pt_xyz(const void *base, uint64_t i) { return (const float *)((const char *)base + 48 * i); }
static inline double pt_time(const void *base, uint64_t i) {
  double t;
  std::memcpy(&t, (const char *)base + 48 * i + 24, 8);
  return t;
}

extern "C" int wco_prefilter_points(const void *pts_in, uint64_t n, const double ext_quat[4], const double ext_t[3], double min_range,
                                    double max_range, const double blind_min[3], const double blind_max[3], void *pts_out,
                                    uint64_t *n_out) {
  const Q4 q{ext_quat[0], ext_quat[1], ext_quat[2], ext_quat[3]};
  const V3 t{ext_t[0], ext_t[1], ext_t[2]};
  uint64_t o = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const float *f = pt_xyz(pts_in, i);
    // pt = (ext_lidar2imu * pt.cast<double>()).cast<float>()   (cc:490)
    const V3 p = qrot(q, V3{(double)f[0], (double)f[1], (double)f[2]}) + t;
    const float x = (float)p.x, y = (float)p.y, z = (float)p.z;
    const float nrm = std::sqrt(x * x + y * y + z * z);  // Eigen float norm
    const bool blind = (double)x >= blind_min[0] && (double)x <= blind_max[0] && (double)y >= blind_min[1] &&
                       (double)y <= blind_max[1] && (double)z >= blind_min[2] && (double)z <= blind_max[2];
    if (nrm < min_range || nrm > max_range || blind) continue;  // cc:492-494
    std::memcpy((char *)pts_out + 48 * o, (const char *)pts_in + 48 * i, 48);
    float *g = pt_xyz(pts_out, o);
    g[0] = x, g[1] = y, g[2] = z;
    ++o;
  }
  *n_out = o;
  return 0;
}

extern "C" int wco_undistort_sweep(const void *pts_in, uint64_t n, const wc_imu_state *imu, uint64_t n_imu, void *pts_out) {
  for (uint64_t i = 0; i < n; ++i) {
    const double t = pt_time(pts_in, i);
    uint64_t lo = 0, hi = n_imu;  // std::lower_bound (cc:147)
    while (lo < hi) {
      const uint64_t mid = (lo + hi) / 2;
      if (imu[mid].t < t)
        lo = mid + 1;
      else
        hi = mid;
    }
    if (!(lo >= 1 && lo < n_imu)) return 2;  // CHECK(idx >= 1 && idx < size) (cc:149)
    const wc_imu_state &a = imu[lo - 1], &b = imu[lo];
    const double f = (t - a.t) / (b.t - a.t);
    const V3 pos = V3{a.pos[0], a.pos[1], a.pos[2]} * (1 - f) + V3{b.pos[0], b.pos[1], b.pos[2]} * f;
    const Q4 rot = qslerp({a.quat[0], a.quat[1], a.quat[2], a.quat[3]}, f, {b.quat[0], b.quat[1], b.quat[2], b.quat[3]});
    const float *fin = pt_xyz(pts_in, i);
    const V3 w = qrot(rot, V3{(double)fin[0], (double)fin[1], (double)fin[2]}) + pos;
    std::memcpy((char *)pts_out + 48 * i, (const char *)pts_in + 48 * i, 48);
    float *g = pt_xyz(pts_out, i);
    g[0] = (float)w.x, g[1] = (float)w.y, g[2] = (float)w.z;
  }
  return 0;
}
