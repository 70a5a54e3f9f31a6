// oracle/spline.cc — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// Post-solve state update of the reference:
//   CubicBSplineInterpolator      src/odometry/spline_interpolation.h:42-113
//   CubicBSplineSampleCorrector   src/odometry/lidar_odometry.cc:22-54
//   PredictPoseOfNewImuState      src/odometry/lidar_odometry.cc:106-123
//   UpdateImuPoses                src/odometry/lidar_odometry.cc:187-215
// Pinned by the reference's own test src/odometry/spline_interpolation_test.cc:79-96 (knot reproduction).
#include <cmath>
#include <cstdint>
#include <vector>

#include "math3.h"
#include "wc_oracle.h"

namespace {
using namespace wco;

struct BSpline {
  std::vector<double> ts;
  int np;
  std::vector<double> Q;  // np x 3 control points
  static constexpr double M[4][4] = {{-1, 3, -3, 1}, {3, -6, 3, 0}, {-3, 0, 3, 0}, {1, 4, 1, 0}};

  // Init(), spline_interpolation.h:75-104: Q = (N^T N)^-1 N^T p with N(i, clamp(i-1..i+2)) += [1 4 1 0]/6
  void fit(const double *timestamps, const double *p3, int n) {
    ts.assign(timestamps, timestamps + n);
    np = n;
    std::vector<double> N((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) {
      double temp[4];
      for (int j = 0; j < 4; ++j) temp[j] = M[3][j] / 6.0;  // tv = (0,0,0,1)
      for (int j = 0; j < 4; ++j) {
        int c = std::min(std::max(i - 1 + j, 0), n - 1);
        N[(size_t)i * n + c] += temp[j];
      }
    }
    // normal equations A = N^T N, B = N^T p, solved by Gaussian elimination with partial pivoting
    std::vector<double> A((size_t)n * n, 0.0), B((size_t)n * 3, 0.0);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        double s = 0;
        for (int k = 0; k < n; ++k) s += N[(size_t)k * n + i] * N[(size_t)k * n + j];
        A[(size_t)i * n + j] = s;
      }
    for (int i = 0; i < n; ++i)
      for (int d = 0; d < 3; ++d) {
        double s = 0;
        for (int k = 0; k < n; ++k) s += N[(size_t)k * n + i] * p3[(size_t)k * 3 + d];
        B[(size_t)i * 3 + d] = s;
      }
    for (int c = 0; c < n; ++c) {
      int piv = c;
      for (int r = c + 1; r < n; ++r)
        if (std::fabs(A[(size_t)r * n + c]) > std::fabs(A[(size_t)piv * n + c])) piv = r;
      if (piv != c) {
        for (int k = 0; k < n; ++k) std::swap(A[(size_t)c * n + k], A[(size_t)piv * n + k]);
        for (int d = 0; d < 3; ++d) std::swap(B[(size_t)c * 3 + d], B[(size_t)piv * 3 + d]);
      }
      for (int r = c + 1; r < n; ++r) {
        double f = A[(size_t)r * n + c] / A[(size_t)c * n + c];
        if (f == 0.0) continue;
        for (int k = c; k < n; ++k) A[(size_t)r * n + k] -= f * A[(size_t)c * n + k];
        for (int d = 0; d < 3; ++d) B[(size_t)r * 3 + d] -= f * B[(size_t)c * 3 + d];
      }
    }
    Q.assign((size_t)n * 3, 0.0);
    for (int r = n - 1; r >= 0; --r)
      for (int d = 0; d < 3; ++d) {
        double s = B[(size_t)r * 3 + d];
        for (int k = r + 1; k < n; ++k) s -= A[(size_t)r * n + k] * Q[(size_t)k * 3 + d];
        Q[(size_t)r * 3 + d] = s / A[(size_t)r * n + r];
      }
  }

  // Interp(), spline_interpolation.h:51-72
  bool interp(double t, double out[3]) const {
    if (t < ts.front() || t > ts.back()) return false;
    double index_f = (t - ts.front()) / (ts.back() - ts.front()) * (np - 1) + 1.0;
    int index_int = (int)std::floor(index_f);
    double u = index_f - index_int;
    double tv[4] = {u * u * u, u * u, u, 1.0};
    double w[4];
    for (int j = 0; j < 4; ++j) {
      w[j] = 0;
      for (int k = 0; k < 4; ++k) w[j] += tv[k] * M[k][j];
    }
    for (int d = 0; d < 3; ++d) out[d] = 0;
    for (int j = 0; j < 4; ++j) {
      int idx = std::min(std::max(index_int - 2 + j, 0), np - 1);
      for (int d = 0; d < 3; ++d) out[d] += w[j] * Q[(size_t)idx * 3 + d];
    }
    for (int d = 0; d < 3; ++d) out[d] /= 6.0;
    return true;
  }
};
constexpr double BSpline::M[4][4];
}  // namespace

extern "C" int wco_bspline_fit_eval(const double *timestamps, const double *points3, uint64_t np,
                                    const double *query_t, uint64_t nq, double *out3, uint8_t *valid) {
  BSpline b;
  b.fit(timestamps, points3, (int)np);
  for (uint64_t i = 0; i < nq; ++i) valid[i] = b.interp(query_t[i], out3 + 3 * i) ? 1 : 0;
  return 0;
}

extern "C" int wco_update_imu_poses(const double *sample_times, const double *x, uint64_t ns, const double ba[3],
                                    const double bg[3], const double grav[3], wc_imu_state *imu, uint64_t n_imu) {
  std::vector<double> rot(ns * 3), pos(ns * 3);
  for (uint64_t i = 0; i < ns; ++i)
    for (int d = 0; d < 3; ++d) {
      rot[i * 3 + d] = x[i * 12 + d];
      pos[i * 3 + d] = x[i * 12 + 3 + d];
    }
  BSpline br, bp;
  br.fit(sample_times, rot.data(), (int)ns);
  bp.fit(sample_times, pos.data(), (int)ns);
  int64_t first = -1, last = -1;
  for (uint64_t i = 0; i < n_imu; ++i) {
    double rc[3], pc[3];
    bool ok = br.interp(imu[i].t, rc);
    bool ok2 = bp.interp(imu[i].t, pc);
    if (ok != ok2) return 4;
    if (!ok) continue;
    Q4 q = qmul(so3_exp({rc[0], rc[1], rc[2]}), {imu[i].quat[0], imu[i].quat[1], imu[i].quat[2], imu[i].quat[3]});
    imu[i].quat[0] = q.w, imu[i].quat[1] = q.x, imu[i].quat[2] = q.y, imu[i].quat[3] = q.z;
    for (int d = 0; d < 3; ++d) imu[i].pos[d] = pc[d] + imu[i].pos[d];
    if (first < 0) first = (int64_t)i;
    last = (int64_t)i;
  }
  if (first != -1) {
    // CHECK_EQ(first, 0); CHECK_EQ(last, size - 2)   (lidar_odometry.cc:209-210)
    if (first != 0 || last != (int64_t)n_imu - 2) return 5;
    const wc_imu_state &i1 = imu[n_imu - 3], &i2 = imu[n_imu - 2];
    wc_imu_state &i3 = imu[n_imu - 1];
    double dt = i3.t - i2.t;
    V3 w = ((V3{i2.gyr[0], i2.gyr[1], i2.gyr[2]} + V3{i3.gyr[0], i3.gyr[1], i3.gyr[2]}) / 2 - V3{bg[0], bg[1], bg[2]}) * dt;
    Q4 q = qmul({i2.quat[0], i2.quat[1], i2.quat[2], i2.quat[3]}, so3_exp(w));
    V3 a = qrot({i1.quat[0], i1.quat[1], i1.quat[2], i1.quat[3]}, V3{i1.acc[0], i1.acc[1], i1.acc[2]} - V3{ba[0], ba[1], ba[2]}) +
           V3{grav[0], grav[1], grav[2]};
    V3 p = (a * dt) * dt + 2 * V3{i2.pos[0], i2.pos[1], i2.pos[2]} - V3{i1.pos[0], i1.pos[1], i1.pos[2]};
    i3.quat[0] = q.w, i3.quat[1] = q.x, i3.quat[2] = q.y, i3.quat[3] = q.z;
    i3.pos[0] = p.x, i3.pos[1] = p.y, i3.pos[2] = p.z;
  }
  return 0;
}
