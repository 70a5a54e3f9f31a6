// cubic_bspline.h — the facade's post-solve corrector spline, header-only so that the known-answer
// tests of the reference (src/odometry/spline_interpolation_test.cc:79-96, scripts/CubicBSpline3D.ipynb) can be run against
// THIS class through host/odom_c_api.cc, not only against the oracle's restatement.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "../csrc/dmath.h"

namespace wc {

// CubicBSplineInterpolator (src/odometry/spline_interpolation.h:42-113): uniform cubic B-spline through Np samples,
// control points from the normal equations of the knot-evaluation matrix, end indices clamped.
class CubicBSpline {
 public:
  CubicBSpline(const std::vector<double> &ts, const std::vector<V3> &pts) : ts_(ts), np_((int)ts.size()), q_(ts.size()) {
    const int n = np_;
    std::vector<double> N((size_t)n * n, 0.0);
    const double w[4] = {1.0 / 6, 4.0 / 6, 1.0 / 6, 0.0};  // (0,0,0,1) . M / 6
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < 4; ++j) N[(size_t)i * n + std::min(std::max(i - 1 + j, 0), n - 1)] += w[j];
    std::vector<double> A((size_t)n * n, 0.0), B((size_t)n * 3, 0.0);
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < n; ++k) {
        const double nki = N[(size_t)k * n + i];
        if (nki == 0.0) continue;
        for (int j = 0; j < n; ++j) A[(size_t)i * n + j] += nki * N[(size_t)k * n + j];
        B[(size_t)i * 3 + 0] += nki * pts[k].x, B[(size_t)i * 3 + 1] += nki * pts[k].y, B[(size_t)i * 3 + 2] += nki * pts[k].z;
      }
    // Cholesky solve of the (symmetric positive definite) normal equations
    for (int j = 0; j < n; ++j) {
      double d = A[(size_t)j * n + j];
      for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
      d = std::sqrt(d);
      A[(size_t)j * n + j] = d;
      for (int i = j + 1; i < n; ++i) {
        double s = A[(size_t)i * n + j];
        for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
        A[(size_t)i * n + j] = s / d;
      }
    }
    for (int c = 0; c < 3; ++c) {
      std::vector<double> y(n);
      for (int i = 0; i < n; ++i) {
        double s = B[(size_t)i * 3 + c];
        for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * y[k];
        y[i] = s / A[(size_t)i * n + i];
      }
      for (int i = n - 1; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * y[k];
        y[i] = s / A[(size_t)i * n + i];
      }
      for (int i = 0; i < n; ++i) (c == 0 ? q_[i].x : (c == 1 ? q_[i].y : q_[i].z)) = y[i];
    }
  }
  bool Interp(double t, V3 &out) const {  // :51-72
    if (t < ts_.front() || t > ts_.back()) return false;
    const double index_f = (t - ts_.front()) / (ts_.back() - ts_.front()) * (np_ - 1) + 1.0;
    const int index_int = (int)std::floor(index_f);
    const double u = index_f - index_int;
    static const double M[4][4] = {{-1, 3, -3, 1}, {3, -6, 3, 0}, {-3, 0, 3, 0}, {1, 4, 1, 0}};
    const double tv[4] = {u * u * u, u * u, u, 1.0};
    out = mk3(0, 0, 0);
    for (int j = 0; j < 4; ++j) {
      double wj = 0;
      for (int k = 0; k < 4; ++k) wj += tv[k] * M[k][j];
      out = out + wj * q_[std::min(std::max(index_int - 2 + j, 0), np_ - 1)];
    }
    out = out / 6.0;
    return true;
  }

 const std::vector<V3> &control_points() const { return q_; }

 private:
  std::vector<double> ts_;
  int np_;
  std::vector<V3> q_;
};


}  // namespace wc
