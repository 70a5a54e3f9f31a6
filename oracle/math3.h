// oracle/math3.h — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// Dependency-free fp64 3-vector / 3x3 / quaternion / SO(3) helpers used by the CPU restatement of the
// reference hot path.  Eigen and Sophus are not available in this image, so the small pieces of them
// the reference leans on are restated here from their published semantics:
//   * SO(3) exp / log  : 3rd-party/Sophus-1.22.10/sophus/so3.hpp:264-311 (log), :694-731 (exp),
//                        epsilon 1e-10 from sophus/common.hpp:157
//   * Hat/Jl/Jl_inv/Jr/Jr_inv : src/common/utils.h:15-67
//   * quaternion product / rotate / slerp / toRotationMatrix : Eigen::Quaternion (upstream, not in
//     reference) as used at src/odometry/lidar_odometry.cc:153,167 and src/odometry/surfel.h:48-91
//   * symmetric 3x3 eigen-decomposition, ascending eigenvalues : stands in for
//     Eigen::SelfAdjointEigenSolver<Matrix3d> (surfel_extraction.cc:49,98; cost_functor.h:23,111).
//     Cyclic Jacobi in fp64 — any accurate solver agrees to ~1e-15 on the eigenvalues.
#pragma once
#include <cmath>
#include <cstring>
#include <utility>

namespace wco {

struct V3 {
  double x, y, z;
  double &operator[](int i) { return (&x)[i]; }
  double operator[](int i) const { return (&x)[i]; }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, double s) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }

struct M3 {
  double m[3][3];
  static M3 zero() {
    M3 r;
    std::memset(r.m, 0, sizeof(r.m));
    return r;
  }
  static M3 identity() {
    M3 r = zero();
    r.m[0][0] = r.m[1][1] = r.m[2][2] = 1.0;
    return r;
  }
};
inline M3 operator*(const M3 &a, const M3 &b) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
inline M3 operator+(const M3 &a, const M3 &b) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] + b.m[i][j];
  return r;
}
inline M3 operator-(const M3 &a, const M3 &b) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] - b.m[i][j];
  return r;
}
inline M3 operator*(double s, const M3 &a) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = s * a.m[i][j];
  return r;
}
inline V3 operator*(const M3 &a, V3 v) {
  return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
          a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
inline M3 transpose(const M3 &a) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
  return r;
}
inline M3 outer(V3 a, V3 b) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a[i] * b[j];
  return r;
}
// row vector times matrix: (v^T A)
inline V3 vecmat(V3 v, const M3 &a) {
  return {v.x * a.m[0][0] + v.y * a.m[1][0] + v.z * a.m[2][0], v.x * a.m[0][1] + v.y * a.m[1][1] + v.z * a.m[2][1],
          v.x * a.m[0][2] + v.y * a.m[1][2] + v.z * a.m[2][2]};
}

// src/common/utils.h:15-22
inline M3 hat(V3 v) {
  M3 r = M3::zero();
  r.m[0][1] = -v.z;
  r.m[0][2] = v.y;
  r.m[1][0] = v.z;
  r.m[1][2] = -v.x;
  r.m[2][0] = -v.y;
  r.m[2][1] = v.x;
  return r;
}

struct Q4 {  // (w, x, y, z)
  double w, x, y, z;
};
inline Q4 qmul(Q4 a, Q4 b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
inline Q4 qconj(Q4 a) { return {a.w, -a.x, -a.y, -a.z}; }
// Eigen QuaternionBase::_transformVector: v + 2 w (q x v) + 2 q x (q x v)
inline V3 qrot(Q4 q, V3 v) {
  V3 u{q.x, q.y, q.z};
  V3 uv = cross(u, v);
  uv = uv + uv;
  return v + q.w * uv + cross(u, uv);
}
inline M3 qmat(Q4 q) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  M3 r;
  r.m[0][0] = 1 - (tyy + tzz);
  r.m[0][1] = txy - twz;
  r.m[0][2] = txz + twy;
  r.m[1][0] = txy + twz;
  r.m[1][1] = 1 - (txx + tzz);
  r.m[1][2] = tyz - twx;
  r.m[2][0] = txz - twy;
  r.m[2][1] = tyz + twx;
  r.m[2][2] = 1 - (txx + tyy);
  return r;
}
// Eigen QuaternionBase::slerp(t, other): no normalisation, linear blend when |dot| >= 1 - eps.
inline Q4 qslerp(Q4 a, double t, Q4 b) {
  const double one = 1.0 - 2.220446049250313e-16;
  double d = a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z;
  double ad = std::fabs(d);
  double s0, s1;
  if (ad >= one) {
    s0 = 1.0 - t;
    s1 = t;
  } else {
    double th = std::acos(ad);
    double st = std::sin(th);
    s0 = std::sin((1.0 - t) * th) / st;
    s1 = std::sin(t * th) / st;
  }
  if (d < 0) s1 = -s1;
  return {s0 * a.w + s1 * b.w, s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z};
}

// Sophus SO3d::exp as a unit quaternion (so3.hpp:694-731).
inline Q4 so3_exp(V3 w) {
  double th2 = dot(w, w);
  double imag, real;
  if (th2 < 1e-10 * 1e-10) {
    double th4 = th2 * th2;
    imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
    real = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
  } else {
    double th = std::sqrt(th2);
    double half = 0.5 * th;
    imag = std::sin(half) / th;
    real = std::cos(half);
  }
  return {real, imag * w.x, imag * w.y, imag * w.z};
}
// Sophus SO3d(q).log() (so3.hpp:264-311).  The SO3 constructor normalises the quaternion first.
inline V3 so3_log(Q4 q) {
  double nn = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  q = {q.w / nn, q.x / nn, q.y / nn, q.z / nn};
  double sq = q.x * q.x + q.y * q.y + q.z * q.z;
  double w = q.w, k;
  if (sq < 1e-10 * 1e-10) {
    double sw = w * w;
    k = 2.0 / w - (2.0 / 3.0) * sq / (w * sw);
  } else {
    double n = std::sqrt(sq);
    double at = (w < 0) ? std::atan2(-n, -w) : std::atan2(n, w);
    k = 2.0 * at / n;
  }
  return {k * q.x, k * q.y, k * q.z};
}

// src/common/utils.h:46-58
inline M3 so3_Jl(V3 v) {
  double th = norm(v);
  if (th > 1e-10) {
    V3 a = v / th;
    double s = std::sin(th) / th;
    M3 r = s * M3::identity() + (1 - s) * outer(a, a) + ((1 - std::cos(th)) / th) * hat(a);
    return r;
  }
  return M3::identity();
}
// src/common/utils.h:32-43
inline M3 so3_Jl_inv(V3 v) {
  double th = norm(v);
  if (th > 1e-10) {
    M3 H = hat(v);
    double k = (1 - th * std::cos(th / 2) / 2 / std::sin(th / 2));
    M3 HH = H * H;
    return M3::identity() - 0.5 * H + (k / dot(v, v)) * HH;
  }
  return M3::identity();
}
inline M3 so3_Jr(V3 v) { return so3_Jl(-v); }          // utils.h:60-63
inline M3 so3_Jr_inv(V3 v) { return so3_Jl_inv(-v); }  // utils.h:65-67

// Symmetric 3x3 eigen-decomposition by cyclic Jacobi rotations. Eigenvalues ascending in `ev`,
// eigenvectors in the columns of `V` (unit length; sign arbitrary, as with Eigen — SURVEY Q9).
inline void eig3_sym(const M3 &A, double ev[3], M3 &V) {
  // reads the lower triangle, like SelfAdjointEigenSolver
  double a00 = A.m[0][0], a11 = A.m[1][1], a22 = A.m[2][2];
  double a01 = A.m[1][0], a02 = A.m[2][0], a12 = A.m[2][1];
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  // one two-sided Jacobi rotation in the (P,Q) plane; R is the third index
  auto rotate = [&](double &app, double &aqq, double &apq, double &apr, double &aqr, int P, int Q) {
    if (apq == 0.0) return;
    double theta = (aqq - app) / (2.0 * apq);
    double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
    double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
    double npp = c * (c * app - s * apq) - s * (c * apq - s * aqq);
    double nqq = s * (s * app + c * apq) + c * (s * apq + c * aqq);
    double npr = c * apr - s * aqr, nqr = s * apr + c * aqr;
    app = npp, aqq = nqq, apq = 0.0, apr = npr, aqr = nqr;
    for (int k = 0; k < 3; ++k) {
      double vp = v[k][P], vq = v[k][Q];
      v[k][P] = c * vp - s * vq;
      v[k][Q] = s * vp + c * vq;
    }
  };
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = a01 * a01 + a02 * a02 + a12 * a12;
    double dg = a00 * a00 + a11 * a11 + a22 * a22;
    if (off <= 1e-34 * dg || off == 0.0) break;
    rotate(a00, a11, a01, a02, a12, 0, 1);
    rotate(a00, a22, a02, a01, a12, 0, 2);
    rotate(a11, a22, a12, a01, a02, 1, 2);
  }
  double d[3] = {a00, a11, a22};
  int idx[3] = {0, 1, 2};
  auto cswap = [&](int i, int j) {
    if (d[i] > d[j]) {
      std::swap(d[i], d[j]);
      std::swap(idx[i], idx[j]);
    }
  };
  cswap(0, 1);
  cswap(1, 2);
  cswap(0, 1);
  for (int c = 0; c < 3; ++c) {
    ev[c] = d[c];
    for (int r = 0; r < 3; ++r) V.m[r][c] = v[r][idx[c]];
  }
}

}  // namespace wco
