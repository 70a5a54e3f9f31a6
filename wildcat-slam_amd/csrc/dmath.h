// dmath.h — fp64 3-vector / 3x3 / quaternion / SO(3) primitives for the gfx950 kernels (and the host-side
// solver glue).  Semantics follow the pieces of Eigen / Sophus the reference path uses:
//   SO(3) exp/log        3rd-party/Sophus-1.22.10/sophus/so3.hpp:694-731 / :264-311 (eps 1e-10, common.hpp:157)
//   Hat, Jl, Jl_inv, Jr  src/common/utils.h:15-67
//   quaternion algebra   Eigen::Quaternion as used in src/odometry/surfel.h:48-91, lidar_odometry.cc:153,167
//   sym. 3x3 eigensolve  Eigen::SelfAdjointEigenSolver<Matrix3d> at surfel_extraction.cc:49,98, cost_functor.h:23,111
// Compile the including TU with -ffp-contract=off: gate decisions are compared bit-for-bit with the CPU path.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define WC_HD __host__ __device__ __forceinline__
#else  // plain host translation units (the C++ facade) use the same arithmetic
#include <cmath>
#define WC_HD inline
using std::acos;
using std::atan2;
using std::cos;
using std::fabs;
using std::floor;
using std::sin;
using std::sqrt;
#endif

namespace wc {

struct V3 {
  double x, y, z;
};
WC_HD V3 mk3(double x, double y, double z) { return V3{x, y, z}; }
WC_HD V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
WC_HD V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
WC_HD V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
WC_HD V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
WC_HD V3 operator*(V3 a, double s) { return {s * a.x, s * a.y, s * a.z}; }
WC_HD V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
WC_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
WC_HD V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
WC_HD double norm(V3 a) { return sqrt(dot(a, a)); }
WC_HD double get(const V3 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

struct M3 {
  double m[3][3];
};
WC_HD M3 m3_zero() {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = 0.0;
  return r;
}
WC_HD M3 m3_identity() {
  M3 r = m3_zero();
  r.m[0][0] = r.m[1][1] = r.m[2][2] = 1.0;
  return r;
}
WC_HD M3 operator*(const M3 &a, const M3 &b) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
WC_HD M3 operator+(const M3 &a, const M3 &b) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] + b.m[i][j];
  return r;
}
WC_HD M3 operator*(double s, const M3 &a) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = s * a.m[i][j];
  return r;
}
WC_HD V3 operator*(const M3 &a, V3 v) {
  return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
          a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
WC_HD M3 transpose(const M3 &a) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
  return r;
}
WC_HD M3 outer(V3 a, V3 b) {
  M3 r;
  r.m[0][0] = a.x * b.x, r.m[0][1] = a.x * b.y, r.m[0][2] = a.x * b.z;
  r.m[1][0] = a.y * b.x, r.m[1][1] = a.y * b.y, r.m[1][2] = a.y * b.z;
  r.m[2][0] = a.z * b.x, r.m[2][1] = a.z * b.y, r.m[2][2] = a.z * b.z;
  return r;
}
WC_HD V3 vecmat(V3 v, const M3 &a) {  // v^T A
  return {v.x * a.m[0][0] + v.y * a.m[1][0] + v.z * a.m[2][0], v.x * a.m[0][1] + v.y * a.m[1][1] + v.z * a.m[2][1],
          v.x * a.m[0][2] + v.y * a.m[1][2] + v.z * a.m[2][2]};
}
WC_HD M3 hat(V3 v) {  // utils.h:15-22
  M3 r = m3_zero();
  r.m[0][1] = -v.z, r.m[0][2] = v.y;
  r.m[1][0] = v.z, r.m[1][2] = -v.x;
  r.m[2][0] = -v.y, r.m[2][1] = v.x;
  return r;
}

struct Q4 {
  double w, x, y, z;
};
WC_HD Q4 qmul(Q4 a, Q4 b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
WC_HD Q4 qconj(Q4 a) { return {a.w, -a.x, -a.y, -a.z}; }
WC_HD V3 qrot(Q4 q, V3 v) {  // v + 2w(u x v) + 2 u x (u x v)
  V3 u{q.x, q.y, q.z};
  V3 uv = cross(u, v);
  uv = uv + uv;
  return v + q.w * uv + cross(u, uv);
}
WC_HD M3 qmat(Q4 q) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  M3 r;
  r.m[0][0] = 1 - (tyy + tzz), r.m[0][1] = txy - twz, r.m[0][2] = txz + twy;
  r.m[1][0] = txy + twz, r.m[1][1] = 1 - (txx + tzz), r.m[1][2] = tyz - twx;
  r.m[2][0] = txz - twy, r.m[2][1] = tyz + twx, r.m[2][2] = 1 - (txx + tyy);
  return r;
}
WC_HD Q4 qslerp(Q4 a, double t, Q4 b) {  // Eigen slerp: unnormalised, linear when |dot| >= 1 - eps
  const double one = 1.0 - 2.220446049250313e-16;
  double d = a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z;
  double ad = fabs(d), s0, s1;
  if (ad >= one) {
    s0 = 1.0 - t;
    s1 = t;
  } else {
    double th = acos(ad), st = sin(th);
    s0 = sin((1.0 - t) * th) / st;
    s1 = sin(t * th) / st;
  }
  if (d < 0) s1 = -s1;
  return {s0 * a.w + s1 * b.w, s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z};
}
WC_HD Q4 so3_exp(V3 w) {  // so3.hpp:694-731
  double th2 = dot(w, w), imag, real;
  if (th2 < 1e-10 * 1e-10) {
    double th4 = th2 * th2;
    imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
    real = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
  } else {
    double th = sqrt(th2), half = 0.5 * th;
    imag = sin(half) / th;
    real = cos(half);
  }
  return {real, imag * w.x, imag * w.y, imag * w.z};
}
WC_HD V3 so3_log(Q4 q) {  // so3.hpp:264-311 after the SO3(q) constructor's normalisation
  double nn = sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  q = {q.w / nn, q.x / nn, q.y / nn, q.z / nn};
  double sq = q.x * q.x + q.y * q.y + q.z * q.z, w = q.w, k;
  if (sq < 1e-10 * 1e-10) {
    k = 2.0 / w - (2.0 / 3.0) * sq / (w * (w * w));
  } else {
    double n = sqrt(sq);
    double at = (w < 0) ? atan2(-n, -w) : atan2(n, w);
    k = 2.0 * at / n;
  }
  return {k * q.x, k * q.y, k * q.z};
}
WC_HD M3 so3_Jl(V3 v) {  // utils.h:46-58
  double th = norm(v);
  if (th > 1e-10) {
    V3 a = v / th;
    double s = sin(th) / th;
    return s * m3_identity() + (1 - s) * outer(a, a) + ((1 - cos(th)) / th) * hat(a);
  }
  return m3_identity();
}
WC_HD M3 so3_Jl_inv(V3 v) {  // utils.h:32-43
  double th = norm(v);
  if (th > 1e-10) {
    M3 H = hat(v);
    double k = (1 - th * cos(th / 2) / 2 / sin(th / 2));
    return m3_identity() + (-0.5) * H + (k / dot(v, v)) * (H * H);
  }
  return m3_identity();
}
WC_HD M3 so3_Jr(V3 v) { return so3_Jl(-v); }
WC_HD M3 so3_Jr_inv(V3 v) { return so3_Jl_inv(-v); }

// Symmetric 3x3 eigen-decomposition (cyclic Jacobi, fp64): ascending eigenvalues, eigenvectors in the columns
// of V.  Fully unrolled over the three (p,q) planes so that everything stays in registers on the GPU.
WC_HD void eig3_sym(const M3 &A, double ev[3], M3 &V) {
  double a00 = A.m[0][0], a11 = A.m[1][1], a22 = A.m[2][2];
  double a01 = A.m[1][0], a02 = A.m[2][0], a12 = A.m[2][1];  // lower triangle, like SelfAdjointEigenSolver
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
#define WC_JACOBI(app, aqq, apq, apr, aqr, P, Q)                                          \
  if (apq != 0.0) {                                                                       \
    double theta = (aqq - app) / (2.0 * apq);                                             \
    double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));     \
    double c = 1.0 / sqrt(t * t + 1.0), s = t * c;                                        \
    double npp = c * (c * app - s * apq) - s * (c * apq - s * aqq);                       \
    double nqq = s * (s * app + c * apq) + c * (s * apq + c * aqq);                       \
    double npr = c * apr - s * aqr, nqr = s * apr + c * aqr;                              \
    app = npp, aqq = nqq, apq = 0.0, apr = npr, aqr = nqr;                                \
    for (int k = 0; k < 3; ++k) {                                                         \
      double vp = v[k][P], vq = v[k][Q];                                                  \
      v[k][P] = c * vp - s * vq;                                                          \
      v[k][Q] = s * vp + c * vq;                                                          \
    }                                                                                     \
  }
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = a01 * a01 + a02 * a02 + a12 * a12;
    double dg = a00 * a00 + a11 * a11 + a22 * a22;
    if (off <= 1e-34 * dg || off == 0.0) break;
    WC_JACOBI(a00, a11, a01, a02, a12, 0, 1)
    WC_JACOBI(a00, a22, a02, a01, a12, 0, 2)
    WC_JACOBI(a11, a22, a12, a01, a02, 1, 2)
  }
#undef WC_JACOBI
  double d0 = a00, d1 = a11, d2 = a22;
  int i0 = 0, i1 = 1, i2 = 2;
  if (d0 > d1) {
    double td = d0;
    d0 = d1, d1 = td;
    int ti = i0;
    i0 = i1, i1 = ti;
  }
  if (d1 > d2) {
    double td = d1;
    d1 = d2, d2 = td;
    int ti = i1;
    i1 = i2, i2 = ti;
  }
  if (d0 > d1) {
    double td = d0;
    d0 = d1, d1 = td;
    int ti = i0;
    i0 = i1, i1 = ti;
  }
  ev[0] = d0, ev[1] = d1, ev[2] = d2;
  for (int r = 0; r < 3; ++r) {
    V.m[r][0] = v[r][i0];
    V.m[r][1] = v[r][i1];
    V.m[r][2] = v[r][i2];
  }
}

}  // namespace wc
