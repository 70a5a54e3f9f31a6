// oracle/window.cc — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// Restatement of the reference's window optimisation problem:
//   SurfelMatchBinaryFactor<0/1/2>  src/odometry/cost_functor.h:100-241  (incl. Jacobian-overwrite quirk Q1)
//   SurfelMatchUnaryFactor          src/odometry/cost_functor.h:16-69
//   ImuFactor<0/1>                  src/odometry/cost_functor.h:264-472  (incl. Q3)
//   BuildSldWinLidarResiduals       src/odometry/lidar_odometry.cc:254-297
//   BuildFixWinLidarResiduals       src/odometry/lidar_odometry.cc:299-317
//   BuildImuResiduals               src/odometry/lidar_odometry.cc:319-363
//   solve options / gauge           src/odometry/lidar_odometry.cc:551-561
// Ceres Solver (un-vendored; Ubuntu 20.04 => 1.14.0, must be < 2.2 because of SetParameterization) is replaced
// by a restatement of its documented default trust-region Levenberg-Marquardt loop (upstream, not in the
// reference): CauchyLoss + Corrector, Jacobi column scaling 1/(1+||col||), LM diagonal clamp [1e-6,1e32]/radius,
// radius0 = 1e4, step quality rho > 1e-3, radius /= max(1/3, 1-(2rho-1)^3) | radius /= nu, nu *= 2,
// function_tolerance 1e-6, gradient_tolerance 1e-10, parameter_tolerance 1e-8, max_num_iterations.
// PARITY UNPINNED against real Ceres (not installed here).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "math3.h"
#include "wc_oracle.h"

namespace {
using namespace wco;

inline V3 v3(const double *p) { return {p[0], p[1], p[2]}; }
inline Q4 q4(const double *p) { return {p[0], p[1], p[2], p[3]}; }

struct SurfelRef {  // what the factors read from a Surfel (surfel.h)
  double t;
  V3 c_body;
  M3 cov_body;
  Q4 rot;
  V3 pos;
};
SurfelRef make_ref(const wc_surfel &s, const wc_pose &p) {
  SurfelRef r;
  r.t = s.t;
  r.c_body = v3(s.center);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.cov_body.m[i][j] = s.cov[3 * i + j];
  r.rot = q4(p.quat);
  r.pos = v3(p.pos);
  return r;
}
M3 cov_world(const SurfelRef &s) {  // GetCovarianceInWorld, surfel.h:89-91
  M3 R = qmat(s.rot);
  return (R * s.cov_body) * transpose(R);
}

struct LidarFactor {
  int kind;  // 0,1,2 = binary mode, 3 = unary
  SurfelRef s1, s2;
  int sp1l, sp1r, sp2l, sp2r;
  V3 n;      // norm_
  double w;  // weight_
};
struct ImuFactorRec {
  int mode;  // 0: three blocks, 1: two blocks
  wc_imu_state i1, i2, i3;
  int sp1, sp2, sp3;
};
}  // namespace

struct wco_window {
  wc_params P;
  std::vector<double> times;
  V3 grav;
  int fix_first_pos;
  std::vector<LidarFactor> lidar;
  std::vector<ImuFactorRec> imu;
  uint64_t counts[6] = {0, 0, 0, 0, 0, 0};
};

namespace {

// ctor of both surfel factors (cost_functor.h:21-25,109-113)
void weight_and_normal(const wc_params &P, const SurfelRef &a, const SurfelRef &b, V3 &n, double &w) {
  M3 cov = cov_world(a) + cov_world(b);
  double ev[3];
  M3 V;
  eig3_sym(cov, ev, V);
  w = 1 / std::sqrt(std::pow(P.surfel_sigma0, 2) + ev[0]);
  n = {V.m[0][0], V.m[1][0], V.m[2][0]};
}

// std::upper_bound over sample timestamps (lidar_odometry.cc:258,263,303)
int upper_bound_idx(const std::vector<double> &ts, double t) {
  return (int)(std::upper_bound(ts.begin(), ts.end(), t) - ts.begin());
}

struct RowBlocks {  // one residual block's Jacobian: nres x (nblk*12), row-major per block
  int nres, nblk;
  int blk[4];
  double r[12];
  double J[4][12 * 12];  // J[b][row*12 + col]
};

// 1x12 surfel Jacobian pieces (cost_functor.h:147-150,162-165 / :42-45)
void surfel_jac(const LidarFactor &f, const SurfelRef &s, V3 r_s, double sign, double out[12]) {
  M3 E = qmat(so3_exp(r_s));
  V3 a = qrot(s.rot, s.c_body);
  M3 T = (E * hat(a)) * so3_Jr(r_s);
  V3 wn = f.w * f.n;
  V3 row = vecmat(wn, T);
  for (int i = 0; i < 3; ++i) {
    out[i] = sign * row[i];
    out[3 + i] = -sign * wn[i];
    out[6 + i] = 0;
    out[9 + i] = 0;
  }
}

void eval_lidar(const wco_window &W, const LidarFactor &f, const double *x, RowBlocks &rb, bool want_jac) {
  const std::vector<double> &ts = W.times;
  const bool quirks = W.P.reference_quirks != 0;
  rb.nres = 1;
  const double *b2l = x + 12 * f.sp2l, *b2r = x + 12 * f.sp2r;
  double f2 = (f.s2.t - ts[f.sp2l]) / (ts[f.sp2r] - ts[f.sp2l]);
  V3 r_s2 = (1 - f2) * v3(b2l) + f2 * v3(b2r);
  V3 t_s2 = (1 - f2) * v3(b2l + 3) + f2 * v3(b2r + 3);
  V3 term2 = qrot(qmul(so3_exp(r_s2), f.s2.rot), f.s2.c_body);
  if (f.kind == 3) {
    // unary: cost_functor.h:28-59
    V3 c1w = qrot(f.s1.rot, f.s1.c_body) + f.s1.pos;
    rb.r[0] = f.w * dot(f.n, ((c1w - term2) - t_s2) - f.s2.pos);
    rb.nblk = 2;
    rb.blk[0] = f.sp2l;
    rb.blk[1] = f.sp2r;
    if (want_jac) {
      double j2[12];
      surfel_jac(f, f.s2, r_s2, +1.0, j2);
      for (int c = 0; c < 12; ++c) {
        rb.J[0][c] = j2[c] * (1 - f2);
        rb.J[1][c] = j2[c] * f2;
      }
    }
    return;
  }
  // binary: cost_functor.h:116-179
  const double *b1l = x + 12 * f.sp1l, *b1r = x + 12 * f.sp1r;
  double f1 = (f.s1.t - ts[f.sp1l]) / (ts[f.sp1r] - ts[f.sp1l]);
  V3 r_s1 = (1 - f1) * v3(b1l) + f1 * v3(b1r);
  V3 t_s1 = (1 - f1) * v3(b1l + 3) + f1 * v3(b1r + 3);
  V3 term1 = qrot(qmul(so3_exp(r_s1), f.s1.rot), f.s1.c_body);
  rb.r[0] = f.w * dot(f.n, ((((term1 + t_s1) + f.s1.pos) - term2) - t_s2) - f.s2.pos);
  // parameter-block slots as handed to AddResidualBlock (lidar_odometry.cc:272-293)
  int slot1l = 0, slot1r = 1, slot2l, slot2r;
  if (f.kind == 0) {
    rb.nblk = 4;
    rb.blk[0] = f.sp1l, rb.blk[1] = f.sp1r, rb.blk[2] = f.sp2l, rb.blk[3] = f.sp2r;
    slot2l = 2, slot2r = 3;
  } else if (f.kind == 1) {
    rb.nblk = 3;
    rb.blk[0] = f.sp1l, rb.blk[1] = f.sp1r, rb.blk[2] = f.sp2r;
    slot2l = 1, slot2r = 2;  // DispatchPtr, cost_functor.h:222-224
  } else {
    rb.nblk = 2;
    rb.blk[0] = f.sp1l, rb.blk[1] = f.sp1r;
    slot2l = 0, slot2r = 1;  // cost_functor.h:225-228
  }
  if (!want_jac) return;
  double j1[12], j2[12];
  surfel_jac(f, f.s1, r_s1, -1.0, j1);
  surfel_jac(f, f.s2, r_s2, +1.0, j2);
  for (int b = 0; b < rb.nblk; ++b)
    for (int c = 0; c < 12; ++c) rb.J[b][c] = 0;
  if (quirks) {
    // four plain assignments in this order; a later write to an aliased slot wins (Q1, cost_functor.h:152-175)
    for (int c = 0; c < 12; ++c) rb.J[slot1l][c] = j1[c] * (1 - f1);
    for (int c = 0; c < 12; ++c) rb.J[slot1r][c] = j1[c] * f1;
    for (int c = 0; c < 12; ++c) rb.J[slot2l][c] = j2[c] * (1 - f2);
    for (int c = 0; c < 12; ++c) rb.J[slot2r][c] = j2[c] * f2;
  } else {
    for (int c = 0; c < 12; ++c) rb.J[slot1l][c] += j1[c] * (1 - f1);
    for (int c = 0; c < 12; ++c) rb.J[slot1r][c] += j1[c] * f1;
    for (int c = 0; c < 12; ++c) rb.J[slot2l][c] += j2[c] * (1 - f2);
    for (int c = 0; c < 12; ++c) rb.J[slot2r][c] += j2[c] * f2;
  }
}

struct Corr {  // ComputeStateCorr, cost_functor.h:358-400
  V3 r, t, bg, ba;
  int bl, br;  // local block slots (0..2)
  double f;
};
Corr state_corr(const double *const blocks[3], const double tsp[3], int mode, double t) {
  Corr c;
  bool first = (mode == 1) ? true : (t >= tsp[0] && t < tsp[1]);
  c.bl = first ? 0 : 1;
  c.br = first ? 1 : 2;
  const double *l = blocks[c.bl], *r = blocks[c.br];
  c.f = (t - tsp[c.bl]) / (tsp[c.br] - tsp[c.bl]);
  c.r = (1 - c.f) * v3(l) + c.f * v3(r);
  c.t = (1 - c.f) * v3(l + 3) + c.f * v3(r + 3);
  c.bg = (1 - c.f) * v3(l + 6) + c.f * v3(r + 6);
  c.ba = (1 - c.f) * v3(l + 9) + c.f * v3(r + 9);
  return c;
}
// F(L, R, r) = Jr_inv(Log(L Exp(r) R)) R^T Jr(r), cost_functor.h:446-448
M3 Ffun(Q4 L, Q4 R, V3 r) {
  V3 lg = so3_log(qmul(qmul(L, so3_exp(r)), R));
  return (so3_Jr_inv(lg) * qmat(qconj(R))) * so3_Jr(r);
}
void set_block(double *tau, int r0, int c0, const M3 &m, double s) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) tau[(r0 + i) * 12 + c0 + j] = s * m.m[i][j];
}

void eval_imu(const wco_window &W, const ImuFactorRec &f, const double *x, RowBlocks &rb, bool want_jac) {
  const wc_params &P = W.P;
  const double dt = P.imu_dt;
  const int nb = f.mode == 0 ? 3 : 2;
  const double *blocks[3] = {x + 12 * f.sp1, x + 12 * f.sp2, f.mode == 0 ? x + 12 * f.sp3 : nullptr};
  const double tsp[3] = {W.times[f.sp1], W.times[f.sp2], f.mode == 0 ? W.times[f.sp3] : DBL_MAX};
  Corr c1 = state_corr(blocks, tsp, f.mode, f.i1.t);
  Corr c2 = state_corr(blocks, tsp, f.mode, f.i2.t);
  Corr c3 = state_corr(blocks, tsp, f.mode, f.i3.t);
  Q4 R1 = q4(f.i1.quat), R2 = q4(f.i2.quat);
  V3 p1 = v3(f.i1.pos), p2 = v3(f.i2.pos), p3 = v3(f.i3.pos);
  Q4 E1R1 = qmul(so3_exp(c1.r), R1);
  Q4 E2R2 = qmul(so3_exp(c2.r), R2);
  // cost_functor.h:291-298
  V3 gyr_est = so3_log(qmul(qconj(E1R1), E2R2)) / dt;
  V3 acc_est = (((c3.t + p3) + (c1.t + p1)) - 2 * (c2.t + p2)) / (dt * dt);
  V3 r0 = P.w_gyr * (((v3(f.i1.gyr) + v3(f.i2.gyr)) / 2 - gyr_est) - c1.bg);
  V3 r1 = P.w_acc * ((qrot(E1R1, v3(f.i1.acc) - c1.ba) - acc_est) + W.grav);
  V3 r2 = P.w_bg * (c1.bg - c2.bg);
  V3 r3 = P.w_ba * (c1.ba - c2.ba);
  rb.nres = 12;
  rb.nblk = nb;
  rb.blk[0] = f.sp1, rb.blk[1] = f.sp2;
  if (nb == 3) rb.blk[2] = f.sp3;
  for (int i = 0; i < 3; ++i) {
    rb.r[i] = r0[i];
    rb.r[3 + i] = r1[i];
    rb.r[6 + i] = r2[i];
    rb.r[9 + i] = r3[i];
  }
  if (!want_jac) return;
  // cost_functor.h:301-321
  double tau[3][144];
  std::memset(tau, 0, sizeof(tau));
  const M3 I = M3::identity();
  set_block(tau[0], 0, 0, Ffun(qconj(R1), E2R2, c1.r), P.w_gyr * (1 / dt));
  set_block(tau[0], 0, 6, I, -P.w_gyr);
  set_block(tau[0], 3, 0, (qmat(so3_exp(c1.r)) * hat(qrot(R1, v3(f.i1.acc) - c1.ba))) * so3_Jr(c1.r), -P.w_acc);
  set_block(tau[0], 3, 3, I, -P.w_acc * (1 / dt / dt));
  set_block(tau[0], 3, 9, qmat(E1R1), -P.w_acc);
  set_block(tau[0], 6, 6, I, P.w_bg);
  set_block(tau[0], 9, 9, I, P.w_ba);
  set_block(tau[1], 0, 0, Ffun(qconj(E1R1), R2, c2.r), -P.w_gyr * (1 / dt));
  if (P.reference_quirks) set_block(tau[1], 0, 6, I, -P.w_gyr);  // Q3 (cost_functor.h:314)
  set_block(tau[1], 3, 3, I, P.w_acc * (2 / dt / dt));
  set_block(tau[1], 6, 6, I, -P.w_bg);
  set_block(tau[1], 9, 9, I, -P.w_ba);
  set_block(tau[2], 3, 3, I, -P.w_acc * (1 / dt / dt));
  for (int b = 0; b < nb; ++b) std::memset(rb.J[b], 0, sizeof(double) * 144);
  // DispatchJacobians (cost_functor.h:402-444): accumulate
  const Corr *cs[3] = {&c1, &c2, &c3};
  for (int k = 0; k < 3; ++k) {
    const Corr &c = *cs[k];
    for (int e = 0; e < 144; ++e) {
      rb.J[c.bl][e] += tau[k][e] * (1 - c.f);
      rb.J[c.br][e] += tau[k][e] * c.f;
    }
  }
}

// ceres::CauchyLoss(a) (upstream): rho(s) = b log(1 + s/b), b = a^2
inline void cauchy(double a, double s, double rho[3]) {
  const double b = a * a, c = 1 / b;
  const double sum = 1 + s * c, inv = 1 / sum;
  rho[0] = b * std::log(sum);
  rho[1] = std::max(DBL_MIN, inv);
  rho[2] = -c * (inv * inv);
}

struct Evaluated {
  double cost;
};

// Walk every residual block in the reference's insertion order (sld binary, fix unary, imu; cc:543-545) and feed
// loss-corrected residual / Jacobian rows to `sink`.
template <class Sink>
double walk(const wco_window &W, const double *x, bool want_jac, Sink &&sink) {
  double cost = 0;
  RowBlocks rb;
  for (const LidarFactor &f : W.lidar) {
    eval_lidar(W, f, x, rb, want_jac);
    double s = rb.r[0] * rb.r[0], rho[3];
    cauchy(W.P.cauchy_a, s, rho);
    cost += 0.5 * rho[0];
    // ceres Corrector: rho'' <= 0 for Cauchy => residual and Jacobian scale by sqrt(rho')
    double sc = std::sqrt(rho[1]);
    rb.r[0] *= sc;
    if (want_jac)
      for (int b = 0; b < rb.nblk; ++b)
        for (int c = 0; c < 12; ++c) rb.J[b][c] *= sc;
    sink(rb);
  }
  for (const ImuFactorRec &f : W.imu) {
    eval_imu(W, f, x, rb, want_jac);
    double s = 0;
    for (int i = 0; i < 12; ++i) s += rb.r[i] * rb.r[i];
    cost += 0.5 * s;  // TrivialLoss
    sink(rb);
  }
  return cost;
}

// dense Cholesky solve of A y = b (A symmetric positive definite, row-major n x n, lower part used; A is destroyed)
bool chol_solve(std::vector<double> &A, std::vector<double> &b, int n) {
  for (int j = 0; j < n; ++j) {
    double *Aj = &A[(size_t)j * n];
    double d = Aj[j];
    for (int k = 0; k < j; ++k) d -= Aj[k] * Aj[k];
    if (!(d > 0)) return false;
    d = std::sqrt(d);
    Aj[j] = d;
    for (int i = j + 1; i < n; ++i) {
      double *Ai = &A[(size_t)i * n];
      double s = Ai[j];
      for (int k = 0; k < j; ++k) s -= Ai[k] * Aj[k];
      Ai[j] = s / d;
    }
  }
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    const double *Ai = &A[(size_t)i * n];
    for (int k = 0; k < i; ++k) s -= Ai[k] * b[k];
    b[i] = s / Ai[i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * b[k];
    b[i] = s / A[(size_t)i * n + i];
  }
  return true;
}

void linearize_dense(const wco_window &W, const double *x, std::vector<double> &H, std::vector<double> &g,
                     double &cost) {
  const int n = 12 * (int)W.times.size();
  H.assign((size_t)n * n, 0.0);
  g.assign(n, 0.0);
  cost = walk(W, x, true, [&](const RowBlocks &rb) {
    for (int r = 0; r < rb.nres; ++r) {
      for (int a = 0; a < rb.nblk; ++a) {
        const double *Ja = &rb.J[a][r * 12];
        for (int i = 0; i < 12; ++i) {
          if (Ja[i] == 0.0) continue;
          const int gi = 12 * rb.blk[a] + i;
          g[gi] += Ja[i] * rb.r[r];
          for (int b = 0; b < rb.nblk; ++b) {
            const double *Jb = &rb.J[b][r * 12];
            double *Hrow = &H[(size_t)gi * n + 12 * rb.blk[b]];
            for (int j = 0; j < 12; ++j) Hrow[j] += Ja[i] * Jb[j];
          }
        }
      }
    }
  });
  if (W.fix_first_pos) {  // SubsetParameterization(12, {3,4,5}) on block 0: those columns do not exist
    for (int c = 3; c < 6; ++c) {
      for (int i = 0; i < n; ++i) H[(size_t)i * n + c] = H[(size_t)c * n + i] = 0.0;
      g[c] = 0.0;
    }
  }
}
}  // namespace

extern "C" wco_window *wco_window_create(const wc_params *P, const double *sample_times, uint64_t ns,
                                         const double grav[3], int fix_first_pos) {
  wco_window *w = new wco_window;
  w->P = *P;
  w->times.assign(sample_times, sample_times + ns);
  w->grav = {grav[0], grav[1], grav[2]};
  w->fix_first_pos = fix_first_pos;
  return w;
}
extern "C" void wco_window_destroy(wco_window *w) { delete w; }

extern "C" int wco_window_add_binary(wco_window *w, const wc_surfel *surf, const wc_pose *pose, const wc_pair *pairs,
                                     uint64_t n) {
  const int ns = (int)w->times.size();
  for (uint64_t k = 0; k < n; ++k) {
    LidarFactor f;
    f.s1 = make_ref(surf[pairs[k].first], pose[pairs[k].first]);
    f.s2 = make_ref(surf[pairs[k].second], pose[pairs[k].second]);
    if (!(f.s1.t < f.s2.t)) return 3;  // CHECK_LT, lidar_odometry.cc:256
    int i1 = upper_bound_idx(w->times, f.s1.t), i2 = upper_bound_idx(w->times, f.s2.t);
    if (i1 == 0 || i1 == ns || i2 == 0 || i2 == ns) return 2;  // CHECKs at cc:259-260,264-265
    f.sp1l = i1 - 1, f.sp1r = i1, f.sp2l = i2 - 1, f.sp2r = i2;
    if (w->times[f.sp1r] < w->times[f.sp2l])
      f.kind = 0;
    else if (f.sp1r == f.sp2l)
      f.kind = 1;
    else
      f.kind = 2;
    weight_and_normal(w->P, f.s1, f.s2, f.n, f.w);
    w->counts[f.kind]++;
    w->lidar.push_back(f);
  }
  return 0;
}

extern "C" int wco_window_add_unary(wco_window *w, const wc_surfel *fix_surf, const wc_pose *fix_pose,
                                    const wc_surfel *sld_surf, const wc_pose *sld_pose, const wc_pair *pairs,
                                    uint64_t n) {
  const int ns = (int)w->times.size();
  for (uint64_t k = 0; k < n; ++k) {
    LidarFactor f;
    f.kind = 3;
    f.s1 = make_ref(fix_surf[pairs[k].first], fix_pose[pairs[k].first]);
    f.s2 = make_ref(sld_surf[pairs[k].second], sld_pose[pairs[k].second]);
    if (!(f.s1.t < f.s2.t)) return 3;  // cc:301
    int i2 = upper_bound_idx(w->times, f.s2.t);
    if (i2 == 0 || i2 == ns) return 2;  // cc:304-305
    f.sp1l = f.sp1r = -1;
    f.sp2l = i2 - 1, f.sp2r = i2;
    weight_and_normal(w->P, f.s1, f.s2, f.n, f.w);
    w->counts[3]++;
    w->lidar.push_back(f);
  }
  return 0;
}

extern "C" int wco_window_add_imu(wco_window *w, const wc_imu_state *imu, uint64_t n_imu) {
  const int ns = (int)w->times.size();
  if (n_imu < 3 || ns < 2) return 0;
  for (uint64_t i = 0; i + 2 < n_imu; ++i) {
    const wc_imu_state &i1 = imu[i], &i3 = imu[i + 2];
    if (i1.t < w->times.front()) continue;  // cc:324-326
    if (i3.t > w->times.back()) break;      // cc:327-329
    int it = upper_bound_idx(w->times, i1.t);
    if (it == 0 || it == ns) return 2;
    ImuFactorRec f;
    f.i1 = imu[i], f.i2 = imu[i + 1], f.i3 = imu[i + 2];
    f.sp1 = it - 1, f.sp2 = it;
    if (it == ns - 1) {
      f.mode = 1, f.sp3 = -1;
    } else {
      f.mode = 0, f.sp3 = it + 1;
    }
    w->counts[4 + f.mode]++;
    w->imu.push_back(f);
  }
  return 0;
}

extern "C" uint64_t wco_window_num_residuals(const wco_window *w) { return w->lidar.size() + 12 * w->imu.size(); }
extern "C" void wco_window_counts(const wco_window *w, uint64_t counts[6]) {
  for (int i = 0; i < 6; ++i) counts[i] = w->counts[i];
}

extern "C" int wco_window_evaluate(const wco_window *w, const double *x, double *cost, double *residuals) {
  uint64_t o = 0;
  *cost = walk(*w, x, false, [&](const RowBlocks &rb) {
    if (residuals)
      for (int r = 0; r < rb.nres; ++r) residuals[o + r] = rb.r[r];
    o += rb.nres;
  });
  return 0;
}

extern "C" int wco_window_linearize(const wco_window *w, const double *x, double *H, double *g, double *cost) {
  std::vector<double> Hv, gv;
  linearize_dense(*w, x, Hv, gv, *cost);
  std::memcpy(H, Hv.data(), Hv.size() * sizeof(double));
  std::memcpy(g, gv.data(), gv.size() * sizeof(double));
  return 0;
}

extern "C" int wco_window_solve(const wco_window *w, double *x_io, wc_solve_summary *sum, double *first_step) {
  const wco_window &W = *w;
  const int n = 12 * (int)W.times.size();
  // active (non-constant) columns
  std::vector<int> act;
  for (int i = 0; i < n; ++i)
    if (!(W.fix_first_pos && i >= 3 && i < 6)) act.push_back(i);
  const int m = (int)act.size();

  std::vector<double> x(x_io, x_io + n), best_x = x, cand(n);
  std::vector<double> H, g, scale(m), Hs((size_t)m * m), gs(m), diag(m), A, y;
  double cost, radius = 1e4, decrease = 2.0, min_cost;
  bool reuse_diagonal = false;
  std::memset(sum, 0, sizeof(*sum));

  auto load_scaled = [&]() {
    for (int i = 0; i < m; ++i) {
      gs[i] = g[act[i]] * scale[i];
      for (int j = 0; j < m; ++j) Hs[(size_t)i * m + j] = H[(size_t)act[i] * n + act[j]] * scale[i] * scale[j];
    }
  };
  auto grad_max = [&]() {
    double mx = 0;
    for (int i = 0; i < m; ++i) mx = std::max(mx, std::fabs(g[act[i]]));
    return mx;
  };
  auto vnorm = [&](const std::vector<double> &v) {
    double s = 0;
    for (double e : v) s += e * e;
    return std::sqrt(s);
  };

  // iteration 0
  linearize_dense(W, x.data(), H, g, cost);
  sum->n_linearizations++;
  for (int i = 0; i < m; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(H[(size_t)act[i] * n + act[i]]));
  load_scaled();
  sum->initial_cost = cost;
  min_cost = cost;
  double x_norm = vnorm(x);
  int iter = 0;
  sum->termination = 1;
  bool first_recorded = false;
  int consecutive_invalid = 0;
  if (grad_max() <= 1e-10) {
    sum->termination = 0;
  } else {
    while (true) {
      if (iter >= W.P.max_iterations) {
        sum->termination = 1;
        break;
      }
      if (grad_max() <= 1e-10 || radius <= 1e-32) {
        sum->termination = 0;
        break;
      }
      ++iter;
      // LevenbergMarquardtStrategy::ComputeStep
      if (!reuse_diagonal)
        for (int i = 0; i < m; ++i) diag[i] = std::min(std::max(Hs[(size_t)i * m + i], 1e-6), 1e32);
      A = Hs;
      for (int i = 0; i < m; ++i) A[(size_t)i * m + i] += diag[i] / radius;
      y = gs;
      bool ok = chol_solve(A, y, m);
      double model_change = 0;
      if (ok) {
        // step = -y ; model_cost_change = -(J s).(r + J s / 2) = y.gs - y.Hs.y / 2
        double yg = 0, yHy = 0;
        for (int i = 0; i < m; ++i) {
          yg += y[i] * gs[i];
          double s = 0;
          for (int j = 0; j < m; ++j) s += Hs[(size_t)i * m + j] * y[j];
          yHy += y[i] * s;
        }
        model_change = yg - 0.5 * yHy;
      }
      if (!ok || !(model_change > 0)) {  // invalid step: StepIsInvalid
        // HandleInvalidStep: max_num_consecutive_invalid_steps = 5
        if (++consecutive_invalid >= 5) {
          sum->termination = 2;
          break;
        }
        radius *= 0.5;
        reuse_diagonal = true;
        sum->unsuccessful_steps++;
        continue;
      }
      consecutive_invalid = 0;
      cand = x;
      for (int i = 0; i < m; ++i) cand[act[i]] = x[act[i]] - y[i] * scale[i];
      if (!first_recorded) {
        first_recorded = true;
        if (first_step)
          for (int i = 0; i < n; ++i) first_step[i] = cand[i] - x[i];
        double s = 0;
        for (int i = 0; i < n; ++i) s += (cand[i] - x[i]) * (cand[i] - x[i]);
        sum->first_step[0] = std::sqrt(s);
      }
      double cand_cost;
      wco_window_evaluate(w, cand.data(), &cand_cost, nullptr);
      sum->n_cost_evaluations++;
      // ParameterToleranceReached
      double step_norm = 0;
      for (int i = 0; i < n; ++i) step_norm += (x[i] - cand[i]) * (x[i] - cand[i]);
      step_norm = std::sqrt(step_norm);
      if (step_norm <= 1e-8 * (x_norm + 1e-8)) {
        sum->termination = 0;
        break;
      }
      // FunctionToleranceReached
      double cost_change = cost - cand_cost;
      if (std::fabs(cost_change) <= 1e-6 * cost) {
        sum->termination = 0;
        break;
      }
      double rho = cost_change / model_change;
      if (rho > 1e-3) {  // HandleSuccessfulStep
        x = cand;
        x_norm = vnorm(x);
        linearize_dense(W, x.data(), H, g, cost);
        sum->n_linearizations++;
        load_scaled();
        radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3));
        radius = std::min(1e16, radius);
        decrease = 2.0;
        reuse_diagonal = false;
        sum->successful_steps++;
        if (cost < min_cost) {
          min_cost = cost;
          best_x = x;
        }
      } else {  // HandleUnsuccessfulStep
        radius = radius / decrease;
        decrease *= 2;
        reuse_diagonal = true;
        sum->unsuccessful_steps++;
      }
    }
  }
  sum->iterations = iter;
  sum->final_cost = min_cost;
  std::memcpy(x_io, best_x.data(), sizeof(double) * n);
  return 0;
}
