// window.hip — the sliding-window optimisation problem on gfx950: residual + Jacobian assembly, the block
// J^T J / J^T r reduction, cost evaluation and the Levenberg-Marquardt loop.
//
// Replaces, for the hot path, the reference's use of Ceres:
//   BuildSldWinLidarResiduals / SurfelMatchBinaryFactor<0/1/2>   lidar_odometry.cc:254-297, cost_functor.h:100-241
//   BuildFixWinLidarResiduals / SurfelMatchUnaryFactor           lidar_odometry.cc:299-317, cost_functor.h:16-69
//   BuildImuResiduals / ImuFactor<0/1>                           lidar_odometry.cc:319-363, cost_functor.h:264-472
//   problem.Evaluate (residual histograms)                       lidar_odometry.cc:56-94
//   ceres::Solve, SPARSE_NORMAL_CHOLESKY, <= 100 iterations, SubsetParameterization gauge   lidar_odometry.cc:551-561
// Factor formulas: SURVEY.md Appendix B (incl. quirks Q1, Q3 behind wc_params.reference_quirks).
//
// Data layout in HBM
//   * one packed record per correspondence, struct-of-arrays, SORTED by the pair of sample intervals it touches:
//       binary (136 B): n[3] w a1[3] a2[3] dp[3] f1 f2 + key      unary (96 B): n[3] w a2[3] d[3] f2 + key
//     (a_k = R_k c_k, dp = p1 - p2, d = c1_world - p2: everything that does not depend on the unknowns is folded in
//     once per solve; per LM iteration a record is read exactly once per pass.)
//   * records with one key form a segment; segments are cut into pieces of <= 256 records; one workgroup per piece
//     evaluates r and the 1 x 24 (1 x 12) Jacobian row of every record into LDS and reduces the piece's Gram matrix
//     [J r]^T [J r] in 4x4 register blocks over slices of the records, slices added in fixed order (no atomics, fixed
//     summation order => bitwise reproducible, which keeps replicated LM state in lock-step across ranks after the
//     all-reduce).
//   * a gather kernel sums the piece partials into the dense normal equations H (12 ns x 12 ns) and g through a CSR
//     source list built on the host once per solve.
//   * LM: Jacobi scaling, damping, blocked Cholesky (own kernels, fp64), back substitution, step and candidate-cost
//     evaluation all stay on the device; only a 6-double mailbox crosses PCIe per iteration.
// The assembly streams 136 B and issues ~0.9 kflop of fp64 per record: no MFMA there (rank-1 fp64 updates of 6-wide blocks;
// a 25-wide Gram matrix wastes most of a 32 x 32 tile).  The dense Cholesky of the damped system does its panel solves,
// trailing updates and the rank-4 updates of the diagonal factor with v_mfma_f64_16x16x4.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "ctx.h"
#include "dmath.h"
#include "so3_fused.h"

using namespace wc;

namespace {

#ifndef WC_PIECE
#define WC_PIECE 256
#endif
constexpr int kPiece = WC_PIECE;   // records per piece (= threads per assembly workgroup)
constexpr int kNB = 32;       // Cholesky block size
constexpr int kLinChunksU = 4;  // chunks of kPiece records a unary piece may hold (round 6)
constexpr uint32_t kHeavySrc = 24;  // block pairs with more gather sources than this get the multi-group gather

// ------------------------------------------------------------------------------------------------------------------
struct ImuRec {
  wc_imu_state i1, i2, i3;
  int sp1, mode;  // mode 0: blocks sp1, sp1+1, sp1+2; mode 1: sp1, sp1+1
};

struct Piece {
  uint32_t begin, count;  // record range (imu: factor range)
  uint32_t key;           // binary: sp1l | sp2l << 16, unary: sp2l, imu: sp1
  uint32_t part_off;      // offset of this piece's partial in the partial buffer (doubles)
};

struct Src {  // one contribution to a 12x12 block pair (I,J) of H
  uint32_t part_off;
  uint8_t p, q, w, T;  // local block indices, local block width (6 or 12), packed-triangle dimension
};
struct GSrc {
  uint32_t part_off;
  uint8_t p, w, T, pad;
};
struct FarJob {  // a block pair more than two sample blocks apart with at most kHeavySrc sources (surfel factors only: its 6 x 6 pose corner)
  uint32_t begin, end, pid;
  uint16_t I, J;
};

struct WinParams {
  double sigma0_sq, cauchy_b, inv_cauchy_b, w_gyr, w_acc, w_bg, w_ba, dt, grav[3];
  int quirks, ns, fix_first;
};

}  // namespace

struct wc_window_state {
  WinParams wp;
  int ns = 0, n = 0, np = 0, ld = 0;
  uint32_t nb = 0, nu = 0, ni = 0;
  uint32_t npiece_b = 0, npiece_u = 0, npiece_i = 0, npart_doubles = 0;
  uint32_t npairs = 0;
  std::vector<double> times;
  // device buffers
  wc_buf times_d, brec, bkey, borig, urec, ukey, uorig, irec, pieces, partial, src, src_begin, gsrc, gsrc_begin;
  wc_buf lin, lin_alt, Linv, heavy, near_l, far_l, Lmat, reduce;  // lin_alt: the linearisation at the LM candidate (see wc_window_solve)
  int lin_sel = 0;                                 // which of the two holds the linearisation at the current point
  uint32_t nheavy = 0, nnear = 0, nfar = 0;  // lin = [H (n*n) | g (np) | cost, spare]
  // multi-GPU: sharded = this problem holds one rank's share of the factors (wc_window_build_sharded, or a caller that shards
  // itself and installs wc_window_set_allreduce); only then are linearisation and cost evaluation collectives.  pair_off[pid] =
  // offset of block pair pid in the reduction buffer: 144 doubles for a pair of sample blocks at most two apart (IMU factors
  // reach that far, cost_functor.h:264-355), 36 - the pose x pose corner - for the others (surfel factors only, :16-179)
  bool sharded = false;
  uint32_t npiece_b_big = 0;  // binary pieces of more than kPiece / 2 records (a prefix of the family): the others are paired in k_lin_fused
  bool unary_multi = false;  // unary pieces hold several chunks of kPiece records: the family is a launch of its own (k_lin_surfel<12, true, true>)
  uint32_t lin_count = 0;  // linearisations enqueued in the two-collective form (parity of the late max |g| slot)
  // the large collective + k_expand_corners on a stream of their own, beside the bias elimination (ordered by events; joined in front of
  // whatever reads pose blocks or writes the buffers again)
  hipStream_t side = nullptr;
  hipEvent_t ev_side_go = nullptr, ev_side_done = nullptr;
  bool side_pending = false;
  bool two_coll = false;  // sharded by wc_window_build_sharded with the IMU factors replicated: two collectives per linearisation
  wc_buf pcr_D[2], pcr_A[2], pcr_R[2], yred;  // bias elimination by parallel cyclic reduction (window_schur.inc)
  wc_buf pair_off;
  uint32_t red_H = 0;  // doubles of the reduction buffer in front of {g (np), cost, spare}
  int (*allreduce)(void *, double *, uint64_t) = nullptr;
  void *allreduce_user = nullptr;
  wc_buf x, xc, scale, diag, A, y, mail, cost_part, keys_tmp[2], vals_tmp[2], heads, status;
  // pinned staging of the two families' segment heads + status words, and the events that say a family's copy has landed
  void *h_pin = nullptr;
  size_t h_pin_cap = 0;
  wc_buf lists;          // device arena of the lists the host builds (irec, pair_off, heavy, pieces, src, src_begin, gsrc, gsrc_begin are views into it)
  void *h_up = nullptr;  // pinned staging of that arena
  size_t h_up_cap = 0;
  bool status_clear = false;  // W->status is zero at rest
  uint32_t heads_zero = 0;    // entries of W->heads known to be zero (the segment heads by key: zero at rest)
  hipEvent_t fam_done[2] = {nullptr, nullptr};  // [0]: segment heads in pinned memory; [1]: the build's last upload has left its staging buffer
  bool built = false;
};

namespace {

// ---- device-side factor arithmetic ---------------------------------------------------------------------------------
__device__ __forceinline__ V3 ld3(const double *p) { return mk3(p[0], p[1], p[2]); }

// Explicitly fused helpers for the factor evaluation (the file is built with -ffp-contract=off for the gate decisions of
// the extraction; here the results are held to 1e-10 relative, and the fp64 instruction count is what bounds phase A).
__device__ __forceinline__ double dotf(V3 a, V3 b) { return fma(a.x, b.x, fma(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ V3 crossf(V3 a, V3 b) {
  return mk3(fma(a.y, b.z, -(a.z * b.y)), fma(a.z, b.x, -(a.x * b.z)), fma(a.x, b.y, -(a.y * b.x)));
}
__device__ __forceinline__ V3 lerp3(const double *l, const double *r, double g, double f) {
  return mk3(fma(f, r[0], g * l[0]), fma(f, r[1], g * l[1]), fma(f, r[2], g * l[2]));
}
// unit quaternion (w, u): R v = v + w t + u x t with t = 2 u x v;  R^T v = v - w t + u x t
__device__ __forceinline__ V3 qrotf(double w, V3 u, V3 v) {
  V3 t = crossf(u, v);
  t = t + t;
  const V3 ut = crossf(u, t);
  return mk3(fma(w, t.x, v.x) + ut.x, fma(w, t.y, v.y) + ut.y, fma(w, t.z, v.z) + ut.z);
}

// one surfel side: correction interpolated between two sample blocks (cost_functor.h:124-136), rotated lever arm
// and, if wanted, the 1x6 Jacobian w.r.t. the interpolated (rot_cor, pos_cor)  (cost_functor.h:147-150, :162-165)
__device__ __forceinline__ void surfel_side(const double *xl, const double *xr, double f, V3 a, V3 wn, double sign,
                                            V3 &rotated_plus_t, double j[6], bool want_jac) {
  const double g = 1 - f;
  const V3 r = lerp3(xl, xr, g, f);
  const V3 t = lerp3(xl + 3, xr + 3, g, f);
  // Exp(r) and Jr(r) from ONE sincos of the half angle (sin th = 2 s c, 1 - cos th = 2 s^2): evaluating so3_exp and so3_Jr
  // separately costs four fp64 sin / cos per side, and this kernel is bound by exactly that arithmetic (fp64 vector rate),
  // not by its 136 B per record.  The Jacobian row  wn^T Exp(r) Hat(a) Jr(r)  is evaluated right to left as vectors
  // (u = Exp(r)^T wn, v = u x a, row = sn v + (1 - sn)(v . an) an + omc (v x an)) instead of two 3x3 matrix products:
  // a third of the operations.  Differs from the matrix form by a few ulp.
  const double th2 = dotf(r, r);
  double qw, sn, omc, ith;
  V3 qu;
  if (th2 < 1e-10 * 1e-10) {  // Jr = I (utils.h:47), Exp by its series (so3.hpp:705-712)
    const Q4 E = so3_exp(r);
    qw = E.w, qu = mk3(E.x, E.y, E.z);
    sn = 1.0, omc = 0.0, ith = 0.0;
  } else {
    ith = rsqrt_nr(th2);
    const double th = th2 * ith;
    double sh, ch;
    sincos_half(0.5 * th, &sh, &ch);
    const double imag = sh * ith;
    qw = ch, qu = imag * r;
    const double s2 = (sh + sh) * ith;
    sn = s2 * ch, omc = s2 * sh;
  }
  rotated_plus_t = qrotf(qw, qu, a) + t;
  if (want_jac) {
    const V3 an = (-ith) * r;  // unit axis of -r (Jr(r) = Jl(-r), utils.h:46-58)
    const V3 u = qrotf(-qw, qu, wn);
    const V3 v = crossf(u, a);
    const V3 c = crossf(v, an);
    const double k = (1 - sn) * dotf(v, an);
    j[0] = sign * fma(sn, v.x, fma(k, an.x, omc * c.x));
    j[1] = sign * fma(sn, v.y, fma(k, an.y, omc * c.y));
    j[2] = sign * fma(sn, v.z, fma(k, an.z, omc * c.z));
    j[3] = -sign * wn.x, j[4] = -sign * wn.y, j[5] = -sign * wn.z;
  }
}

// ceres::CauchyLoss(a): rho(s) = b log(1 + s/b), b = a^2; returns rho, sets sqrt(rho') (Corrector with rho'' < 0)
// (1 / b comes with the parameters and sqrt(1 / sum) is ONE reciprocal square root with a Newton step: the division, the
// reciprocal and the square root of the literal form were ~70 fp64 instructions per record)
__device__ __forceinline__ double cauchy(double b, double inv_b, double s, double &sqrt_rho1) {
  const double sum = fma(s, inv_b, 1.0);
  sqrt_rho1 = fmax(1.4916681462400413e-154 /* sqrt(DBL_MIN) */, rsqrt_nr(sum));
  return b * log(sum);
}

// Evaluate one binary record -> loss-corrected residual, cost contribution, and (optionally) the padded 24-wide
// Jacobian row in the reference's parameter-block layout including the overwrite quirk (Q1).
__device__ __forceinline__ void eval_binary(const WinParams &wp, const double *rec, uint32_t nb, uint32_t k, uint32_t key,
                                            const double *x, double &r_out, double &cost, double *v /*24 or null*/) {
  const int sp1l = (int)(key & 0xFFFF), sp2l = (int)(key >> 16);
  const V3 n = mk3(rec[0 * (size_t)nb + k], rec[1 * (size_t)nb + k], rec[2 * (size_t)nb + k]);
  const double w = rec[3 * (size_t)nb + k];
  const V3 a1 = mk3(rec[4 * (size_t)nb + k], rec[5 * (size_t)nb + k], rec[6 * (size_t)nb + k]);
  const V3 a2 = mk3(rec[7 * (size_t)nb + k], rec[8 * (size_t)nb + k], rec[9 * (size_t)nb + k]);
  const V3 dp = mk3(rec[10 * (size_t)nb + k], rec[11 * (size_t)nb + k], rec[12 * (size_t)nb + k]);
  const double f1 = rec[13 * (size_t)nb + k], f2 = rec[14 * (size_t)nb + k];
  const V3 wn = w * n;
  V3 s1, s2;
  double j1[6], j2[6];
  surfel_side(x + 12 * sp1l, x + 12 * (sp1l + 1), f1, a1, wn, -1.0, s1, j1, v != nullptr);
  surfel_side(x + 12 * sp2l, x + 12 * (sp2l + 1), f2, a2, wn, +1.0, s2, j2, v != nullptr);
  double r = w * dotf(n, (s1 + dp) - s2);  // cost_functor.h:140
  double sc;
  cost = 0.5 * cauchy(wp.cauchy_b, wp.inv_cauchy_b, r * r, sc);
  r_out = r * sc;
  if (!v) return;
  const int mode = (sp2l > sp1l + 1) ? 0 : (sp2l == sp1l + 1 ? 1 : 2);
  // local slots of (sp1l, sp1r, sp2l, sp2r) among the distinct blocks (DispatchPtr, cost_functor.h:216-229): side 1 sits
  // in slots 0 / 1, side 2 in 2 / 3 (mode 0), 1 / 2 (mode 1) or 0 / 1 (mode 2).  Written with static indices and selects:
  // indexing v[] with the slot number puts the whole row into scratch memory (208 B per lane).
  const double w1l = sc * (1 - f1), w1r = sc * f1, w2l = sc * (1 - f2), w2r = sc * f2;  // corrector folded in
  // The reference's dispatch (DispatchPtr, cost_functor.h:216-229, with the later write winning: Q1) as a BLEND: every entry of
  // the row is j1 * A + j2 * B with weights that depend on the mode alone (zero where a side does not reach a slot; x + 0 = x,
  // so the values are those of the assignments).  As nested selects per entry the dispatch was ~200 v_cndmask per record; as
  // branches per mode the row went to scratch memory.
  const bool q = wp.quirks != 0;
  const double A0 = (mode == 2 && q) ? 0.0 : w1l, B0 = mode == 2 ? w2l : 0.0;
  const double A1 = (mode != 0 && q) ? 0.0 : w1r, B1 = mode == 0 ? 0.0 : (mode == 1 ? w2l : w2r);
  const double B2 = mode == 0 ? w2l : (mode == 1 ? w2r : 0.0), B3 = mode == 0 ? w2r : 0.0;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    v[c] = fma(j2[c], B0, j1[c] * A0);
    v[6 + c] = fma(j2[c], B1, j1[c] * A1);
    v[12 + c] = j2[c] * B2;
    v[18 + c] = j2[c] * B3;
  }
}

__device__ __forceinline__ void eval_unary(const WinParams &wp, const double *rec, uint32_t nu, uint32_t k, uint32_t key,
                                           const double *x, double &r_out, double &cost, double *v /*12 or null*/) {
  const int sp2l = (int)key;
  const V3 n = mk3(rec[0 * (size_t)nu + k], rec[1 * (size_t)nu + k], rec[2 * (size_t)nu + k]);
  const double w = rec[3 * (size_t)nu + k];
  const V3 a2 = mk3(rec[4 * (size_t)nu + k], rec[5 * (size_t)nu + k], rec[6 * (size_t)nu + k]);
  const V3 d = mk3(rec[7 * (size_t)nu + k], rec[8 * (size_t)nu + k], rec[9 * (size_t)nu + k]);
  const double f2 = rec[10 * (size_t)nu + k];
  const V3 wn = w * n;
  V3 s2;
  double j2[6];
  surfel_side(x + 12 * sp2l, x + 12 * (sp2l + 1), f2, a2, wn, +1.0, s2, j2, v != nullptr);
  double r = w * dotf(n, d - s2);  // cost_functor.h:39
  double sc;
  cost = 0.5 * cauchy(wp.cauchy_b, wp.inv_cauchy_b, r * r, sc);
  r_out = r * sc;
  if (!v) return;
  const double wl = sc * (1 - f2), wr = sc * f2;
  for (int c = 0; c < 6; ++c) {
    v[c] = j2[c] * wl;
    v[6 + c] = j2[c] * wr;
  }
}

// interpolated state at an IMU timestamp (ComputeStateCorr, cost_functor.h:358-400)
struct StateCorr {
  V3 r, t, bg, ba;
  int bl;  // local left block (0 or 1); right = bl + 1
  double f;
};
__device__ __forceinline__ StateCorr state_corr(const double *x, const double *times, int sp1, int mode, double t) {
  StateCorr c;
  const bool first = (mode == 1) ? true : (t >= times[sp1] && t < times[sp1 + 1]);
  c.bl = first ? 0 : 1;
  const int gl = sp1 + c.bl;
  const double *l = x + 12 * gl, *r = x + 12 * (gl + 1);
  c.f = (t - times[gl]) / (times[gl + 1] - times[gl]);
  c.r = (1 - c.f) * ld3(l) + c.f * ld3(r);
  c.t = (1 - c.f) * ld3(l + 3) + c.f * ld3(r + 3);
  c.bg = (1 - c.f) * ld3(l + 6) + c.f * ld3(r + 6);
  c.ba = (1 - c.f) * ld3(l + 9) + c.f * ld3(r + 9);
  return c;
}
__device__ __forceinline__ M3 Ffun(Q4 L, Q4 E, Q4 R, const M3 &Jr) {  // cost_functor.h:446-448
  M3 Jri;
  log_jr_inv(qmul(qmul(L, E), R), &Jri);
  return (Jri * qmat(qconj(R))) * Jr;
}

// IMU factor: 12 residuals; if rows != null also the 12 x 36 Jacobian, written as rows[row * stride + col]
// (cost_functor.h:272-355).  ROLE < 0: all of it.  ROLE 0 .. 3: the share of one of k_lin_imu's four wavefronts - the factor is
// a single thread's chain of dependent fp64 operations (20.7 k clocks), and different code only runs side by side in
// different wavefronts: 0 = residuals, 1 = F1 and its term of the group (0, 0), 2 = F2 (handed back in `late`: its term is added
// to the group after a barrier, the same two operations in the same order), 3 = every other group.  Every value is formed by the expressions of the full version.
struct ImuLate {
  M3 F2;
  double w2[3];
};
template <int ROLE>
__device__ void eval_imu(const WinParams &wp, const ImuRec &f, const double *x, const double *times, double res[12],
                         double *rows, int stride, ImuLate *late = nullptr) {
  constexpr bool kAll = ROLE < 0;
  const double dt = wp.dt;
  const StateCorr c1 = state_corr(x, times, f.sp1, f.mode, f.i1.t);
  const StateCorr c2 = state_corr(x, times, f.sp1, f.mode, f.i2.t);
  const StateCorr c3 = state_corr(x, times, f.sp1, f.mode, f.i3.t);
  const Q4 R1{f.i1.quat[0], f.i1.quat[1], f.i1.quat[2], f.i1.quat[3]};
  const Q4 R2{f.i2.quat[0], f.i2.quat[1], f.i2.quat[2], f.i2.quat[3]};
  const ExpJr X1 = exp_jr(c1.r), X2 = exp_jr(c2.r);
  const Q4 E1R1 = qmul(X1.E, R1), E2R2 = qmul(X2.E, R2);
  if (kAll || ROLE == 0) {
    const V3 p1 = ld3(f.i1.pos), p2 = ld3(f.i2.pos), p3 = ld3(f.i3.pos);
    const V3 gyr_est = log_jr_inv(qmul(qconj(E1R1), E2R2), nullptr) / dt;
    const V3 acc_est = (((c3.t + p3) + (c1.t + p1)) - 2 * (c2.t + p2)) / (dt * dt);
    const V3 grav = mk3(wp.grav[0], wp.grav[1], wp.grav[2]);
    const V3 r0 = wp.w_gyr * (((ld3(f.i1.gyr) + ld3(f.i2.gyr)) / 2 - gyr_est) - c1.bg);
    const V3 r1 = wp.w_acc * ((qrot(E1R1, ld3(f.i1.acc) - c1.ba) - acc_est) + grav);
    const V3 r2 = wp.w_bg * (c1.bg - c2.bg);
    const V3 r3 = wp.w_ba * (c1.ba - c2.ba);
    res[0] = r0.x, res[1] = r0.y, res[2] = r0.z, res[3] = r1.x, res[4] = r1.y, res[5] = r1.z;
    res[6] = r2.x, res[7] = r2.y, res[8] = r2.z, res[9] = r3.x, res[10] = r3.y, res[11] = r3.z;
  }
  if (!rows) return;
  // tau Jacobians (cost_functor.h:301-321) scattered with (1 - f), f onto the bracketing blocks (:402-444).  The caller has
  // zeroed the rows; every 3x3 group is summed in registers over the states that contribute to it (in the reference's
  // order: state 1, 2, 3) and stored once per block - a read-modify-write per contribution on LDS rows serialises ~230
  // dependent round trips per factor.
  double w1[3], w2[3], w3[3];
  for (int b = 0; b < 3; ++b) {
    w1[b] = c1.bl == b ? 1 - c1.f : (c1.bl + 1 == b ? c1.f : 0.0);
    w2[b] = c2.bl == b ? 1 - c2.f : (c2.bl + 1 == b ? c2.f : 0.0);
    w3[b] = c3.bl == b ? 1 - c3.f : (c3.bl + 1 == b ? c3.f : 0.0);
  }
  auto group = [&](int r0_, int c0_, const M3 &ma, double sa, const M3 *mb, double sb, const M3 *mc, double sc_) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        const double va = sa * ma.m[i][j], vb = mb ? sb * mb->m[i][j] : 0.0, vc = mc ? sc_ * mc->m[i][j] : 0.0;
        for (int b = 0; b < 3; ++b) {
          double acc = va * w1[b];
          if (mb) acc += vb * w2[b];
          if (mc) acc += vc * w3[b];
          rows[(r0_ + i) * stride + b * 12 + c0_ + j] = acc;
        }
      }
  };
  const M3 I = m3_identity();
  if (kAll) {
    const M3 F1 = Ffun(qconj(R1), X1.E, E2R2, X1.Jr), F2 = Ffun(qconj(E1R1), X2.E, R2, X2.Jr);
    group(0, 0, F1, wp.w_gyr * (1 / dt), &F2, -wp.w_gyr * (1 / dt), nullptr, 0.0);
  } else if (ROLE == 1) {  // acc = va * w1[b]; wavefront 2 adds vb * w2[b] behind the barrier
    const M3 F1 = Ffun(qconj(R1), X1.E, E2R2, X1.Jr);
    group(0, 0, F1, wp.w_gyr * (1 / dt), nullptr, 0.0, nullptr, 0.0);
  } else if (ROLE == 2) {
    late->F2 = Ffun(qconj(E1R1), X2.E, R2, X2.Jr);
    for (int b = 0; b < 3; ++b) late->w2[b] = w2[b];
  }
  if (kAll || ROLE == 3) group(0, 6, I, -wp.w_gyr, wp.quirks ? &I : nullptr, -wp.w_gyr, nullptr, 0.0);  // Q3 (cost_functor.h:314)
  if (kAll || ROLE == 3) group(3, 0, (qmat(X1.E) * hat(qrot(R1, ld3(f.i1.acc) - c1.ba))) * X1.Jr, -wp.w_acc, nullptr, 0.0, nullptr, 0.0);
  if (kAll || ROLE == 3) group(3, 3, I, -wp.w_acc * (1 / dt / dt), &I, wp.w_acc * (2 / dt / dt), &I, -wp.w_acc * (1 / dt / dt));
  if (kAll || ROLE == 3) group(3, 9, qmat(E1R1), -wp.w_acc, nullptr, 0.0, nullptr, 0.0);
  if (kAll || ROLE == 3) group(6, 6, I, wp.w_bg, &I, -wp.w_bg, nullptr, 0.0);
  if (kAll || ROLE == 3) group(9, 9, I, wp.w_ba, &I, -wp.w_ba, nullptr, 0.0);
}

__device__ __forceinline__ uint32_t tri_index(uint32_t i, uint32_t j, uint32_t T) {  // i <= j, row-major upper
  return i * T - (i * (i - 1)) / 2 + (j - i);
}

// ---- record construction (once per solve) ---------------------------------------------------------------------------
__device__ __forceinline__ int upper_bound_times(const double *times, int ns, double t) {
  int lo = 0, hi = ns;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (t < times[mid])
      hi = mid;
    else
      lo = mid + 1;
  }
  return lo;
}

// One family of surfel correspondences as the record kernels see it (binary: both surfels of the sliding window; unary: s1 / p1 the
// fixed window).  Both families go through ONE chain of launches - keys, sort, records, segment heads - since round 4's last
// part: two chains of ~120 us of small launches each were most of wc_window_build's 0.40 ms in the odometry step.
struct FamArgs {
  const wc_surfel *s1, *s2;
  const wc_pose *p1, *p2;
  const wc_pair *pairs;
  uint32_t n;
  double *rec;
  uint32_t *key_out, *orig_out;
};

// sort keys of the correspondences: (sp1l, sp2l) from std::upper_bound on the sample timestamps (cc:258-268,:303-307).  Binary
// keys sp1l * ns + sp2l < ns^2, unary keys ns^2 + sp2l: the sorted array holds the binary family first, each family in the order
// a sort of its own would give (the radix sort is stable, vals ascend).  A flagged record takes its family's key 0.
__global__ void __launch_bounds__(256) k_pair_keys(FamArgs B, FamArgs U, const double *times, int ns, uint32_t *keys, uint32_t *vals,
                                                  uint32_t *status) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= B.n + U.n) return;
  const bool unary = k >= B.n;
  const uint32_t kl = unary ? k - B.n : k;
  const wc_pair pr = unary ? U.pairs[kl] : B.pairs[kl];
  const double t1 = (unary ? U.s1 : B.s1)[pr.first].t, t2 = (unary ? U.s2 : B.s2)[pr.second].t;
  if (!(t1 < t2)) atomicOr(&status[1], 2u);  // CHECK_LT (cc:256,:301)
  const int i2 = upper_bound_times(times, ns, t2);
  int i1 = 1;
  if (!unary) i1 = upper_bound_times(times, ns, t1);
  const uint32_t base = unary ? (uint32_t)ns * (uint32_t)ns : 0u;
  if (i2 == 0 || i2 == ns || i1 == 0 || i1 == ns) {
    atomicOr(&status[1], 1u);
    keys[k] = base;
  } else {
    keys[k] = base + (unary ? (uint32_t)(i2 - 1) : ((uint32_t)(i1 - 1) * (uint32_t)ns + (uint32_t)(i2 - 1)));
  }
  vals[k] = k;
}

__device__ __forceinline__ void surfel_world(const wc_surfel &s, const wc_pose &p, V3 &a, V3 &pos, M3 &cov_w) {
  const Q4 q{p.quat[0], p.quat[1], p.quat[2], p.quat[3]};
  a = qrot(q, ld3(s.center));
  pos = ld3(p.pos);
  M3 C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C.m[i][j] = s.cov[3 * i + j];
  const M3 R = qmat(q);
  cov_w = (R * C) * transpose(R);  // GetCovarianceInWorld (surfel.h:89-91)
}

// packed records in sorted order; ctor arithmetic of both surfel factors (cost_functor.h:21-25, :109-113)
__global__ void __launch_bounds__(256) k_build_records(FamArgs B, FamArgs U, const uint32_t *sorted_idx, const uint32_t *sorted_keys,
                                                      const double *times, int ns, double sigma0_sq) {
  const uint32_t kg = blockIdx.x * blockDim.x + threadIdx.x;
  if (kg >= B.n + U.n) return;
  const bool unary = kg >= B.n;  // (every binary key sorts in front of every unary key)
  const FamArgs F = unary ? U : B;
  const uint32_t k = unary ? kg - B.n : kg;
  const uint32_t o = sorted_idx[kg] - (unary ? B.n : 0u);
  const wc_pair pr = F.pairs[o];
  const wc_surfel A = F.s1[pr.first], Bs = F.s2[pr.second];
  V3 a1, pos1, a2, pos2;
  M3 c1, c2;
  surfel_world(A, F.p1[pr.first], a1, pos1, c1);
  surfel_world(Bs, F.p2[pr.second], a2, pos2, c2);
  double ev[3];
  M3 V;
  eig3_sym(c1 + c2, ev, V);
  const double w = 1 / sqrt(sigma0_sq + ev[0]);
  const uint32_t key = sorted_keys[kg] - (unary ? (uint32_t)ns * (uint32_t)ns : 0u);
  const size_t N = F.n;
  double *rec = F.rec;
  rec[0 * N + k] = V.m[0][0], rec[1 * N + k] = V.m[1][0], rec[2 * N + k] = V.m[2][0];
  rec[3 * N + k] = w;
  if (!unary) {
    const int sp1l = (int)(key / (uint32_t)ns), sp2l = (int)(key % (uint32_t)ns);
    rec[4 * N + k] = a1.x, rec[5 * N + k] = a1.y, rec[6 * N + k] = a1.z;
    rec[7 * N + k] = a2.x, rec[8 * N + k] = a2.y, rec[9 * N + k] = a2.z;
    const V3 dp = pos1 - pos2;
    rec[10 * N + k] = dp.x, rec[11 * N + k] = dp.y, rec[12 * N + k] = dp.z;
    rec[13 * N + k] = (A.t - times[sp1l]) / (times[sp1l + 1] - times[sp1l]);
    rec[14 * N + k] = (Bs.t - times[sp2l]) / (times[sp2l + 1] - times[sp2l]);
    F.key_out[k] = (uint32_t)sp1l | ((uint32_t)sp2l << 16);
  } else {
    const int sp2l = (int)key;
    rec[4 * N + k] = a2.x, rec[5 * N + k] = a2.y, rec[6 * N + k] = a2.z;
    const V3 d = (a1 + pos1) - pos2;  // c1_world - p2
    rec[7 * N + k] = d.x, rec[8 * N + k] = d.y, rec[9 * N + k] = d.z;
    rec[10 * N + k] = (Bs.t - times[sp2l]) / (times[sp2l + 1] - times[sp2l]);
    F.key_out[k] = (uint32_t)sp2l;
  }
  F.orig_out[k] = o;
}

// segment heads of a sorted key array, BY KEY: heads[key] = position + 1 of the first record with that key (0 = no such record).
// The array is zero at rest (the host clears it behind its read-back) and the host walks it in key order - which is position order,
// the keys being sorted: no append counter on the device, no sort of the heads on the host (rounds 1 - 4: an unordered append of
// (position, key) pairs, one atomic per workgroup, and a std::sort of the ~4 000 pairs on the host - half of the ~75 us the host
// needs between the heads' arrival and the upload of its lists).
__global__ void __launch_bounds__(1024) k_seg_heads(const uint32_t *keys, uint32_t n, uint32_t *heads) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t key = keys[k];
  if (k == 0 || key != keys[k - 1]) heads[key] = k + 1u;
}

// ---- assembly: one workgroup per piece ------------------------------------------------------------------------------
// (tried: workgroups of 256 / 128 / 64 threads by the size of the piece - a third of C4's binary pieces holds <= 64 records.
// No gain: phase A is bound by its fp64 arithmetic per WAVEFRONT, and a small piece already keeps only one wavefront busy.
// Round 3, -DWC_PIECE=128: pieces of <= 128 records, 128 threads and half the LDS each, six workgroups per CU instead of
// three - 0.182 ms per linearisation at C4 against 0.178 with 256 (19 879 pieces against 12 299): not the pieces in flight
// either.)
// Phase A: thread k evaluates record k of the piece (residual + W-wide Jacobian row) into LDS (row k of V = [J r]).
// Phase B: the Gram matrix V^T V ((W+1)^2, packed upper triangle; the corner (W,W) carries the piece's cost instead of
//          sum r^2) in 4x4 register blocks: thread = (block of the upper block triangle, slice of the records); per
//          record it reads 8 values and does 16 products.  One thread per output entry (2 LDS reads per product) made
//          this phase LDS-bandwidth bound at ~190 us per linearisation; an fp64-MFMA variant is no faster (fp64 MFMA has
//          the vector rate and a 25-wide Gram wastes 58 % of 32x32 tiles).  The slices are added in fixed order: no
//          atomics, bitwise reproducible.  Round 4 built the matrix-core form again, in 16 x 16 tiles (v_mfma_f64_16x16x4: one
//          double per lane and operand, a quarter of the LDS traffic; a wavefront takes every fourth group of four records, two sets
//          of accumulators, operands requested a step ahead): correct (all parity tests), and the same 0.160 ms per linearisation in
//          an A/B on one box - 48 matrix instructions of 64 clocks per wavefront are the 8.5 k clocks the register blocks took, three
//          tiles for 325 of their 768 entries.
// where entry e of a piece's packed upper triangle sits in a slice's partial blocks (offset in doubles; 0xFFFF: the corner, which
// carries the piece's cost): a table instead of the closed-form inversion of e -> (i, j) - a square root, two correction steps and
// the block arithmetic per entry, in the tail every piece pays
template <int W>
struct LinTailTable {
  static constexpr int T = W + 1, NB = (T + 3) / 4, PB = 17, NOUT = T * (T + 1) / 2;
  uint16_t off[NOUT];
  constexpr LinTailTable() : off() {
    int e = 0;
    for (int i = 0; i < T; ++i)
      for (int j = i; j < T; ++j, ++e) {
        const int ti = i >> 2, tj = j >> 2;
        const int q = ti * NB - ti * (ti - 1) / 2 + (tj - ti);
        off[e] = (i == W) ? (uint16_t)0xFFFF : (uint16_t)(q * PB + (i & 3) * 4 + (j & 3));
      }
  }
};
__constant__ const LinTailTable<24> kLinTail24{};
__constant__ const LinTailTable<12> kLinTail12{};

// LDS of a surfel piece in doubles: V (later the per-slice partial blocks) + the wavefronts' cost sums.
// Round 5: the rows of a binary piece pass through LDS in ROUNDS of kPiece / RND rows (every thread keeps its row in registers
// until its round comes: 50 VGPRs) and the slices' partial blocks are folded in groups of NSH - a full piece took 53 KB, which
// capped a CU at three pieces = 12 wavefronts of dependent fp64 chains (the kernel is bound by how many such
// chains a CU interleaves, not by their instruction count).
#ifndef WC_LIN_ROUNDS
#define WC_LIN_ROUNDS 2
#endif
#ifndef WC_LIN_WG_PER_CU
#define WC_LIN_WG_PER_CU 4
#endif
template <int W>
struct LinSurfelLds {
  static constexpr int T = W + 1;
  static constexpr int NB = (T + 3) / 4;            // 4-column blocks of V
  static constexpr int NBLK = NB * (NB + 1) / 2;    // blocks (bi <= bj) of the Gram matrix
  static constexpr int NS = kPiece / NBLK;          // record slices
  static constexpr int TS = T + (T & 1);            // row stride in LDS: even, so that a row's 4-column blocks are 16-byte aligned (ds_read_b128)
#ifdef WC_LIN_RH
  static constexpr int RH = W == 24 ? WC_LIN_RH : kPiece;  // rows per round
#else
  static constexpr int RND = W == 24 ? WC_LIN_ROUNDS : 1;  // rounds of rows through LDS
  static constexpr int RH = kPiece / RND;           // rows per round
#endif
  static constexpr int VSZ = RH * TS + 4;           // + 4: the padded columns of the last block read past the last row
  static constexpr int PB = 17;                     // doubles per partial block in LDS: 16 + 1 (a stride of 16 doubles puts every second lane on the same banks)
  // (four pieces per CU leave 40 KB each: the partial blocks of all slices fit - 34.3 KB - and are not folded; with V's 26.7 KB
  // as the budget, nine slices in folds of 7 + 2, the kernel took 4 us longer at C4: a barrier and a read-modify-write round more)
#ifndef WC_LIN_LDS_DOUBLES
#define WC_LIN_LDS_DOUBLES 4300
#endif
  static constexpr int BUD = WC_LIN_LDS_DOUBLES > VSZ ? WC_LIN_LDS_DOUBLES : VSZ;   // doubles the partial blocks may take
  static constexpr int NSH = BUD / (NBLK * PB) > NS ? NS : BUD / (NBLK * PB);  // slices per fold of the partial blocks (they take V's storage)
  static_assert(NSH >= 1, "a fold of the partial blocks fits V's storage");
  static constexpr int PSZ = NSH * NBLK * PB;
  static constexpr int VMAX = VSZ > PSZ ? VSZ : PSZ;
  static constexpr int DOUBLES = VMAX + 4;
};
template <int W, bool UNARY, bool MULTI = false>
__device__ __forceinline__ void lin_surfel_body(const WinParams &wp, const Piece pc, const double *rec, uint32_t nrec, const double *x,
                                                double *partial, double *smem /* 16-byte aligned, LinSurfelLds<W>::DOUBLES */,
                                                uint32_t cost_slot /* the piece's entry of the cost array behind the partials */) {
  using L = LinSurfelLds<W>;
  constexpr int T = L::T, NB = L::NB, NBLK = L::NBLK, NS = L::NS, TS = L::TS, PB = L::PB, RH = L::RH, NSH = L::NSH;
  double *sV = smem, *sC = smem + L::VMAX;
  const int tid = threadIdx.x;
  // where this thread's output entries sit in the partial blocks: requested first (a table look-up in the tail was a global-memory
  // round trip on every piece's critical path, behind the last barrier)
  constexpr int NOUT = T * (T + 1) / 2, NTE = (NOUT + kPiece - 1) / kPiece;
  static_assert(W == 24 || W == 12, "tail tables exist for the binary and the unary family");
  int toff[NTE];
  {
    const uint16_t *tail_off = W == 24 ? kLinTail24.off : kLinTail12.off;
#pragma unroll
    for (int i = 0; i < NTE; ++i) toff[i] = tid + i * kPiece < NOUT ? (int)tail_off[tid + i * kPiece] : -1;
  }
#if defined(WC_LIN_KNOCK) && (WC_LIN_KNOCK & 16)  // timing knock-out: the launch alone
  if (pc.count < 100000) return;
#endif
#ifdef WC_PROF_LIN
  long long lt_[6];
  lt_[0] = clock64();
#endif
  double c = 0.0;
#ifdef WC_PROF_LIN
  long long la_[4] = {0, 0, 0, 0};
  asm volatile("" ::"s"(pc.count), "s"(pc.begin));
  la_[0] = clock64();  // descriptor has arrived
#endif
  double acc[4][4] = {{0.0}};
  const int blk = tid % NBLK, slice = tid / NBLK;
  int bi = 0, remb = blk;
  while (remb >= NB - bi) {
    remb -= NB - bi;
    ++bi;
  }
  const int bj = bi + remb;
  const double *pi = sV + 4 * bi, *pj = sV + 4 * bj;
  // Round 6: a piece may hold SEVERAL chunks of kPiece records (the unary family: up to kLinChunksU; a unary key of C4 has ~7.9 k
  // records, i.e. 31 pieces of 256 with 31 descriptors, 31 exchanges of partial blocks, 31 tails and 31 gather sources for the same
  // two sample blocks).  The workgroup evaluates a chunk, passes its rows through LDS, adds their Gram blocks onto the SAME register
  // accumulators, and goes on with the next chunk: partial blocks, tail and output once per piece.
  // (MULTI only in k_lin_surfel's unary instantiation, a launch of its own with the full register file: inside k_lin_fused - 128 VGPRs at
  // four workgroups per CU - the loop spilled 50 - 170 registers and the kernel took 112 us instead of 87 at C4)
  constexpr uint32_t kMaxCh = (UNARY && MULTI) ? (uint32_t)kLinChunksU : 1u;
#pragma unroll 1
  for (uint32_t chi = 0; chi < kMaxCh; ++chi) {
  const uint32_t ch0 = chi * (uint32_t)kPiece;
  if (chi && ch0 >= pc.count) break;
  const int ccount = (int)min((uint32_t)kPiece, pc.count - ch0);  // records of this chunk (uniform over the workgroup)
  if (chi) __syncthreads();  // (the rows of the chunk before have been read)
  double v[W], r = 0.0;
  if (tid < ccount) {
    const uint32_t k = pc.begin + ch0 + tid;
    constexpr int NF = UNARY ? 11 : 15;
    double rv[NF];
#if defined(WC_LIN_KNOCK) && (WC_LIN_KNOCK & 4)  // timing knock-out: no record loads
#pragma unroll
    for (int f = 0; f < NF; ++f) rv[f] = 1e-3 * (double)(k + f);
#else
#pragma unroll
    for (int f = 0; f < NF; ++f) rv[f] = rec[(size_t)f * nrec + k];
#endif
#ifdef WC_PROF_LIN
    asm volatile("" ::"v"(rv[0]), "v"(rv[NF - 1]), "v"(rv[5]));
    la_[1] = clock64();  // records have arrived
#endif
#if defined(WC_LIN_KNOCK) && (WC_LIN_KNOCK & 2)  // timing knock-out: no evaluation (the loads stay)
    for (int i = 0; i < W; ++i) v[i] = rv[i % NF];
    r = rv[1], c = rv[2];
#else
    double cc_ = 0.0;  // (a thread's costs of all chunks are added in chunk order)
    // (the sample blocks' corrections are read again in every chunk: hoisted out of the chunk loop the compiler kept their 24 doubles
    // in VGPRs across it and spilled ~18 of them - k_lin_fused 87 -> 112 us at C4; the pointer is made opaque per chunk)
    const double *xq = x;
    if (UNARY)
      eval_unary(wp, rv, 1, 0, pc.key, xq, r, cc_, v);  // one key per piece: the sample blocks are wave-uniform
    else
      eval_binary(wp, rv, 1, 0, pc.key, xq, r, cc_, v);
    c += cc_;
#endif
  }
#ifdef WC_PROF_LIN
  lt_[1] = clock64();
  lt_[2] = lt_[1];
#endif
  // the rows go through LDS in rounds of RH (the chunk's count is uniform over the workgroup: so are the rounds and their barriers)
#pragma unroll 1
  for (int h = 0; h * RH < ccount; ++h) {
    if (h) __syncthreads();  // the round before has been read
#if defined(WC_LIN_KNOCK) && (WC_LIN_KNOCK & 32)  // timing knock-out: no row writes
    if (tid < ccount && tid / RH == h && v[0] == 12345.0) {
#else
    if (tid < ccount && tid / RH == h) {
#endif
      const int row = tid - h * RH;
#pragma unroll
      for (int i = 0; i < W; ++i) sV[row * TS + i] = v[i];
      sV[row * TS + W] = r;
    }
    __syncthreads();
    if (slice < NS) {
      // the slices partition the round's rows [0, cnt); columns past T (last block) belong to the next row (or, for the last
      // row, to unwritten storage): those products land in accumulator entries that are never read
      const int cnt = min(ccount - h * RH, RH);
      const int sl = ((cnt + NS - 1) / NS) | 1;  // odd: the slices that share a wavefront then start on different banks
#if defined(WC_LIN_KNOCK) && (WC_LIN_KNOCK & 1)  // timing knock-out: no Gram loop
      const int k0 = min(slice * sl, cnt), k1 = min(k0 + 1, cnt);
#else
      const int k0 = min(slice * sl, cnt), k1 = min(k0 + sl, cnt);
#endif
      for (int k = k0; k < k1; ++k) {
        double a[4], c4[4];
        {
          const double2 a01 = *(const double2 *)(pi + k * TS), a23 = *(const double2 *)(pi + k * TS + 2);
          const double2 c01 = *(const double2 *)(pj + k * TS), c23 = *(const double2 *)(pj + k * TS + 2);
          a[0] = a01.x, a[1] = a01.y, a[2] = a23.x, a[3] = a23.y;
          c4[0] = c01.x, c4[1] = c01.y, c4[2] = c23.x, c4[3] = c23.y;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[p][q] = fma(a[p], c4[q], acc[p][q]);
      }
    }
  }
  }  // (chunks)
  // cost of the piece: fixed shuffle tree per wavefront, the four wavefront sums are added in the tail
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d, 64);
  if ((tid & 63) == 0) sC[tid >> 6] = c;
#ifdef WC_PROF_LIN
  lt_[3] = clock64();
#endif
#if defined(WC_LIN_KNOCK) && (WC_LIN_KNOCK & 64)  // timing knock-out: no partial blocks, no tail
  if (acc[0][0] != 12345.0) return;
#endif
  // everybody is done with V: its storage takes the partial blocks, NSH slices at a time (slice s is added onto slot s % NSH in
  // the order of s: fixed, bitwise reproducible)
#pragma unroll 1
  for (int base = 0; base < NS; base += NSH) {
    __syncthreads();
    if (slice >= base && slice < base + NSH && slice < NS) {
      double *dst = sV + ((slice - base) * NBLK + blk) * PB;
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[p * 4 + q] = base ? dst[p * 4 + q] + acc[p][q] : acc[p][q];
    }
  }
  __syncthreads();
#if defined(WC_LIN_KNOCK) && (WC_LIN_KNOCK & 8)  // timing knock-out: one output entry per piece
  for (int i = 0; i < 1; ++i) {
    const int e = tid + i * kPiece, off = e < 1 ? toff[i] : -1;
#else
#pragma unroll
  for (int i = 0; i < NTE; ++i) {
    const int e = tid + i * kPiece, off = toff[i];
#endif
    if (off < 0) continue;
    double out = 0.0;
    if (off == 0xFFFF) {
      out = kPiece == 256 ? (sC[0] + sC[1]) + (sC[2] + sC[3]) : kPiece == 128 ? sC[0] + sC[1] : sC[0];
      partial[cost_slot] = out;  // (k_gather's cost workgroup reads the costs as ONE contiguous array)
    } else {
#pragma unroll
      for (int sl = 0; sl < NSH; ++sl) out += sV[sl * NBLK * PB + off];
    }
    partial[pc.part_off + e] = out;
  }
#ifdef WC_PROF_LIN
  lt_[4] = clock64();
  if (tid == 0 && (blockIdx.x % 997) == 500)
    printf("lin W=%d blk %u count %u: A %lld (descriptor %lld records %lld evaluate %lld) sync %lld B %lld tail %lld total %lld\n", W, blockIdx.x,
           pc.count, lt_[1] - lt_[0], la_[0] - lt_[0], la_[1] - la_[0], lt_[1] - la_[1], lt_[2] - lt_[1], lt_[3] - lt_[2], lt_[4] - lt_[3],
           lt_[4] - lt_[0]);
#endif
}
// Round 6: TWO binary pieces of at most 128 records each in one workgroup.  A binary key of C4 holds ~125 records: as a
// piece of its own it keeps half a workgroup idle through the evaluation and pays a descriptor round trip and a launch slot for it.  Here
// wavefronts 0 - 1 evaluate piece A's records and wavefronts 2 - 3 piece B's at the same time (the key is wave-uniform: readfirstlane keeps
// the sample blocks' corrections in scalar loads); then, piece by piece, exactly what lin_surfel_body does for a piece of <= 128 records:
// its rows through LDS (one round), the Gram blocks by (block pair, slice), partial blocks, tail, cost = its two wavefronts' sums.  Same
// slices, same order of every sum: the same bits as two workgroups.  The pieces keep their own output and cost slots: k_gather does not
// know the difference.
__device__ __forceinline__ void lin_binary_pair_body(const WinParams &wp, const Piece pcA, const Piece pcB /* count 0: none */, const double *rec,
                                                     uint32_t nrec, const double *x, double *partial, double *smem, uint32_t cost_slotA,
                                                     uint32_t cost_slotB) {
  constexpr int W = 24;
  using L = LinSurfelLds<W>;
  constexpr int T = L::T, NB = L::NB, NBLK = L::NBLK, NS = L::NS, TS = L::TS, PB = L::PB, RH = L::RH, NSH = L::NSH;
  static_assert(RH == kPiece / 2 && kPiece == 256, "a piece of the pair is one round of rows");
  double *sV = smem, *sC = smem + L::VMAX;
  const int tid = threadIdx.x;
  constexpr int NOUT = T * (T + 1) / 2, NTE = (NOUT + kPiece - 1) / kPiece;
  int toff[NTE];
#pragma unroll
  for (int i = 0; i < NTE; ++i) toff[i] = tid + i * kPiece < NOUT ? (int)kLinTail24.off[tid + i * kPiece] : -1;
  const bool second = tid >= RH;
  const int row = tid & (RH - 1);
  const uint32_t my_count = (uint32_t)__builtin_amdgcn_readfirstlane((int)(second ? pcB.count : pcA.count));
  const uint32_t my_begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)(second ? pcB.begin : pcA.begin));
  const uint32_t my_key = (uint32_t)__builtin_amdgcn_readfirstlane((int)(second ? pcB.key : pcA.key));
  double c = 0.0;
  double v[W], r = 0.0;
  if (row < (int)my_count) {
    const uint32_t k = my_begin + (uint32_t)row;
    double rv[15];
#pragma unroll
    for (int f = 0; f < 15; ++f) rv[f] = rec[(size_t)f * nrec + k];
    eval_binary(wp, rv, 1, 0, my_key, x, r, c, v);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d, 64);
  if ((tid & 63) == 0) sC[tid >> 6] = c;
  const int blk = tid % NBLK, slice = tid / NBLK;
  int bi = 0, remb = blk;
  while (remb >= NB - bi) {
    remb -= NB - bi;
    ++bi;
  }
  const int bj = bi + remb;
  const double *pi = sV + 4 * bi, *pj = sV + 4 * bj;
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    const uint32_t cnt_u = h ? pcB.count : pcA.count, part_off = h ? pcB.part_off : pcA.part_off, cslot = h ? cost_slotB : cost_slotA;
    if (cnt_u == 0u) break;  // (uniform: no second piece)
    if (h) __syncthreads();  // the first piece's tail has read its partial blocks
    if ((tid >= RH) == (h == 1) && row < (int)cnt_u) {
#pragma unroll
      for (int i = 0; i < W; ++i) sV[row * TS + i] = v[i];
      sV[row * TS + W] = r;
    }
    __syncthreads();
    double acc[4][4] = {{0.0}};
    if (slice < NS) {
      const int cnt = (int)cnt_u;
      const int sl = ((cnt + NS - 1) / NS) | 1;
      const int k0 = min(slice * sl, cnt), k1 = min(k0 + sl, cnt);
      for (int k = k0; k < k1; ++k) {
        double a[4], c4[4];
        {
          const double2 a01 = *(const double2 *)(pi + k * TS), a23 = *(const double2 *)(pi + k * TS + 2);
          const double2 c01 = *(const double2 *)(pj + k * TS), c23 = *(const double2 *)(pj + k * TS + 2);
          a[0] = a01.x, a[1] = a01.y, a[2] = a23.x, a[3] = a23.y;
          c4[0] = c01.x, c4[1] = c01.y, c4[2] = c23.x, c4[3] = c23.y;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[p][q] = fma(a[p], c4[q], acc[p][q]);
      }
    }
#pragma unroll 1
    for (int base = 0; base < NS; base += NSH) {
      __syncthreads();
      if (slice >= base && slice < base + NSH && slice < NS) {
        double *dst = sV + ((slice - base) * NBLK + blk) * PB;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q) dst[p * 4 + q] = base ? dst[p * 4 + q] + acc[p][q] : acc[p][q];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NTE; ++i) {
      const int e = tid + i * kPiece, off = toff[i];
      if (off < 0) continue;
      double out = 0.0;
      if (off == 0xFFFF) {
        out = h ? (sC[2] + sC[3]) + 0.0 : (sC[0] + sC[1]) + 0.0;  // (a piece alone: (its two wavefronts' sums) + (0 + 0) of the idle ones)
        partial[cslot] = out;
      } else {
#pragma unroll
        for (int sl = 0; sl < NSH; ++sl) out += sV[sl * NBLK * PB + off];
      }
      partial[part_off + e] = out;
    }
  }
}

template <int W, bool UNARY, bool MULTI = false>
__global__ void __launch_bounds__(kPiece) k_lin_surfel(WinParams wp, const Piece *pieces, const double *rec, const uint32_t *keys,
                                                      uint32_t nrec, const double *x, double *partial, uint32_t cost_slot0) {
  __shared__ __attribute__((aligned(16))) double smem[LinSurfelLds<W>::DOUBLES];
  lin_surfel_body<W, UNARY, MULTI>(wp, pieces[blockIdx.x], rec, nrec, x, partial, smem, cost_slot0 + blockIdx.x);
}

using f64x4 = __attribute__((ext_vector_type(4))) double;

// Entries (i <= j) of the IMU piece's 37 x 37 Gram matrix in the order of their row support.  A Jacobian column of class
// rot / pos / bg / ba (column % 12 / 3) is non-zero only in the residual groups {gyr, acc} / {acc} / {gyr, bg} / {acc, ba}
// (the group() calls of eval_imu: cost_functor.h:301-321), the residual column 36 in all four; bit g of the support of an
// entry = residual group g (rows 3 g .. 3 g + 2 of every factor) contributes.  ent = i | j << 6 | support << 12.
struct ImuGramOrder {
  uint16_t ent[37 * 38 / 2];
  constexpr ImuGramOrder() : ent() {
    constexpr uint32_t sup[5] = {0x3u, 0x2u, 0x5u, 0xAu, 0xFu};
    int n = 0;
    for (uint32_t want = 0; want < 16; ++want)  // entries of equal support next to each other
      for (int i = 0; i < 37; ++i)
        for (int j = i; j < 37; ++j) {
          const uint32_t m = sup[i == 36 ? 4 : (i % 12) / 3] & sup[j == 36 ? 4 : (j % 12) / 3];
          if (m == want) ent[n++] = (uint16_t)((uint32_t)i | ((uint32_t)j << 6) | (m << 12));
        }
  }
};
__constant__ const ImuGramOrder kImuGram{};

// IMU factors of one sample interval: <= kImuMax factors x 12 residual rows, 36-wide Jacobian
constexpr int kImuMax = 8;  // (16: 30 us at C4, 8: 24 us - one round of 252 workgroups, 4: 37 us - two rounds)
constexpr int kLinImuLds = kImuMax * 12 * 37 + kImuMax;  // doubles: the pieces' rows + the factors' costs
__device__ __forceinline__ void lin_imu_body(const WinParams &wp, const Piece pc, const ImuRec *recs, const double *x, const double *times,
                                             double *partial, double *smem /* kLinImuLds */, uint32_t cost_slot) {
  constexpr int T = 37;
  double *sV = smem, *sC = smem + kImuMax * 12 * T;
  const int tid = threadIdx.x;
  for (int e = tid; e < (int)pc.count * 12 * T; e += 256) sV[e] = 0.0;  // eval_imu stores the non-zero 3x3 groups only
  __syncthreads();
  {  // the factor's evaluation, one share per wavefront (see eval_imu); lane = factor
    const int wv = tid >> 6, ln = tid & 63;
    const bool act = ln < (int)pc.count;
    ImuLate late;
    if (act) {
      const ImuRec &f = recs[pc.begin + ln];
      double *rows = &sV[ln * 12 * T];
      double res[12];
      if (wv == 0) {
        eval_imu<0>(wp, f, x, times, res, rows, T);
        double c = 0;
        for (int r = 0; r < 12; ++r) {
          rows[r * T + 36] = res[r];
          c += res[r] * res[r];
        }
        sC[ln] = 0.5 * c;  // TrivialLoss
      } else if (wv == 1) {
        eval_imu<1>(wp, f, x, times, res, rows, T);
      } else if (wv == 2) {
        eval_imu<2>(wp, f, x, times, res, rows, T, &late);
      } else {
        eval_imu<3>(wp, f, x, times, res, rows, T);
      }
    }
    __syncthreads();
    if (act && wv == 2) {  // group (0, 0): + (sb F2[i][j]) w2[b] onto wavefront 1's va w1[b]
      double *rows = &sV[ln * 12 * T];
      const double sb = -wp.w_gyr * (1 / wp.dt);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          const double vb = sb * late.F2.m[i][j];
          for (int b = 0; b < 3; ++b) rows[i * T + b * 12 + j] += vb * late.w2[b];
        }
    }
  }
  __syncthreads();
  // Gram matrix of the piece's rows.  Every entry is summed over the rows in row order (the fp64 matrix cores do this
  // phase in a fifth of the time, but sum in groups of four: the facade parity test, which amplifies last-bit differences
  // over 16 sweeps, then leaves its 5e-6 band).  Only the row groups in which BOTH columns can be non-zero are visited
  // (a skipped term is + 0.0 * 0.0; see ImuGramOrder): 3.4 of 12 rows on average, and the entries are handed out in the
  // order of their support, so that the lanes of a wavefront skip the same groups.
  constexpr int NOUT = T * (T + 1) / 2;
  for (int round = 0; round < (NOUT + 255) / 256; ++round) {
    // (the table is ordered by support = by cost: the 64-entry groups go to the wavefronts in a snake, light and heavy alternating)
    const int wv = tid >> 6, e0 = ((round & 1) ? 4 * round + (3 - wv) : 4 * round + wv) * 64 + (tid & 63);
    if (e0 >= NOUT) continue;
    const uint32_t ent = kImuGram.ent[e0];
    const int i = (int)(ent & 63u), j = (int)((ent >> 6) & 63u);
    const uint32_t m = ent >> 12;
    double acc = 0.0;
    if (i == 36) {
      for (uint32_t k = 0; k < pc.count; ++k) acc += sC[k];
      partial[cost_slot] = acc;
    } else {
      for (int k = 0; k < (int)pc.count; ++k) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          if ((m >> g4) & 1u) {
            const double *ra = &sV[(k * 12 + 3 * g4) * T];
            const double a0 = ra[i], a1 = ra[T + i], a2 = ra[2 * T + i], b0 = ra[j], b1 = ra[T + j], b2 = ra[2 * T + j];
            acc += a0 * b0;
            acc += a1 * b1;
            acc += a2 * b2;
          }
      }
    }
    partial[pc.part_off + tri_index((uint32_t)i, (uint32_t)j, (uint32_t)T)] = acc;
  }
}
__global__ void __launch_bounds__(256) k_lin_imu(WinParams wp, const Piece *pieces, const ImuRec *recs, const double *x,
                                                const double *times, double *partial, uint32_t cost_slot0) {
  __shared__ double smem[kLinImuLds];
  lin_imu_body(wp, pieces[blockIdx.x], recs, x, times, partial, smem, cost_slot0 + blockIdx.x);
}

// ALL families of a linearisation in one launch (round 3): workgroups [0, n_imu) take the IMU pieces - dispatched first: a few
// hundred workgroups of dependent fp64 chains, 14.5 us as a launch of their own whatever the window -, then the binary pieces,
// then the unary ones.  The binary pieces' 53 KB of LDS already cap a CU at three workgroups, which is what the IMU body's 168
// VGPRs allow: the fused kernel's occupancy is the binary kernel's.  Measured, alternating on one box: IMU inside the binary
// launch 0.177 -> 0.166 - 0.172 ms per linearisation at C4 (1 775 -> 1 795 - 1 820 LM it/s), odometry-step solve 2.42 -> 2.30 ms;
// the unary pieces inside too (at the binary pieces' LDS: three workgroups per CU instead of four, but no drain between the
// families): solve 2.30 -> 2.25 ms, C4 unchanged.
// (k_lin_imu's workgroups have 256 threads: with -DWC_PIECE below 256, an experiment, the families are launches of their own)
// OCC = workgroups per CU the register budget is cut for: 4 (128 VGPRs - the IMU body, a latency chain of a few hundred workgroups,
// then spills 47 registers) when the surfel pieces outnumber the chip's workgroup slots, 3 (168 VGPRs, no spills) for the small windows
// of a real stream, where the IMU chain IS the kernel (facade: 15.7 -> 13.5 us)
// n_big: binary pieces of more than 128 records (the family is sorted by size, largest first): a workgroup each; the others go two to a
// workgroup (lin_binary_pair_body).  n_bwg = n_big + ceil((n_b - n_big) / 2) workgroups for the binary family (n_big = n_b: no pairing).
template <bool WITH_UNARY, int OCC>
__global__ void __launch_bounds__(256, OCC) k_lin_fused(WinParams wp, const Piece *pieces, uint32_t n_imu, uint32_t n_b, uint32_t n_u, const double *brec,
                                                     uint32_t nb, const double *urec, uint32_t nu, const ImuRec *irec, const double *times,
                                                     const double *x, double *partial, uint32_t cost_slot0, uint32_t n_big, uint32_t n_bwg) {
  constexpr int SZ0 = LinSurfelLds<24>::DOUBLES > kLinImuLds ? LinSurfelLds<24>::DOUBLES : kLinImuLds;
  constexpr int SZ = LinSurfelLds<12>::DOUBLES > SZ0 ? LinSurfelLds<12>::DOUBLES : SZ0;
  __shared__ __attribute__((aligned(16))) double smem[SZ];
  const uint32_t b = blockIdx.x;
  if (b < n_imu) {
    lin_imu_body(wp, pieces[n_b + n_u + b], irec, x, times, partial, smem, cost_slot0 + n_b + n_u + b);
  } else if (b < n_imu + n_bwg) {
    const uint32_t q = b - n_imu;
    if (q < n_big) {
      lin_surfel_body<24, false>(wp, pieces[q], brec, nb, x, partial, smem, cost_slot0 + q);
    } else {
      const uint32_t p0 = n_big + 2u * (q - n_big), p1 = p0 + 1u;
      Piece none = pieces[p0];
      none.count = 0u;
      lin_binary_pair_body(wp, pieces[p0], p1 < n_b ? pieces[p1] : none, brec, nb, x, partial, smem, cost_slot0 + p0, cost_slot0 + p1);
    }
  } else if (WITH_UNARY) {
    const uint32_t q = n_b + (b - n_imu - n_bwg);
    lin_surfel_body<12, true>(wp, pieces[q], urec, nu, x, partial, smem, cost_slot0 + q);
  }
}

// Gather of the piece partials into the dense normal equations, g and the cost: ONE launch with four roles by workgroup
// index (four dependent launches of latency-bound kernels cost 110 us per linearisation; run side by side they cost what
// the longest role does).  Workgroup = 7 groups of 144 threads (one thread per entry of a 12x12 block pair):
//   heavy  - block pairs with more than kHeavySrc sources (the diagonal band, hundreds each): one pair per workgroup, the
//            groups stride over the source list, partial sums combined in fixed order;
//   light  - the other pairs (at most a few sources): one pair per group;
//   g      - one sample block per workgroup, 84 groups of 12 lanes stride over the block's source list;
//   cost   - the last workgroup adds the cost slots of all partials.
// Every source costs two dependent loads (descriptor, value): four (heavy pairs: sixteen) sources are in flight per thread, added in list order
// (bitwise reproducible, no atomics).
constexpr int kGG = 7;
constexpr int kLightSets = 2;  // block pairs per group of a light gather workgroup
constexpr int kFarGroups = 144 * kGG / 36, kFarSets = 2;  // 36-lane groups of a far-pair workgroup, pairs per group
constexpr int kHeavyIlp = 8;   // sources in flight per thread of a heavy pair (with the next eight descriptors: 48 VGPRs -
                               // above 64 only ONE 1008-thread workgroup fits a CU; 16 in flight measured no faster)
struct GatherArgs {
  const Src *src;
  const uint32_t *src_begin;
  const GSrc *gsrc;
  const uint32_t *gsrc_begin;
  const Piece *pieces;
  const uint32_t *heavy, *near;  // pairs with long source lists; the other pairs at most two sample blocks apart
  const FarJob *far;
  const double *partial;
  double *H, *g, *cost;
  uint32_t nheavy, nnear, nfar, npairs, npieces, nb_pieces, nu_pieces, cost_base;
  int ns, fix_first;
  int packed;  // H = the multi-GPU reduction buffer: block pairs in pair order at pair_off[pid] (144 doubles, or the 6 x 6 pose
               // corner of a pair more than two sample blocks apart); else the dense n x n matrix
  const uint32_t *pair_off;
  // Round 6, the two-collective form of a sharded window: the IMU factors are REPLICATED on every rank, the surfel factors
  // sharded; a linearisation gathers twice -
  //   only = 1 (with packed): sources of SURFEL pieces alone (local block width 6) -> the 6 x 6 pose corner of EVERY pair at 36 pid, the
  //            pose half of g at 36 npairs + 6 I, the surfel pieces' cost: what the ranks sum (one all-reduce, needed by k_schur_form);
  //   only = 2 (dense): sources of IMU pieces alone (width 12) -> near / heavy pairs of H, g, the IMU pieces' cost: complete on every
  //            rank without a collective, and all the bias elimination reads
  // only = 0: every source (one GPU, or the one-collective form).
  int only;
  // post != 0: the last of the ns + 1 workgroups that form g and the cost also forms max |g| and stores the mailbox (what
  // k_post_reduce does as a launch of its own): `done` counts them, and is left at zero
  int post, mail_slot;
  uint32_t *done;
  double *mail, *host_mail;
  unsigned long long ticket;
};

// Sum of the sources s0, s0 + STRIDE, ... of one entry (u, v) of a block pair, in list order.  A source costs two dependent
// loads (descriptor, value); the descriptors of the NEXT trip of ILP sources are requested right behind the values of this
// one (they only depend on the list position), so a list of n sources is 1 + n / ILP round trips deep instead of 2 n / ILP:
// the kernel lasts as long as its longest heavy pair (350 sources, 50 per group), 39.7 us of 41 by per-workgroup stamps.
template <int STRIDE, int ILP>
__device__ __forceinline__ double gather_pair_sum(const Src *src, const double *partial, uint32_t s0, uint32_t eend, int u, int v,
                                                  double acc = 0.0, uint32_t want_w = 0u /* 6 / 12: sources of that block width only */) {
  const uint2 *src2 = (const uint2 *)src;
  static_assert(sizeof(Src) == 8, "descriptor = two words");
  uint2 d[ILP];
#pragma unroll
  for (int q = 0; q < ILP; ++q) {
    const uint32_t sq = s0 + q * STRIDE;
    d[q] = src2[sq < eend ? sq : s0];
  }
  for (uint32_t s = s0; s < eend; s += ILP * STRIDE) {
    double val[ILP];
    bool ok[ILP];
#pragma unroll
    for (int q = 0; q < ILP; ++q) {
      const uint32_t sq = s + q * STRIDE;
      const bool live = sq < eend;
      const uint32_t part_off = d[q].x, sp = d[q].y & 0xFFu, sq_ = (d[q].y >> 8) & 0xFFu, sw = (d[q].y >> 16) & 0xFFu, sT = d[q].y >> 24;
      ok[q] = live && (uint32_t)u < sw && (uint32_t)v < sw && (want_w == 0u || sw == want_w);
      uint32_t r = sp * sw + u, c = sq_ * sw + v;
      if (r > c) {
        const uint32_t t = r;
        r = c, c = t;
      }
      val[q] = partial[part_off + (ok[q] ? tri_index(r, c, sT) : 0u)];
    }
    const uint32_t sn = s + ILP * STRIDE;
    if (sn < eend) {
#pragma unroll
      for (int q = 0; q < ILP; ++q) {
        const uint32_t sq = sn + q * STRIDE;
        d[q] = src2[sq < eend ? sq : sn];
      }
    }
#pragma unroll
    for (int q = 0; q < ILP; ++q)
      if (ok[q]) acc += val[q];
  }
  return acc;
}

// k_post_reduce inside k_gather: called by every thread of the ns + 1 workgroups that wrote g / the cost; local_max (thread 0) =
// max |g| over the entries this workgroup wrote (0 for the cost workgroup).  The host's mailbox leaves while the block-pair
// workgroups of the launch are still summing.  No fence and no second look at g: a workgroup publishes its maximum with an atomic
// maximum on the bits of the (non-negative) double - mail[61], zero at rest - and the cost workgroup stores the cost atomically,
// both at agent scope and acknowledged (vmcnt) before the workgroup is counted; the workgroup that counts last reads the two words
// back atomically.  (Rounds 3 - 4: a __threadfence() in every workgroup, another in the last one, which then read all of g again -
// two L2 write-backs of ~3.5 us each and a round of loads at the end of every linearisation.)
__device__ __forceinline__ void gather_post(const GatherArgs &a, double local_max) {
  __shared__ uint32_t s_last;
  const int tid = threadIdx.x;
  unsigned long long *gm = (unsigned long long *)(a.mail + 61);
  if (tid == 0) {
    if (local_max > 0.0)  // (a NaN entry is skipped, as fmax did)
      (void)__hip_atomic_fetch_max(gm, (unsigned long long)__double_as_longlong(local_max), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = atomicAdd(a.done, 1u) == (uint32_t)a.ns ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last || tid >= 64) return;
  const double mx = __longlong_as_double((long long)__hip_atomic_load(gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  const double cost = __hip_atomic_load(&a.cost[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a.host_mail && tid < 40) a.host_mail[tid] = (tid == a.mail_slot) ? cost : (tid == a.mail_slot + 1 ? mx : a.mail[tid]);
  if (tid == 0) {
    a.mail[a.mail_slot] = cost;
    a.mail[a.mail_slot + 1] = mx;
  }
  if (a.host_mail && a.ticket) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the wavefront's 40 stores to the pinned mailbox
    if (tid == 0) {
      __threadfence_system();
      __hip_atomic_store((unsigned long long *)(a.host_mail + 48), a.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (tid == 0) {
    *a.done = 0u;
    __hip_atomic_store(gm, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void __launch_bounds__(144 * kGG, 8) k_gather(GatherArgs a) {
  __shared__ double sred[kGG * 144], sred2[kGG * 144];
  const int tid = threadIdx.x;
  const uint32_t nlight = (a.nnear + kGG * kLightSets - 1) / (kGG * kLightSets);
  const uint32_t nfarwg = (a.nfar + kFarGroups * kFarSets - 1) / (kFarGroups * kFarSets);
  // dispatch order: g and the cost first (few workgroups with the longest chains of dependent loads), then the heavy pairs,
  // then the rest - dispatched last they only started when everything else had drained (47 us instead of ~30)
  const uint32_t nfirst = (uint32_t)a.ns + 1u;
  uint32_t blk = blockIdx.x < nfirst ? a.nheavy + nlight + nfarwg + blockIdx.x : blockIdx.x - nfirst;
  if (blk >= a.nheavy + nlight && blk < a.nheavy + nlight + nfarwg) {
    // FAR pairs (round 5): 28 groups of 36 lanes, one thread per entry of the pair's 6 x 6 pose corner - the 144-lane groups above
    // kept 108 lanes idle on them and wrote the 108 zeros of the rest of the block (and of its mirror) on every linearisation: three
    // quarters of H's 18.6 MB at C4.  Those entries are zero from the build (window_build_impl clears both linearisation buffers)
    // and nobody writes them.  Sources in list order, as everywhere.
#ifdef WC_GATHER_KNOCK
    if (WC_GATHER_KNOCK & 2) return;
#endif
    if (a.only == 2) return;  // (far pairs have surfel sources only)
    const int grp = tid / 36, e = tid % 36, u = e / 6, v = e % 6;
    const uint2 *src2 = (const uint2 *)a.src;
    const int n = 12 * a.ns;
    FarJob job[kFarSets];
    bool live[kFarSets];
    uint2 d[kFarSets][4];
#pragma unroll
    for (int k = 0; k < kFarSets; ++k) {
      const uint32_t idx = ((blk - a.nheavy - nlight) * kFarSets + k) * kFarGroups + grp;
      live[k] = idx < a.nfar;
      job[k] = a.far[live[k] ? idx : 0u];
    }
#pragma unroll
    for (int k = 0; k < kFarSets; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) d[k][q] = src2[(live[k] && job[k].begin + q < job[k].end) ? job[k].begin + q : 0u];
    double val[kFarSets][4];
#pragma unroll
    for (int k = 0; k < kFarSets; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t sp = d[k][q].y & 0xFFu, sq_ = (d[k][q].y >> 8) & 0xFFu, sT = d[k][q].y >> 24;
        val[k][q] = a.partial[d[k][q].x + tri_index(sp * 6u + (uint32_t)u, sq_ * 6u + (uint32_t)v, sT)];  // (p < q: row < column)
      }
#pragma unroll
    for (int k = 0; k < kFarSets; ++k) {
      double acc = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (live[k] && job[k].begin + q < job[k].end) acc += val[k][q];
      for (uint32_t s0 = job[k].begin + 4u; live[k] && s0 < job[k].end; s0 += 4u) {  // (rare: more than four sources)
        uint2 dd[4];
        double vv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) dd[q] = src2[s0 + q < job[k].end ? s0 + q : s0];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t sp = dd[q].y & 0xFFu, sq_ = (dd[q].y >> 8) & 0xFFu, sT = dd[q].y >> 24;
          vv[q] = a.partial[dd[q].x + tri_index(sp * 6u + (uint32_t)u, sq_ * 6u + (uint32_t)v, sT)];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (s0 + q < job[k].end) acc += vv[q];
      }
      const int I = job[k].I, J = job[k].J, gi = I * 12 + u, gj = J * 12 + v;
      if (a.fix_first && ((gi >= 3 && gi < 6) || (gj >= 3 && gj < 6))) acc = 0.0;  // SubsetParameterization(12,{3,4,5})
      if (a.packed) {
        if (live[k]) a.H[a.only ? (size_t)36 * job[k].pid + e : (size_t)a.pair_off[job[k].pid] + e] = acc;
        continue;
      }
      __syncthreads();  // (the previous set's mirror has been read)
      sred[tid] = acc;
      __syncthreads();
      if (live[k] && job[k].begin < job[k].end) {  // (a far pair without sources stays what the build made it: zero)
        a.H[(size_t)gi * n + gj] = acc;
        a.H[(size_t)(J * 12 + u) * n + I * 12 + v] = sred[grp * 36 + v * 6 + u];  // the mirror block, row-wise too
      }
    }
    return;
  }
  if (blk >= a.nheavy + nlight) blk -= nfarwg;
  if (blk < a.nheavy + nlight) {
    const bool heavy = blk < a.nheavy;
#ifdef WC_GATHER_KNOCK  // timing knock-outs (results wrong on purpose): 1 no heavy pairs, 2 no light pairs, 4 no g, 8 no cost
    if ((WC_GATHER_KNOCK & 1) && heavy) return;
    if ((WC_GATHER_KNOCK & 2) && !heavy) return;
#endif
    const int grp = tid / 144, e = tid % 144;
    const int u = e / 12, v = e % 12;
    const uint32_t want_w = a.only == 1 ? 6u : (a.only == 2 ? 12u : 0u);
    // A heavy workgroup sums ONE pair; a light one kLightSets x kGG pairs, kLightSets per group, whose first four sources
    // are requested together: a light pair is a chain of three round trips (list bounds, descriptors, values) and hardly any
    // arithmetic, and with one pair per group the 1 162 light workgroups of C4 queued for the 512 workgroup slots of the chip
    // behind the heavy ones (median start 21 us into a 35 us kernel, per-workgroup stamps).  Two sets: with four the kernel
    // needs more than 64 VGPRs (or spills) and loses the second workgroup per CU.
    uint32_t pidk[kLightSets];
    double acck[kLightSets];
    bool writek[kLightSets];
    if (heavy) {
      // Round 5: the pair's SURFEL sources (6 x 6 corners: hundreds on the diagonal band) go to 28 groups of 36 lanes, its IMU sources
      // (12 x 12, a handful, last in the list) to the seven groups of 144 - with 144 lanes per source three quarters of the lanes
      // idled and 350 sources were seven dependent trips of eight per group; now two.  Partial sums combined in fixed order.
      const uint32_t pid = a.heavy[2 * blk], mid = a.heavy[2 * blk + 1];
      const uint32_t b = a.src_begin[pid], eend = a.src_begin[pid + 1];
      const int g36 = tid / 36, e36 = tid % 36;
      const double acc6 = a.only == 2 ? 0.0 : gather_pair_sum<kFarGroups, kHeavyIlp>(a.src, a.partial, b + g36, mid, e36 / 6, e36 % 6);
      // (only = 2: the IMU sources one after the other by group 0 - the order of the light pairs' sums.  A pair is heavy on one rank and
      // light on another - the ranks hold different surfel shares -, and these sums do not pass through a collective that would make the
      // ranks agree: every rank must form them the same way, to the bit)
      double acc = a.only == 1 ? 0.0
                   : a.only == 2 ? (grp == 0 ? gather_pair_sum<1, 4>(a.src, a.partial, mid, eend, u, v) : 0.0)
                                 : gather_pair_sum<kGG, 4>(a.src, a.partial, mid + grp, eend, u, v);
      sred[tid] = acc6;
      sred2[grp * 144 + e] = acc;
      __syncthreads();
      if (grp == 0) {
        acc = 0.0;
        if (u < 6 && v < 6)
          for (int q = 0; q < kFarGroups; ++q) acc += sred[q * 36 + u * 6 + v];
        for (int q = 0; q < kGG; ++q) acc += sred2[q * 144 + e];
      }
#pragma unroll
      for (int k = 0; k < kLightSets; ++k) pidk[k] = pid, acck[k] = acc, writek[k] = k == 0 && grp == 0;
    } else {
      const uint32_t set0 = (blk - a.nheavy) * kLightSets;
      uint32_t bk[kLightSets], ek[kLightSets];
#pragma unroll
      for (int k = 0; k < kLightSets; ++k) {
        const uint32_t idx = (set0 + k) * kGG + grp;
        const bool valid = idx < a.nnear;
        pidk[k] = valid ? a.near[idx] : 0u;
        bk[k] = valid ? a.src_begin[pidk[k]] : 0u, ek[k] = valid ? a.src_begin[pidk[k] + 1] : 0u;
        writek[k] = valid && ek[k] - bk[k] <= kHeavySrc;  // (a pair with more sources is summed by its heavy workgroup)
      }
      const uint2 *src2 = (const uint2 *)a.src;
      uint2 d[kLightSets][4];
#pragma unroll
      for (int k = 0; k < kLightSets; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) d[k][q] = src2[(writek[k] && bk[k] + q < ek[k]) ? bk[k] + q : 0u];
      double val[kLightSets][4];
      bool ok[kLightSets][4];
#pragma unroll
      for (int k = 0; k < kLightSets; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t part_off = d[k][q].x, sp = d[k][q].y & 0xFFu, sq_ = (d[k][q].y >> 8) & 0xFFu, sw = (d[k][q].y >> 16) & 0xFFu, sT = d[k][q].y >> 24;
          ok[k][q] = writek[k] && bk[k] + q < ek[k] && (uint32_t)u < sw && (uint32_t)v < sw && (want_w == 0u || sw == want_w);
          uint32_t r = sp * sw + u, c = sq_ * sw + v;
          if (r > c) {
            const uint32_t t = r;
            r = c, c = t;
          }
          val[k][q] = a.partial[ok[k][q] ? part_off + tri_index(r, c, sT) : 0u];
        }
#pragma unroll
      for (int k = 0; k < kLightSets; ++k) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (ok[k][q]) acc += val[k][q];
        if (writek[k] && ek[k] - bk[k] > 4u) acc = gather_pair_sum<1, 8>(a.src, a.partial, bk[k] + 4u, ek[k], u, v, acc, want_w);  // (same order: onto acc)
        acck[k] = acc;
      }
    }
    const int ns = a.ns;
    const int n = 12 * ns;
    const int slot = heavy ? 0 : grp;
#pragma unroll
    for (int k = 0; k < kLightSets; ++k) {
      if (heavy && k > 0) break;
      const uint32_t pid = pidk[k];
      const bool write = writek[k];
      double acc = acck[k];
      // invert pid = I*ns - I(I-1)/2 + (J-I)
      int I = 0, J = 0;
      if (write) {
        const float f = 2.f * ns + 1.f;
        I = (int)((f - sqrtf(fmaxf(f * f - 8.f * (float)pid, 0.f))) * 0.5f);
        I = max(0, min(I, ns - 1));
        while (I > 0 && (uint32_t)(I * ns - I * (I - 1) / 2) > pid) --I;
        while ((uint32_t)((I + 1) * ns - (I + 1) * I / 2) <= pid) ++I;
        J = I + (int)(pid - (uint32_t)(I * ns - I * (I - 1) / 2));
      }
      const int gi = I * 12 + u, gj = J * 12 + v;
      if (a.fix_first && ((gi >= 3 && gi < 6) || (gj >= 3 && gj < 6))) acc = 0.0;  // SubsetParameterization(12,{3,4,5})
      if (a.packed) {
        if (write) {
          if (a.only) {
            if (u < 6 && v < 6) a.H[(size_t)36 * pid + u * 6 + v] = acc;  // (surfel factors touch pose corners only)
          } else if (J - I <= 2)
            a.H[(size_t)a.pair_off[pid] + e] = acc;
          else if (u < 6 && v < 6)
            a.H[(size_t)a.pair_off[pid] + u * 6 + v] = acc;  // (the rest of a far pair's block is zero: no IMU factor reaches it)
        }
        continue;
      }
      // block (I, J) row by row, and its transpose as block (J, I) ALSO row by row: the transposed entries come through
      // LDS (written straight from the registers the mirror block is 144 scattered 8-byte stores per pair, 9 MB of them)
      __syncthreads();  // (heavy: every group has read the partial sums; light: the previous set's mirror has been read)
      if (write) sred[slot * 144 + e] = acc;
      __syncthreads();
      if (write) {
        a.H[(size_t)gi * n + gj] = acc;
        a.H[(size_t)(J * 12 + u) * n + I * 12 + v] = sred[slot * 144 + v * 12 + u];
      }
    }
    return;
  }
  blk -= a.nheavy + nlight;
#ifdef WC_GATHER_KNOCK
  if ((WC_GATHER_KNOCK & 4) && blk < (uint32_t)a.ns) {
    if (a.post) gather_post(a, 0.0);
    return;
  }
  if ((WC_GATHER_KNOCK & 8) && blk >= (uint32_t)a.ns) {
    if (a.post) gather_post(a, 0.0);
    return;
  }
#endif
  if (blk < (uint32_t)a.ns) {  // g = J^T r of sample block blk
    // all 84 groups of 12 lanes stride over the block's source list (~320 sources: one trip of four each)
    constexpr int NGg = 144 * kGG / 12;
    const int I = (int)blk, grp = tid / 12, u = tid % 12;
    double acc = 0.0;
    if (a.only == 2) {
      // the IMU pieces' sources are the tail of the block's list (piece order); where it begins depends on this rank's surfel share, the
      // sum must not: group j fetches IMU source j, lane u adds them in list order (rank-independent, to the bit)
      __shared__ uint32_t s_mid;
      const uint32_t b0 = a.gsrc_begin[I], eend = a.gsrc_begin[I + 1];
      if (tid == 0) s_mid = eend;
      __syncthreads();
      for (uint32_t s = b0 + tid; s < eend; s += 144 * kGG)
        if (a.gsrc[s].w == 12) atomicMin(&s_mid, s);
      __syncthreads();
      const uint32_t mid = s_mid;
      double tot = 0.0;
      for (uint32_t c0 = mid; c0 < eend; c0 += NGg) {
        const uint32_t sq = c0 + grp;
        double val = 0.0;
        if (sq < eend) {
          const GSrc sr = a.gsrc[sq];
          if (u < sr.w) val = a.partial[sr.part_off + tri_index(sr.p * sr.w + u, sr.T - 1, sr.T)];
        }
        __syncthreads();
        sred[tid] = val;
        __syncthreads();
        if (tid < 12) {
          const uint32_t m = min((uint32_t)NGg, eend - c0);
          for (uint32_t q = 0; q < m; ++q) tot += sred[q * 12 + u];
        }
      }
      acc = tot;
      if (tid < 12) {
        const int gi = I * 12 + u;
        if (a.fix_first && gi >= 3 && gi < 6) acc = 0.0;
        a.g[gi] = acc;
      }
      return;
    }
    {
      const uint32_t eend = a.gsrc_begin[I + 1];
      for (uint32_t s = a.gsrc_begin[I] + grp; s < eend; s += 4 * NGg) {
        double val[4];
        bool ok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t sq = s + q * NGg;
          const bool live = sq < eend;
          const GSrc sr = a.gsrc[live ? sq : s];
          ok[q] = live && u < sr.w && (a.only == 0 || (a.only == 1) == (sr.w == 6));
          val[q] = a.partial[sr.part_off + (ok[q] ? tri_index(sr.p * sr.w + u, sr.T - 1, sr.T) : 0u)];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (ok[q]) acc += val[q];
      }
      sred[tid] = acc;
    }
    __syncthreads();
    if (tid < 12) {
      acc = 0.0;
      for (int q = 0; q < NGg; ++q) acc += sred[q * 12 + u];
      const int gi = I * 12 + u;
      if (a.fix_first && gi >= 3 && gi < 6) acc = 0.0;
      if (a.only == 1) {
        if (u < 6) a.g[I * 6 + u] = acc;  // (the pose half, packed)
      } else {
        a.g[gi] = acc;
      }
    }
    if (a.post) {
      double mx = tid < 12 ? fabs(acc) : 0.0;  // (lanes 0 .. 11 of wavefront 0 hold the block's entries)
      if (!(mx == mx)) mx = 0.0;
      if (tid < 64) {
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) mx = fmax(mx, __shfl_xor(mx, d, 64));
      }
      gather_post(a, mx);
    }
    return;
  }
  {  // cost: deterministic sum of the pieces' costs (one contiguous array behind the partials since round 5: as the corner of
     // every piece's partial they were 12 k scattered lines behind 12 k descriptor loads through ONE CU - 14 us at C4, the longest
     // chain of the kernel; same sums in the same order)
    constexpr int NT = 144 * kGG;
    double acc = 0.0;
    const uint32_t nsurf = a.nb_pieces + a.nu_pieces;
    const uint32_t pbeg = a.only == 2 ? nsurf : 0u, pend = a.only == 1 ? nsurf : a.npieces;  // (surfel pieces first, then the IMU pieces)
    const double *pcost = a.partial + a.cost_base + pbeg;
    const uint32_t npc = pend - pbeg;
    for (uint32_t p0 = tid; p0 < npc; p0 += 4 * NT) {
      double val[4];
      bool ok[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t p = p0 + q * NT;
        ok[q] = p < npc;
        val[q] = pcost[ok[q] ? p : p0];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (ok[q]) acc += val[q];
    }
    sred[tid] = acc;
    __syncthreads();
    if (tid < 16) {  // 1008 = 16 x 63
      double t = 0.0;
      for (int q = 0; q < 63; ++q) t += sred[tid * 63 + q];
      sred[tid * 63] = t;
    }
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int q = 0; q < 16; ++q) t += sred[q * 63];
      __hip_atomic_store(&a.cost[0], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (gather_post: read back by the workgroup that counts last)
    }
    if (a.post) gather_post(a, 0.0);
  }
}

// multi-GPU: the reduced block pairs (pair order, 144 doubles each) -> both triangles of the dense matrix; g and the cost
// follow the pairs in the reduction buffer and are copied behind H
__global__ void __launch_bounds__(144) k_expand_pairs(const double *packed, uint32_t npairs, int ns, int np, double *H, double *g,
                                                     const uint32_t *pair_off, uint32_t red_H) {
  const uint32_t pid = blockIdx.x;
  const int e = threadIdx.x;
  if (pid == npairs) {  // tail: g (np doubles) + cost, spare
    for (int i = e; i < np + 2; i += 144) g[i] = packed[(size_t)red_H + i];
    return;
  }
  const float f = 2.f * ns + 1.f;
  int I = (int)((f - sqrtf(fmaxf(f * f - 8.f * (float)pid, 0.f))) * 0.5f);
  I = max(0, min(I, ns - 1));
  while (I > 0 && (uint32_t)(I * ns - I * (I - 1) / 2) > pid) --I;
  while ((uint32_t)((I + 1) * ns - (I + 1) * I / 2) <= pid) ++I;
  const int J = I + (int)(pid - (uint32_t)(I * ns - I * (I - 1) / 2));
  const int n = 12 * ns, u = e / 12, w = e % 12, gi = I * 12 + u, gj = J * 12 + w;
  double v = 0.0;
  if (J - I <= 2)
    v = packed[(size_t)pair_off[pid] + e];
  else if (u < 6 && w < 6)
    v = packed[(size_t)pair_off[pid] + u * 6 + w];
  H[(size_t)gi * n + gj] = v;
  H[(size_t)gj * n + gi] = v;
}

// The two-collective form: what the ranks summed - the 6 x 6 pose corner of every pair (36 doubles at 36 pid), the pose
// half of g - joins what every rank holds complete: near pairs' corners ADD to the IMU factors' sums k_gather (only = 2) has just
// written, far pairs' corners are the sum itself.  Workgroup npairs: g's pose half, then max |g| over all unknowns -> mail[gslot] and -
// late_host != nullptr - the pinned mailbox's late slot (the LM loop reads it with the NEXT iteration's ticket: the trust-region decision
// needs the cost alone, which k_post_cost has sent ahead).
__global__ void __launch_bounds__(64) k_expand_corners(const double *red, uint32_t npairs, int ns, double *H, double *g, double *mail, int gslot,
                                                      double *late_host) {
  const uint32_t pid = blockIdx.x;
  const int e = threadIdx.x;
  const int n = 12 * ns;
  if (pid == npairs) {
    __shared__ double smx[64];
    const double *rg = red + (size_t)36 * npairs;
    for (int i = e; i < 6 * ns; i += 64) g[12 * (i / 6) + i % 6] += rg[i];
    __syncthreads();
    double mx = 0.0;
    for (int i = e; i < n; i += 64) mx = fmax(mx, fabs(g[i]));
    smx[e] = mx;
    __syncthreads();
    for (int st = 32; st > 0; st >>= 1) {
      if (e < st) smx[e] = fmax(smx[e], smx[e + st]);
      __syncthreads();
    }
    if (e == 0) {
      mail[gslot] = smx[0];
      if (late_host) *late_host = smx[0];
    }
    return;
  }
  if (e >= 36) return;
  const float f = 2.f * ns + 1.f;
  int I = (int)((f - sqrtf(fmaxf(f * f - 8.f * (float)pid, 0.f))) * 0.5f);
  I = max(0, min(I, ns - 1));
  while (I > 0 && (uint32_t)(I * ns - I * (I - 1) / 2) > pid) --I;
  while ((uint32_t)((I + 1) * ns - (I + 1) * I / 2) <= pid) ++I;
  const int J = I + (int)(pid - (uint32_t)(I * ns - I * (I - 1) / 2));
  const int u = e / 6, w = e % 6, gi = I * 12 + u, gj = J * 12 + w;
  double v = red[(size_t)36 * pid + e];
  if (J - I <= 2) v += H[(size_t)gi * n + gj];  // (the IMU factors' part, written by this linearisation's second gather)
  H[(size_t)gi * n + gj] = v;
  if (I != J) H[(size_t)gj * n + gi] = v;  // (a diagonal block's thread (u, w) owns entry (u, w) alone: (w, u) is thread (w, u)'s)
}

// the linearisation's cost = the IMU factors' (every rank's own sum: mail[54]) + the surfel factors' (summed over the ranks: mail[52]),
// sent to the host as soon as the 16-byte collective is through
__global__ void __launch_bounds__(64) k_post_cost(double *mail, int slot, double *cost_out, double *host_mail, unsigned long long ticket) {
  const int tid = threadIdx.x;
  const double c = mail[54] + mail[52];
  if (tid == 0) mail[slot] = c, cost_out[0] = c;
  if (host_mail) {
    if (tid < 40) host_mail[tid] = tid == slot ? c : mail[tid];
    if (ticket) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (tid == 0) {
        __threadfence_system();
        __hip_atomic_store((unsigned long long *)(host_mail + 48), ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// ---- cost-only evaluation (candidate step; problem.Evaluate) ----------------------------------------------------------
// binary factors (workgroups 0 .. gb-1) and unary factors (the rest) in ONE launch: two launches of these short kernels
// cost a kernel boundary and the tail of the first
struct EvalArgs {
  const double *rec;
  const uint32_t *keys, *orig;
  uint32_t n;
  double *residuals;
};
__global__ void __launch_bounds__(256) k_eval_surfel(WinParams wp, EvalArgs B, EvalArgs U, uint32_t gb, const double *x, double *block_cost) {
  __shared__ double s[256];
  const bool unary = blockIdx.x >= gb;
  const EvalArgs &E = unary ? U : B;
  const uint32_t k = (blockIdx.x - (unary ? gb : 0u)) * blockDim.x + threadIdx.x;
  double c = 0.0;
  if (k < E.n) {
    double r;
    if (unary)
      eval_unary(wp, E.rec, E.n, k, E.keys[k], x, r, c, nullptr);
    else
      eval_binary(wp, E.rec, E.n, k, E.keys[k], x, r, c, nullptr);
    if (E.residuals) E.residuals[E.orig[k]] = r;
  }
  s[threadIdx.x] = c;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) block_cost[blockIdx.x] = s[0];
}

__global__ void __launch_bounds__(256) k_eval_imu(WinParams wp, const ImuRec *recs, uint32_t n, const double *x, const double *times,
                                                 double *residuals, double *block_cost) {
  __shared__ double s[256];
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  double c = 0.0;
  if (k < n) {
    double res[12];
    eval_imu<-1>(wp, recs[k], x, times, res, nullptr, 0);
    for (int r = 0; r < 12; ++r) {
      c += res[r] * res[r];
      if (residuals) residuals[(size_t)k * 12 + r] = res[r];
    }
    c *= 0.5;
  }
  s[threadIdx.x] = c;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) block_cost[blockIdx.x] = s[0];
}

// (host_mail != null: the whole 40-double mailbox is stored to pinned host memory right here - the copy node that would
// follow is a 4 us blit kernel on the critical path of every LM iteration)
__global__ void __launch_bounds__(1024) k_sum_blocks(const double *v, uint32_t n, double *mail, int slot, double *host_mail) {
  __shared__ double s[1024];
  const int tid = threadIdx.x;
  double acc = 0.0;
  for (uint32_t i = tid; i < n; i += 1024) acc += v[i];
  s[tid] = acc;
  __syncthreads();
  for (int st = 512; st > 0; st >>= 1) {
    if (tid < st) s[tid] += s[tid + st];
    __syncthreads();
  }
  if (tid == 0) mail[slot] = s[0];
  if (host_mail) {
    __syncthreads();
    if (tid < 40) host_mail[tid] = (tid == slot) ? s[0] : mail[tid];
  }
}

// ---- LM linear algebra on the device -------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_scale_init(const double *H, int n, double *scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) scale[i] = 1.0 / (1.0 + sqrt(H[(size_t)i * n + i]));  // ceres jacobi_scaling: 1 / (1 + ||col||)
}

// A (lower, row-major, ld) = S H S + diag(clamp(diag(S H S), 1e-6, 1e32) / radius); row n = (S g)^T; padding = identity
__device__ __forceinline__ double damped_entry(const double *H, const double *g, const double *scale, int n, double radius, int i, int j,
                                               double *diag_out) {
  double v;
  if (i < n) {
    v = H[(size_t)i * n + j] * scale[i] * scale[j];
    if (i == j) {
      const double d = fmin(fmax(v, 1e-6), 1e32);  // LM min/max diagonal
      if (diag_out) diag_out[i] = d / radius;
      v += d / radius;
    }
  } else if (i == n) {
    // augmented row: z = L^-1 (S g) falls out of the factorisation; the corner only has to keep the pivot positive
    v = (j < n) ? g[j] * scale[j] : 1e300;
  } else {
    v = (i == j) ? 1.0 : 0.0;
  }
  return v;
}

// ---- blocked right-looking Cholesky, one launch per 32-column panel ----------------------------------------------------
// Step k consumes L_kk^-1 (left behind by step k-1) and, for every 64x64 tile (ti >= tj) of the trailing matrix:
//   L_ik = A_ik L_kk^-T for the tile's rows and columns (recomputed per tile: two 64x32x32 products, cheaper than a
//   separate TRSM launch and its dependency), A_ij -= L_ik L_jk^T, and (tiles of the first column) L_ik -> Lmat.
// Tile (0,0) then factors the next diagonal block in place (look-ahead), so the whole factorisation is nblk launches
// with no separate diagonal kernel on the critical path.  L is collected in Lmat (A keeps being updated in place).

// Cholesky factor L of a 32x32 block AND its inverse, by ONE wavefront with everything in registers: lanes 0..31 hold
// the rows of the block (-> rows of L), lanes 32..63 the rows of the identity (-> rows of L^-1, built by applying the
// same column eliminations).  Fully unrolled: a[] indices are compile-time constants, operands of other lanes come
// through v_readlane with uniform lane ids, no LDS and no barriers on the 32-step pivot chain.
// in: sB rows 0..31 (lower part).  out: sB = L (lower), sXi = L^-1.  Returns false on a non-positive pivot.
__device__ __forceinline__ double readlane_d(double v, int src_lane) {  // src_lane must be wave-uniform: v_readlane_b32 x2
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), src_lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
}

// 32x32 diagonal block: L (Cholesky) and L^-1, by the whole workgroup (256 threads), in LDS, in EIGHT block steps of four
// columns instead of 32 single-column steps: what bounds this routine is the number of dependent LDS round trips and
// barriers on the pivot chain, not arithmetic (a single-wavefront version with one step per column measured 13-28 us
// per block, register resident or not).  Per block step every thread factors and inverts the 4x4 pivot block
// redundantly in registers (no communication), 32 threads solve their row of the 4-column panel, 32 threads form the
// four new rows of L^-1 (forward substitution on 4-row blocks, one column each), then everybody applies the rank-4 update.
// in: sB rows 0..31 (lower part).  out: sB = L (lower, zeros above), sXi = L^-1.  Returns false on a non-positive pivot.


struct Piv4 {  // Cholesky factor of a 4x4 pivot block and its inverse (both lower triangular)
  double l00, l10, l11, l20, l21, l22, l30, l31, l32, l33;
  double i00, i10, i11, i20, i21, i22, i30, i31, i32, i33;
};

// Writes of block step pj that nobody reads while the next step's updates run (so they need no barrier of their own):
// the L rows of the 4-column panel (row r of A times Lp^-T), the pivot rows of L, zeros above the diagonal (lanes 0..31,
// one row each) and rows pj..pj+3 of X = Lp^-1 times the same rows of W (lanes 32..63, one column each).  Both are the
// same lower-triangular 4x4 product on four values that sit 1 resp. kNB+1 doubles apart; load, arithmetic and store are
// separate calls so that the loads go out with the step's other loads and the arithmetic fills the pivot chain's bubbles.
struct Fin4 {
  double *base;
  int stride;
  double v0, v1, v2, v3;
};
__device__ __forceinline__ Fin4 factor_finish_load(double (*sB)[kNB + 1], double (*sXi)[kNB + 1], int pj, int tid) {
  Fin4 f;
  const int l = tid & 63;
  f.base = l < kNB ? &sB[l][pj] : &sXi[pj][l - kNB];
  f.stride = l < kNB ? 1 : kNB + 1;
  f.v0 = f.base[0], f.v1 = f.base[f.stride], f.v2 = f.base[2 * f.stride], f.v3 = f.base[3 * f.stride];
  return f;
}
__device__ __forceinline__ void factor_finish_store(const Fin4 f, int pj, const Piv4 p, int tid) {
  const int l = tid & 63, a = l - pj;  // lanes < 32: row l of the panel; a = its position relative to the pivot rows
  double o0 = f.v0 * p.i00;
  double o1 = fma(f.v0, p.i10, f.v1 * p.i11);
  double o2 = fma(f.v0, p.i20, fma(f.v1, p.i21, f.v2 * p.i22));
  double o3 = fma(f.v0, p.i30, fma(f.v1, p.i31, fma(f.v2, p.i32, f.v3 * p.i33)));
  if (l < kNB && a < 4) {
    if (a < 0) {
      o0 = o1 = o2 = o3 = 0.0;  // above the diagonal
    } else {
      o0 = a == 0 ? p.l00 : (a == 1 ? p.l10 : (a == 2 ? p.l20 : p.l30));
      o1 = a == 0 ? 0.0 : (a == 1 ? p.l11 : (a == 2 ? p.l21 : p.l31));
      o2 = a <= 1 ? 0.0 : (a == 2 ? p.l22 : p.l32);
      o3 = a <= 2 ? 0.0 : p.l33;
    }
  }
  f.base[0] = o0, f.base[f.stride] = o1, f.base[2 * f.stride] = o2, f.base[3 * f.stride] = o3;
}

// element lk of the forward substitution of (v0..v3) against the pivot block: (Lp^-1 v)[lk], picked with the lane's 0/1
// weights m[k] = (lk == k).  Straight-line on purpose: with a select by lk the compiler builds a branch tree per call
// (lanes with lk = 0 "save" three elements), which cannot be interleaved with the pivot chain or with the second call.
__device__ __forceinline__ double piv_solve_elem(const Piv4 &q, double v0, double v1, double v2, double v3, const double (&m)[4]) {
  const double f0 = v0 * q.i00;
  const double f1 = fma(-f0, q.l10, v1) * q.i11;
  const double f2 = fma(-f1, q.l21, fma(-f0, q.l20, v2)) * q.i22;
  const double f3 = fma(-f2, q.l32, fma(-f1, q.l31, fma(-f0, q.l30, v3))) * q.i33;
  return fma(m[3], f3, fma(m[2], f2, fma(m[1], f1, m[0] * f0)));
}

// ONE barrier per block step, and the rank-4 updates on the fp64 matrix cores.  A block step's updates are
//   A[r][c] -= (a_r Lp^-T)(a_c Lp^-T)^T  (trailing block)   and   W[r][c] -= (a_r Lp^-T)(Lp^-1 Wp[:,c])  (inverse),
// over rows r below the pivot block: one 32 x 32 x 4 product = four 16x16x4 MFMA tiles, one per wavefront (tile (0,1) lies
// above the diagonal: idle).  A lane supplies ONE element of each operand, so it solves its two 4-vectors against the
// pivot block itself (the pivot chain runs redundantly in every thread's registers; nobody waits for a panel solve by
// other threads); the panel / pivot / X rows of a step are written during the next step (nobody reads them before the
// end).  What bounds a block step is instruction issue, not the dependent chain: with one thread per update entry (24
// fp64 operations each, 3.5 entries per thread) a step took 1.6 k clk whatever the length of the pivot chain; with the
// panel solve between two barriers and a load - compute - store loop over the entries, 2.7 k.
__device__ __forceinline__ bool factor_inv32_blk(double (*sB)[kNB + 1], double (*sXi)[kNB + 1]) {
  const int tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int rt = w >> 1, ct = w & 1;
  const int ra = 16 * rt + li, cb = 16 * ct + li;  // this lane's row of the A operand / column of the B operand
  const double mk[4] = {lk == 0 ? 1.0 : 0.0, lk == 1 ? 1.0 : 0.0, lk == 2 ? 1.0 : 0.0, lk == 3 ? 1.0 : 0.0};
  // sXi starts as the identity: rows below the current block step hold W = E - L X (right-looking substitution), rows
  // above it the finished rows of X = L^-1
  for (int e = tid; e < kNB * kNB; e += 256) sXi[e / kNB][e % kNB] = (e / kNB == e % kNB) ? 1.0 : 0.0;
  bool ok = true;
  Piv4 prev;
  __syncthreads();
  // The update tile of a wavefront stays in registers for the whole factorisation (MFMA C / D layout: register r4 of a
  // lane = element (16 rt + lk + 4 r4, cb)): LDS only carries what OTHER wavefronts read - the next step's panel columns
  // and W's next pivot rows.  A column holds the trailing block until its block step, then W (whose untouched part below
  // the diagonal is zero).
  f64x4 cc = {0.0, 0.0, 0.0, 0.0};
  if (w != 1) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) cc[r4] = sB[16 * rt + lk + 4 * r4][cb];
  }
  for (int j0 = 0; j0 < kNB; j0 += 4) {
    // the pivot block first: the chain below waits for nothing else
    const double p00 = sB[j0][j0], p10 = sB[j0 + 1][j0], p11 = sB[j0 + 1][j0 + 1], p20 = sB[j0 + 2][j0], p21 = sB[j0 + 2][j0 + 1],
                 p22 = sB[j0 + 2][j0 + 2], p30 = sB[j0 + 3][j0], p31 = sB[j0 + 3][j0 + 1], p32 = sB[j0 + 3][j0 + 2],
                 p33 = sB[j0 + 3][j0 + 3];
    // operands of this wavefront's update tile, requested before the pivot chain
    const bool tile = (w != 1) && (16 * rt + 15 >= j0 + 4);  // wave-uniform
    const bool bt = cb >= j0 + 4;                             // this lane's column: trailing block (else a column of W)
    if ((cb >> 2) == (j0 >> 2)) cc = f64x4{0.0, 0.0, 0.0, 0.0};  // the pivot columns turn into columns of W
    double va[4] = {0.0, 0.0, 0.0, 0.0}, vb[4] = {0.0, 0.0, 0.0, 0.0};
    if (tile) {
      const double *o = bt ? &sB[cb][j0] : &sXi[j0][cb];  // row cb of the panel / column cb of W's pivot rows
      const int os = bt ? 1 : kNB + 1;
#pragma unroll
      for (int t = 0; t < 4; ++t) va[t] = sB[ra][j0 + t], vb[t] = o[t * os];
    }
    Fin4 fin;
    if (w == 1) fin = factor_finish_load(sB, sXi, j0 > 0 ? j0 - 4 : 0, tid);  // wavefront 1 has no tile: it finishes the previous step
    // ---- 4x4 pivot block: factor + inverse, every thread on its own ----
    // (explicit fma: the solve is compared with the oracle at 1e-6, not bit for bit)
    Piv4 q;
    // pivots in pairs: 1 / l11 = sqrt(p00) rsqrt(p00 p11 - p10^2), so the two reciprocal square roots of a 2x2 block run
    // side by side (the determinant carries the same cancellation as p11 - l10^2); same for the 2x2 Schur complement
    q.i00 = rsqrt_nr(p00);
    const double det01 = fma(p00, p11, -(p10 * p10));
    const double rd01 = rsqrt_nr(det01);
    q.l00 = p00 * q.i00, q.l10 = p10 * q.i00, q.l20 = p20 * q.i00, q.l30 = p30 * q.i00;
    q.i11 = q.l00 * rd01;
    q.l11 = (det01 * rd01) * q.i00;
    q.l21 = fma(-q.l20, q.l10, p21) * q.i11, q.l31 = fma(-q.l30, q.l10, p31) * q.i11;
    const double s22 = fma(-q.l21, q.l21, fma(-q.l20, q.l20, p22));
    const double s32 = fma(-q.l31, q.l21, fma(-q.l30, q.l20, p32));
    const double s33 = fma(-q.l31, q.l31, fma(-q.l30, q.l30, p33));
    q.i22 = rsqrt_nr(s22);
    const double det23 = fma(s22, s33, -(s32 * s32));
    const double rd23 = rsqrt_nr(det23);
    q.l22 = s22 * q.i22, q.l32 = s32 * q.i22;
    q.i33 = q.l22 * rd23;
    q.l33 = (det23 * rd23) * q.i22;
    if (!(p00 > 0.0 && det01 > 0.0 && s22 > 0.0 && det23 > 0.0)) ok = false;
    if (w == 1) {  // the off-diagonal part of Lp^-1 is only needed to finish the step
      q.i10 = -(q.l10 * q.i00) * q.i11;
      q.i21 = -(q.l21 * q.i11) * q.i22;
      q.i20 = -fma(q.l20, q.i00, q.l21 * q.i10) * q.i22;
      q.i32 = -(q.l32 * q.i22) * q.i33;
      q.i31 = -fma(q.l31, q.i11, q.l32 * q.i21) * q.i33;
      q.i30 = -fma(q.l30, q.i00, fma(q.l31, q.i10, q.l32 * q.i20)) * q.i33;
    }
    // ---- rank-4 update of this wavefront's tile ----
    const double fa = piv_solve_elem(q, va[0], va[1], va[2], va[3], mk);
    const double fb = piv_solve_elem(q, vb[0], vb[1], vb[2], vb[3], mk);
    if (tile) {
      cc = __builtin_amdgcn_mfma_f64_16x16x4f64(ra >= j0 + 4 ? -fa : 0.0, fb, cc, 0, 0, 0);
      if (bt) {
        if (cb < j0 + 8) {  // the next step's panel (and pivot block): lower part, rows from the next pivot block on
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int row = 16 * rt + lk + 4 * r4;
            if (row >= j0 + 4 && cb <= row) sB[row][cb] = cc[r4];
          }
        }
      } else {  // W's next pivot rows j0+4 .. j0+7: row j0 + 4 + lk is register (j0 + 4 - 16 rt) / 4 of this lane
        const int rn = j0 + 4 - 16 * rt;
        if (rn >= 0 && rn < 16) sXi[j0 + 4 + lk][cb] = rn == 0 ? cc[0] : (rn == 4 ? cc[1] : (rn == 8 ? cc[2] : cc[3]));
      }
    }
    if (w == 1 && j0 > 0) factor_finish_store(fin, j0 - 4, prev, tid);
    prev = q;
    __syncthreads();
  }
  if (w == 1) factor_finish_store(factor_finish_load(sB, sXi, kNB - 4, tid), kNB - 4, prev, tid);
  return ok;
}

template <int WV>
__device__ __forceinline__ bool factor_inv32_role(double (*sB)[kNB + 1], double (*sXi)[kNB + 1]) {
  const int tid = threadIdx.x;
  constexpr int w = WV;
  const int lane = tid & 63, li = lane & 15, lk = lane >> 4;
  constexpr int rt = w >> 1, ct = w & 1;
  const int ra = 16 * rt + li, cb = 16 * ct + li;  // this lane's row of the A operand / column of the B operand
  const double mk[4] = {lk == 0 ? 1.0 : 0.0, lk == 1 ? 1.0 : 0.0, lk == 2 ? 1.0 : 0.0, lk == 3 ? 1.0 : 0.0};
  // sXi starts as the identity: rows below the current block step hold W = E - L X (right-looking substitution), rows
  // above it the finished rows of X = L^-1
  for (int e = tid; e < kNB * kNB; e += 256) sXi[e / kNB][e % kNB] = (e / kNB == e % kNB) ? 1.0 : 0.0;
  bool ok = true;
  Piv4 prev;
  __syncthreads();
  // The update tile of a wavefront stays in registers for the whole factorisation (MFMA C / D layout: register r4 of a
  // lane = element (16 rt + lk + 4 r4, cb)): LDS only carries what OTHER wavefronts read - the next step's panel columns
  // and W's next pivot rows.  A column holds the trailing block until its block step, then W (whose untouched part below
  // the diagonal is zero).
  f64x4 cc = {0.0, 0.0, 0.0, 0.0};
  if (w != 1) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) cc[r4] = sB[16 * rt + lk + 4 * r4][cb];
  }
  for (int j0 = 0; j0 < kNB; j0 += 4) {
    // the pivot block first: the chain below waits for nothing else
    const double p00 = sB[j0][j0], p10 = sB[j0 + 1][j0], p11 = sB[j0 + 1][j0 + 1], p20 = sB[j0 + 2][j0], p21 = sB[j0 + 2][j0 + 1],
                 p22 = sB[j0 + 2][j0 + 2], p30 = sB[j0 + 3][j0], p31 = sB[j0 + 3][j0 + 1], p32 = sB[j0 + 3][j0 + 2],
                 p33 = sB[j0 + 3][j0 + 3];
    // operands of this wavefront's update tile, requested before the pivot chain
    const bool tile = (w != 1) && (16 * rt + 15 >= j0 + 4);  // wave-uniform
    const bool bt = cb >= j0 + 4;                             // this lane's column: trailing block (else a column of W)
    if ((cb >> 2) == (j0 >> 2)) cc = f64x4{0.0, 0.0, 0.0, 0.0};  // the pivot columns turn into columns of W
    double va[4] = {0.0, 0.0, 0.0, 0.0}, vb[4] = {0.0, 0.0, 0.0, 0.0};
    if (tile) {
      const double *o = bt ? &sB[cb][j0] : &sXi[j0][cb];  // row cb of the panel / column cb of W's pivot rows
      const int os = bt ? 1 : kNB + 1;
#pragma unroll
      for (int t = 0; t < 4; ++t) va[t] = sB[ra][j0 + t], vb[t] = o[t * os];
    }
    Fin4 fin;
    if (w == 1) fin = factor_finish_load(sB, sXi, j0 > 0 ? j0 - 4 : 0, tid);  // wavefront 1 has no tile: it finishes the previous step
    // ---- 4x4 pivot block: factor + inverse, every thread on its own ----
    // (explicit fma: the solve is compared with the oracle at 1e-6, not bit for bit)
    Piv4 q;
    // pivots in pairs: 1 / l11 = sqrt(p00) rsqrt(p00 p11 - p10^2), so the two reciprocal square roots of a 2x2 block run
    // side by side (the determinant carries the same cancellation as p11 - l10^2); same for the 2x2 Schur complement
    q.i00 = rsqrt_nr(p00);
    const double det01 = fma(p00, p11, -(p10 * p10));
    const double rd01 = rsqrt_nr(det01);
    q.l00 = p00 * q.i00, q.l10 = p10 * q.i00, q.l20 = p20 * q.i00, q.l30 = p30 * q.i00;
    q.i11 = q.l00 * rd01;
    q.l11 = (det01 * rd01) * q.i00;
    q.l21 = fma(-q.l20, q.l10, p21) * q.i11, q.l31 = fma(-q.l30, q.l10, p31) * q.i11;
    const double s22 = fma(-q.l21, q.l21, fma(-q.l20, q.l20, p22));
    const double s32 = fma(-q.l31, q.l21, fma(-q.l30, q.l20, p32));
    const double s33 = fma(-q.l31, q.l31, fma(-q.l30, q.l30, p33));
    q.i22 = rsqrt_nr(s22);
    const double det23 = fma(s22, s33, -(s32 * s32));
    const double rd23 = rsqrt_nr(det23);
    q.l22 = s22 * q.i22, q.l32 = s32 * q.i22;
    q.i33 = q.l22 * rd23;
    q.l33 = (det23 * rd23) * q.i22;
    if (!(p00 > 0.0 && det01 > 0.0 && s22 > 0.0 && det23 > 0.0)) ok = false;
    if (w == 1) {  // the off-diagonal part of Lp^-1 is only needed to finish the step
      q.i10 = -(q.l10 * q.i00) * q.i11;
      q.i21 = -(q.l21 * q.i11) * q.i22;
      q.i20 = -fma(q.l20, q.i00, q.l21 * q.i10) * q.i22;
      q.i32 = -(q.l32 * q.i22) * q.i33;
      q.i31 = -fma(q.l31, q.i11, q.l32 * q.i21) * q.i33;
      q.i30 = -fma(q.l30, q.i00, fma(q.l31, q.i10, q.l32 * q.i20)) * q.i33;
    }
    // ---- rank-4 update of this wavefront's tile ----
    const double fa = piv_solve_elem(q, va[0], va[1], va[2], va[3], mk);
    const double fb = piv_solve_elem(q, vb[0], vb[1], vb[2], vb[3], mk);
    if (tile) {
      cc = __builtin_amdgcn_mfma_f64_16x16x4f64(ra >= j0 + 4 ? -fa : 0.0, fb, cc, 0, 0, 0);
      if (bt) {
        if (cb < j0 + 8) {  // the next step's panel (and pivot block): lower part, rows from the next pivot block on
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int row = 16 * rt + lk + 4 * r4;
            if (row >= j0 + 4 && cb <= row) sB[row][cb] = cc[r4];
          }
        }
      } else {  // W's next pivot rows j0+4 .. j0+7: row j0 + 4 + lk is register (j0 + 4 - 16 rt) / 4 of this lane
        const int rn = j0 + 4 - 16 * rt;
        if (rn >= 0 && rn < 16) sXi[j0 + 4 + lk][cb] = rn == 0 ? cc[0] : (rn == 4 ? cc[1] : (rn == 8 ? cc[2] : cc[3]));
      }
    }
    if (w == 1 && j0 > 0) factor_finish_store(fin, j0 - 4, prev, tid);
    prev = q;
    __syncthreads();
  }
  if (w == 1) factor_finish_store(factor_finish_load(sB, sXi, kNB - 4, tid), kNB - 4, prev, tid);
  return ok;
}

// Round 6: the same routine with the wavefront's role (its tile, or the finishing role of wavefront 1) as a compile-time constant: one
// branch on the wavefront at the top instead of ~25 wave-uniform predicates per block step, which the fully unrolled body kept in SGPRs
// spilled to VGPR lanes (two v_readlane + wait states per use).  11.7 k -> 9.6 k shader clocks, the same bits (wc_selftest_factor32, variant 1 against 0).
// (The block-step loop unrolled by two instead of fully - predicates formed in place, an eighth of the code: 11.1 k.)
__device__ __forceinline__ bool factor_inv32_roles(double (*sB)[kNB + 1], double (*sXi)[kNB + 1]) {
  switch (threadIdx.x >> 6) {
    case 0: return factor_inv32_role<0>(sB, sXi);
    case 1: return factor_inv32_role<1>(sB, sXi);
    case 2: return factor_inv32_role<2>(sB, sXi);
    default: return factor_inv32_role<3>(sB, sXi);
  }
}

// (Round 6 rebuilt this routine twice around measured costs - profiles/micro/dep64.hip: a dependent fp64 operation 9 clocks, v_rsq_f64 20, a
// matrix-core step 68 - 81, an LDS round trip inside a wavefront 108, barrier + LDS ~150 - with the pivot chain on ONE wavefront instead of
// redundantly in all four, and the two 4 x 4 forward substitutions of the update replaced by matrix-core products with Lp^-1 as a 16 x 4
// operand (the product's C / D layout IS the update's A / B operand layout, and each lane's value is the element of L resp. L^-1 it owns:
// no finishing pass).  Form 1: the pivot wavefront a block step ahead - it forms the next pivot block itself from block row s + 1 of the
// panel, one entry per lane, 16-lane exchange through LDS - one barrier per step: 11.8 k clocks, exact to 1e-15.  Form 2: two half steps -
// wavefront 0 forms the next pivot block and wavefront 2 turns Lp into the operand table while the pivot wavefront stores, then the pivot
// wavefront factors while the others update: 12.4 k.  This routine: 11.8 k.  Stamps inside form 2: the update's three matrix-core steps
// with their LDS reads and predicated stores last 850 clocks per block step, the next pivot block 700, the 4 x 4 factor 610 - every piece
// about twice its dependent-latency sum: a wavefront alone on its SIMD pays ~6 - 9 clocks for every instruction, LDS and predication
// included, and a block step is ~200 of them however they are dealt out.  Neither form kept.)
// (Round 5 tried this factor + inverse by ONE wavefront, a column of [A | I] per lane, in two forms - measured on their own with
// wc_selftest_factor32 / profiles/dev/factor32.py against the block form's 11.7 - 12.8 k shader clocks: the columns in LDS, the pivot
// column as LDS broadcasts, one fma per row below the pivot: 73 k clocks (every row is a load - fma - store round trip; the compiler
// cannot move a row's loads above the previous row's store); the columns in registers, all 32 steps unrolled, the pivot column by
// v_readlane: 23.5 k (1 500 readlane -> fma pairs, each through an SGPR with its wait states).  Both exact to 1e-15, neither kept: the
// matrix-core rank-4 updates of the block form are what make it fast.)
// Damping and the factor of the first diagonal block in ONE launch: grid row 0 holds one workgroup that forms the damped
// 32 x 32 corner itself (the same expression as the other rows' threads) and factors + inverts it - a launch of its own for
// that block was 9 - 11 us of every LM iteration behind a 5 - 7 us k_damp.  It also owns the failure flag of the factorisation.
__global__ void __launch_bounds__(256) k_damp_first(const double *H, const double *g, const double *scale, int n, int np, int ld,
                                                   double radius, double *A, double *diag, double *Lmat, double *Linv, int *fail) {
  if (blockIdx.y == 0) {  // (dispatched first: it is the longest chain of the launch)
    if (blockIdx.x != 0) return;
    __shared__ double sB[kNB][kNB + 1];
    __shared__ double sXi[kNB][kNB + 1];
    for (int e = threadIdx.x; e < kNB * kNB; e += 256) {
      const int r = e / kNB, c = e % kNB;
      sB[r][c] = c <= r ? damped_entry(H, g, scale, n, radius, r, c, nullptr) : 0.0;
    }
    __syncthreads();
    const bool ok = factor_inv32_roles(sB, sXi);
    if (threadIdx.x == 0) *fail = ok ? 0 : 1;
    __syncthreads();
    for (int e = threadIdx.x; e < kNB * kNB; e += 256) {
      const int r = e / kNB, c = e % kNB;
      Lmat[(size_t)r * ld + c] = sB[r][c];
      Linv[e] = sXi[r][c];
    }
    return;
  }
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = (int)blockIdx.y - 1;
  if (j >= np || j > i) return;
  A[(size_t)i * ld + j] = damped_entry(H, g, scale, n, radius, i, j, diag);
}

#ifdef WC_PROF_CHOL  // -DWC_PROF_CHOL: phase timers of the lead tile (the critical path of the factorisation), printed at step 20
#define WC_CT(i) ct_[i] = clock64()
#else
#define WC_CT(i)
#endif
// Lead workgroup of a step (see k_chol_step): L rows of the next diagonal block, its trailing update, factor + inverse.
// sP: panel rows of the block, later L^-1 of the block; sXk: L_kk^-1; sD: old diagonal block -> updated -> L; sL: the
// block's rows of L (panel solve).  The four arrays alias the tile path's LDS (two workgroups per CU stay resident).
__device__ __forceinline__ void chol_lead(const double *A, int ld, int k, int nblk, double *Lmat, double *Linv, int *fail,
                                          double (*sP)[kNB + 1], double (*sXk)[kNB + 1], double (*sD)[kNB + 1],
                                          double (*sL)[kNB + 1]) {
#ifdef WC_PROF_CHOL
  long long ct_[8];
#endif
  WC_CT(0);
  const int tid = threadIdx.x;
  const int first = (k + 1) * kNB;
  const size_t pc = (size_t)k * kNB;
  const int failed = *fail;
#pragma unroll
  for (int e0 = 0; e0 < kNB * kNB; e0 += 256) {
    const int e = e0 + tid, r = e / kNB, c = e % kNB;
    sXk[r][c] = Linv[(size_t)k * kNB * kNB + e];
    sP[r][c] = A[(size_t)(first + r) * ld + pc + c];
    sD[r][c] = A[(size_t)(first + r) * ld + first + (c <= r ? c : r)];
  }
  __syncthreads();
  WC_CT(1);
  if (failed) return;
  const int w = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int rt = w >> 1, ct = w & 1;  // wavefront = one 16 x 16 tile of the 32 x 32 block
  {
    f64x4 t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < kNB / 4; ++ks)
      t = __builtin_amdgcn_mfma_f64_16x16x4f64(sP[16 * rt + li][4 * ks + lk], sXk[16 * ct + li][4 * ks + lk], t, 0, 0, 0);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) sL[16 * rt + lk + 4 * r4][16 * ct + li] = t[r4];
  }
  __syncthreads();
  WC_CT(2);
  {
    f64x4 t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < kNB / 4; ++ks)
      t = __builtin_amdgcn_mfma_f64_16x16x4f64(sL[16 * rt + li][4 * ks + lk], sL[16 * ct + li][4 * ks + lk], t, 0, 0, 0);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) sD[16 * rt + lk + 4 * r4][16 * ct + li] -= t[r4];
  }
  __syncthreads();
  WC_CT(3);
  const bool ok = factor_inv32_roles(sD, sP);  // in: lower part of sD; out: sD = L, sP = L^-1
  __syncthreads();
  WC_CT(4);
#ifdef WC_PROF_CHOL
  if (tid == 0 && k == 20)
    printf("chol step 20, lead workgroup, shader clocks: loads %lld trsm %lld update %lld factor %lld\n", ct_[1] - ct_[0], ct_[2] - ct_[1],
           ct_[3] - ct_[2], ct_[4] - ct_[3]);
#endif
  if (tid == 0 && !ok) atomicOr(fail, 1);
  for (int e = tid; e < kNB * kNB; e += 256) {
    const int r = e / kNB, c = e % kNB;
    Lmat[(size_t)(first + r) * ld + first + c] = sD[r][c];
    Linv[(size_t)(k + 1) * kNB * kNB + e] = sP[r][c];
  }
}

// Round 6: the back substitution L^T y = z as ONE matrix-vector product.  Rows appended below a matrix that is being factored leave
// the factorisation as R L^-T (that is how row n = c^T becomes z^T); appended IDENTITY rows leave it as L^-T.  Block row b of that
// identity is all zero until panel step b, where its panel block is I (so its L block is Linv_b^T and its trailing blocks
// -Linv_b^T L_jb^T); from then on it is updated like any other row.  The identity is never stored: a tile of the appended rows (rows
// nrow .. of A / Lmat, "E") synthesises it, treats rows of blocks > k as absent and the old values of block k as zero.  The tiles of E
// are up to (k + 1) / 2 x tiles more workgroups per step, on compute units the step leaves idle (the step lasts as long as its lead
// workgroup's chain); k_back_mul then forms y = L^-T z - one launch of ~5 us for the three chunk solves + two products (52 us at 127
// sample states, 25 at 64) of rounds 2 - 5, whose block rows were a chain of 1.8 us steps through one compute unit.
__device__ __forceinline__ void chol_extra_tile(double *A, int ld, int k, int nblk, double *Lmat, const double *Linv, const int *fail, int te, int tj,
                                                double (*sA)[kNB + 1], double (*sLi)[kNB + 1], double (*sLj)[kNB + 1], double (*sX)[kNB + 1], int dbg) {
  if (dbg & 1) return;
  const int tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int nrow = nblk * kNB, first = (k + 1) * kNB;
  const int re0 = 64 * te, col0 = first + tj * 64;
  const int act = (k + 1) * kNB, oldlim = k * kNB;  // E rows below `act` exist at this step; those below `oldlim` hold values
  const size_t pc = (size_t)k * kNB;
  const int failed = *fail;
  double old[4][4];
#pragma unroll
  for (int tq = 0; tq < 4; ++tq)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int re = re0 + 16 * w + lk + 4 * r4, c = col0 + 16 * tq + li;
      const bool in = re < oldlim && c < nrow && !(dbg & 2);
      const double v = A[in ? (size_t)(nrow + re) * ld + c : (size_t)first * ld + first];
      old[tq][r4] = in ? v : 0.0;
    }
  double xv_[4], av_[8], jv_[8];  // (all global loads before the first LDS store, as in the tiles of the factorisation)
#pragma unroll
  for (int q = 0; q < 4; ++q) xv_[q] = Linv[(size_t)k * kNB * kNB + tid + 256 * q];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int e = 256 * q + tid, r = e / kNB, c = e % kNB, re = re0 + r;
    av_[q] = A[re < oldlim ? (size_t)(nrow + re) * ld + pc + c : (size_t)first * ld + first];
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int e = 256 * q + tid, r = e / kNB, c = e % kNB;
    jv_[q] = A[(size_t)(col0 + r < nrow ? col0 + r : first) * ld + pc + c];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) sX[(tid + 256 * q) / kNB][(tid + 256 * q) % kNB] = xv_[q];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int e = 256 * q + tid, r = e / kNB, c = e % kNB, re = re0 + r;
    sA[r][c] = re < oldlim ? av_[q] : ((re < act && re - oldlim == c) ? 1.0 : 0.0);
    sLj[r][c] = col0 + r < nrow ? jv_[q] : 0.0;
  }
  __syncthreads();
  if (failed) return;
  {
    f64x4 t0 = {0.0, 0.0, 0.0, 0.0}, t1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < kNB / 4; ++ks) {
      const double a = sA[16 * w + li][4 * ks + lk];
      t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sX[li][4 * ks + lk], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sX[16 + li][4 * ks + lk], t1, 0, 0, 0);
    }
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int r = 16 * w + lk + 4 * r4;
      sLi[r][li] = t0[r4];
      sLi[r][16 + li] = t1[r4];
      if (tj == 0 && re0 + r < act) {
        Lmat[(size_t)(nrow + re0 + r) * ld + pc + li] = t0[r4];
        Lmat[(size_t)(nrow + re0 + r) * ld + pc + 16 + li] = t1[r4];
      }
    }
  }
  {  // the tile's columns: their panel rows, solved in place (a wavefront reads and writes only its own 16 rows, and writes after its last read)
    f64x4 t0 = {0.0, 0.0, 0.0, 0.0}, t1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < kNB / 4; ++ks) {
      const double a = sLj[16 * w + li][4 * ks + lk];
      t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sX[li][4 * ks + lk], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sX[16 + li][4 * ks + lk], t1, 0, 0, 0);
    }
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      sLj[16 * w + lk + 4 * r4][li] = t0[r4];
      sLj[16 * w + lk + 4 * r4][16 + li] = t1[r4];
    }
  }
  __syncthreads();
  f64x4 acc[4];
#pragma unroll
  for (int tq = 0; tq < 4; ++tq) acc[tq] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < kNB / 4; ++ks) {
    const double a = sLi[16 * w + li][4 * ks + lk];
#pragma unroll
    for (int tq = 0; tq < 4; ++tq) acc[tq] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sLj[16 * tq + li][4 * ks + lk], acc[tq], 0, 0, 0);
  }
#pragma unroll
  for (int tq = 0; tq < 4; ++tq)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int re = re0 + 16 * w + lk + 4 * r4, c = col0 + 16 * tq + li;
      if (re < act && c < nrow && !(dbg & 4)) A[(size_t)(nrow + re) * ld + c] = old[tq][r4] - acc[tq][r4];
    }
}

// n_tri: workgroups of the factorisation proper (1 + tiles (tiles + 1) / 2); the grid's remaining workgroups are tiles of the appended
// identity rows (chol_extra_tile), `tiles` per tile row
__global__ void __launch_bounds__(256) k_chol_step(double *A, int ld, int k, int nblk, double *Lmat, double *Linv, int *fail, int n_real, int n_tri,
                                                   int tiles, int dbg_in = 0, long long *dbgbuf = nullptr) {
#ifdef WC_DEV_KNOBS  // (development option dbg_lm: per-workgroup wall-clock stamps (8), phase clocks of one tile (16), knock-outs of the appended tiles (1, 2, 4))
  const int dbg = dbg_in;
#else
  constexpr int dbg = 0;  // (the release kernel carries none of it)
#endif
  struct Stamp {
    long long *p;
    __device__ Stamp(long long *q) : p(q) {
      if (p && threadIdx.x == 0) p[0] = wall_clock64();
    }
    __device__ ~Stamp() {
      if (p && threadIdx.x == 0) p[1] = wall_clock64();
    }
  } stamp_((dbg & 8) ? dbgbuf + ((size_t)k * 512 + blockIdx.x) * 2 : nullptr);
  __shared__ double sA[64][kNB + 1];
  __shared__ double sLi[64][kNB + 1];
  __shared__ double sLj[64][kNB + 1];
  __shared__ double sX[kNB][kNB + 1];
  if ((int)blockIdx.x >= n_tri) {
    const int x = (int)blockIdx.x - n_tri;
    chol_extra_tile(A, ld, k, nblk, Lmat, Linv, fail, x / tiles, x % tiles, sA, sLi, sLj, sX, dbg);
    return;
  }
  // row 0 of the grid holds the lead workgroup (x == 0): the critical path of the factorisation.  It redoes the 32 x 32
  // corner of tile (0, 0) - panel solve of the next diagonal block's rows, its update - and factors + inverts that block
  // right away; a quarter of a tile's matrix-core work (fp64 MFMA runs at the vector rate: 64 clk per 16x16x4) and no
  // write-back stand between the launch and the factor.
  // Linear grid: workgroup 0 is the lead, workgroup 1 + ti (ti + 1) / 2 + tj the tile (ti, tj) of the lower triangle.  (A
  // tiles x (tiles + 1) grid whose upper half returned at once had up to 600 workgroups for 277 real ones: with more than
  // 256 of them the dispatcher put a tile onto the lead's CU and the lead took 9 us instead of 7 - the first 17 steps.)
  if (blockIdx.x == 0) {
    // (a next diagonal block that holds only the augmented row and padding - the unknowns fill whole blocks, e.g. 64 sample states -
    // is never used: the back substitution stops in front of it, z's last columns come from this launch's tile.  No lead, no
    // 5 us factor in the last step's chain.)
    if ((k + 1) * kNB >= n_real) return;
    chol_lead(A, ld, k, nblk, Lmat, Linv, fail, sA, sX, sLi, sLj);
    return;
  }
  const long long t_start_ = clock64();
  const int b_ = (int)blockIdx.x - 1;
  int ti = (int)((sqrtf(8.0f * (float)b_ + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) / 2 <= b_) ++ti;
  while (ti * (ti + 1) / 2 > b_) --ti;
  const int tj = b_ - ti * (ti + 1) / 2;
  const int failed = *fail;  // tested after the first barrier: one round trip together with the tile's loads, not before them
  const int tid = threadIdx.x;
  const int nrow = nblk * kNB;
  const int first = (k + 1) * kNB;
  const int row0 = first + ti * 64, col0 = first + tj * 64;
  const size_t pc = (size_t)k * kNB;  // first column of the panel
  const int w = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  // the tile's old values, requested with everything else the tile reads: unconditional loads from clamped addresses (a
  // guarded load is a branch per element), all of them before the first store (a store to the same array orders every
  // later load behind it: 16 dependent round trips, 17 us of a 42 us step, measured)
  double old[4][4];
#pragma unroll
  for (int tq = 0; tq < 4; ++tq)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int r = row0 + 16 * w + lk + 4 * r4, c = col0 + 16 * tq + li;
      const bool in = r < nrow && c <= r;
      old[tq][r4] = A[in ? (size_t)r * ld + c : (size_t)first * ld + first];
    }
  // L_kk^-1 was left behind by the previous launch (tile (0,0) factors and inverts the next diagonal block in registers)
  // Round 6: every global load of the tile - 16 old values, 4 of L_kk^-1, 8 + 8 panel values - is issued into registers BEFORE the first
  // LDS store.  Written as load - store loops, the compiler drained the queue (s_waitcnt vmcnt(0)) between the groups: three to four
  // round trips to memory another compute unit wrote in the previous launch, 7.3 k of a tile's 14.6 k shader clocks (stamps of round 6).
  double xv_[4], av_[8], jv_[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) xv_[q] = Linv[(size_t)k * kNB * kNB + tid + 256 * q];
#pragma unroll
  for (int q = 0; q < 8; ++q) {  // L rows of this tile's row range: Li = A[row0.., panel] * Linv^T
    const int e = 256 * q + tid, r = e / kNB, c = e % kNB;
    av_[q] = A[(size_t)(row0 + r < nrow ? row0 + r : first) * ld + pc + c];
  }
  if (ti != tj) {  // panel rows of the tile's columns, in the same round trip
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = 256 * q + tid, r = e / kNB, c = e % kNB;
      jv_[q] = A[(size_t)(col0 + r < nrow ? col0 + r : first) * ld + pc + c];
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) sX[(tid + 256 * q) / kNB][(tid + 256 * q) % kNB] = xv_[q];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int e = 256 * q + tid, r = e / kNB, c = e % kNB;
    sA[r][c] = row0 + r < nrow ? av_[q] : 0.0;
  }
  if (ti != tj) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = 256 * q + tid, r = e / kNB, c = e % kNB;
      sLj[r][c] = col0 + r < nrow ? jv_[q] : 0.0;
    }
  }
  __syncthreads();
  long long pc_[6];
  const bool prof_ = (dbg & 16) && b_ == 1 && k == 5;
  if (prof_) pc_[0] = clock64();
  if (failed) return;
  // The two small GEMMs of a tile run on the fp64 matrix cores: v_mfma_f64_16x16x4 takes A[i = l & 15][k = l >> 4] and
  // B[k = l >> 4][j = l & 15] as ONE double per lane, i.e. one LDS read per lane feeds 16 x 16 x 4 products; the scalar
  // loops (9 LDS reads per 8 products, 8 per 16) were LDS-bandwidth bound at 3.4 us each.  fp64 MFMA runs at the fp64 vector
  // rate - the gain is operand reuse.  Wavefront w owns rows 16 w .. 16 w + 15 of the tile.
  {  // Li = A[rows, panel] * Linv^T  (64 x 32, K = 32)
    f64x4 t0 = {0.0, 0.0, 0.0, 0.0}, t1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < kNB / 4; ++ks) {
      const double a = sA[16 * w + li][4 * ks + lk];
      t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sX[li][4 * ks + lk], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sX[16 + li][4 * ks + lk], t1, 0, 0, 0);
    }
    // D layout of the fp64 form: register r of lane l is element (row = (l >> 4) + 4 r, col = l & 15)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int r = 16 * w + lk + 4 * r4;
      sLi[r][li] = t0[r4];
      sLi[r][16 + li] = t1[r4];
      if (tj == 0 && row0 + r < nrow) {
        Lmat[(size_t)(row0 + r) * ld + pc + li] = t0[r4];
        Lmat[(size_t)(row0 + r) * ld + pc + 16 + li] = t1[r4];
      }
    }
  }
  __syncthreads();
  if (prof_) pc_[1] = clock64();
  if (ti != tj) {  // the tile's columns: their panel rows were staged in sLj with the first loads; solved in place (a
                   // wavefront reads and writes only its own 16 rows, and writes after its last read)
    f64x4 t0 = {0.0, 0.0, 0.0, 0.0}, t1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < kNB / 4; ++ks) {
      const double a = sLj[16 * w + li][4 * ks + lk];
      t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sX[li][4 * ks + lk], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sX[16 + li][4 * ks + lk], t1, 0, 0, 0);
    }
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      sLj[16 * w + lk + 4 * r4][li] = t0[r4];
      sLj[16 * w + lk + 4 * r4][16 + li] = t1[r4];
    }
  } else {
    for (int e = tid; e < 64 * kNB; e += 256) sLj[e / kNB][e % kNB] = sLi[e / kNB][e % kNB];
  }
  __syncthreads();
  if (prof_) pc_[2] = clock64();
  // trailing update of this tile: A_ij -= Li Lj^T (64 x 64, K = 32)
  f64x4 acc[4];
#pragma unroll
  for (int tq = 0; tq < 4; ++tq) acc[tq] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < kNB / 4; ++ks) {
    const double a = sLi[16 * w + li][4 * ks + lk];
#pragma unroll
    for (int tq = 0; tq < 4; ++tq) acc[tq] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sLj[16 * tq + li][4 * ks + lk], acc[tq], 0, 0, 0);
  }
  if (prof_) pc_[3] = clock64();
  const bool corner = (ti == 0 && tj == 0);  // the next diagonal block belongs to the lead workgroup (which reads its old values)
#pragma unroll
  for (int tq = 0; tq < 4; ++tq)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int rl = 16 * w + lk + 4 * r4, cl = 16 * tq + li;
      const int r = row0 + rl, c = col0 + cl;
      if (r < nrow && c <= r && !(corner && rl < kNB)) A[(size_t)r * ld + c] = old[tq][r4] - acc[tq][r4];
    }
  if (prof_) {
    pc_[4] = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pc_[5] = clock64();
    if (tid == 0)
      printf("tile (1,0) step 5, shader clocks from the first barrier: Li %lld  Lj %lld  update %lld  stores issued %lld  stores done %lld ; kernel start -> first barrier %lld\n",
             pc_[1] - pc_[0], pc_[2] - pc_[1], pc_[3] - pc_[2], pc_[4] - pc_[3], pc_[5] - pc_[4], pc_[0] - t_start_);
  }
}

// workgroup barrier that only waits for this wavefront's LDS traffic: __syncthreads() also drains the outstanding global
// loads (vmcnt(0)), which would serialise the prefetches below with the dependent chain
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// back substitution L^T y = z (z = row n of the factored augmented matrix), right-looking over 32-wide blocks from last to
// first.  One workgroup doing all of it is bound by a single CU's load rate (it reads all of L: 9.4 MB at ~50 GB/s =
// 0.19 ms at n = 1536), so the block rows are cut into chunks of kBackChunk: k_chol_back_chunk solves inside a chunk (one
// workgroup, reads only the chunk's diagonal triangle), k_chol_back_gemv then removes the chunk's y from every z left
// of it, one workgroup per 32 columns.  Inside the chunk: y_k = Linv_kk^T z_k (32x32 products + column sums), then
// z_j -= L[block k rows][j] . y_k for the chunk's j left of block k; L and Linv of a step do not depend on y and are
// loaded before the step's first barrier.
// candidate point and the scalars the trust-region logic needs:
//   xc = x - y * scale ; mail[2] = model_cost_change = (y.gs + sum D y^2) / 2 ; mail[3] = |step| ; mail[4] = |x|
//   mail[0] = cost, mail[1] = max |g| of the linearisation this step starts from (what k_post_reduce writes after a
//   stand-alone linearisation; inside the LM loop that launch is saved)
// Runs at the end of the LAST back-substitution launch (the workgroup that has just finished y; a launch of its own was 5 - 6 us
// of every LM iteration).
struct StepArgs {
  const double *x, *scale, *g, *diag, *lin_cost;
  double *xc, *mail, *host_xc;
  int on;
};
template <int NT = 1024>
__device__ __forceinline__ void lm_step(const double *x, const double *y, const double *scale, const double *g, const double *diag, int n,
                                        double *xc, double *mail, double *host_xc, const double *lin_cost) {
  __shared__ double s0[NT], s1[NT], s2[NT], s3[NT];
  const int tid = threadIdx.x;
  double mc = 0.0, sn = 0.0, xn = 0.0, gm = 0.0;
  for (int i = tid; i < n; i += NT) {
    const double d = -y[i] * scale[i];
    xc[i] = x[i] + d;
    host_xc[i] = x[i] + d;  // pinned host staging: an accepted candidate is host state (|x|, best point) without a copy node
    mc += y[i] * (g[i] * scale[i]) + diag[i] * y[i] * y[i];
    sn += d * d;
    xn += x[i] * x[i];
    gm = fmax(gm, fabs(g[i]));
  }
  s0[tid] = mc, s1[tid] = sn, s2[tid] = xn, s3[tid] = gm;
  __syncthreads();
  for (int st = NT / 2; st > 0; st >>= 1) {
    if (tid < st) {
      s0[tid] += s0[tid + st];
      s1[tid] += s1[tid + st];
      s2[tid] += s2[tid + st];
      s3[tid] = fmax(s3[tid], s3[tid + st]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    mail[2] = 0.5 * s0[0];
    mail[3] = sqrt(s1[0]);
    mail[4] = sqrt(s2[0]);
    mail[0] = lin_cost[0];
    mail[1] = s3[0];
  }
}

// (Round 4, tried: chunks of 16 block rows - two (column, quarter) items per thread -, so that the odometry step's 12 block rows
// and the facade's 9 - 11 are ONE launch instead of chunk + gemv + chunk, and C4's 24 are 16 + 8: correct, and no faster - 1 807 against
// 1 858 LM it/s at C4, the step's solve 2.29 against 2.25 ms: a step inside the wide chunk lasts as much longer as the launches saved.)
constexpr int kBackChunk = 8;
// (Round 5, measured against this kernel's 14.4 us per chunk of eight block rows at C4 and not kept: two LDS barriers per block row
// instead of four - the column sums of Linv^T z inside half a wavefront by shuffles, a column's update by neighbouring lanes -: 14.2 us,
// i.e. nothing, the barriers and the serial sums are not what a block row's 1.8 us are; on top of it the loads of L / Linv two block
// rows ahead in registers of their own (the eight steps unrolled, so that no copy of a stage waits for the load just issued) with
// this kernel's thread mapping: 19.5 us; the same with every load unconditional (clamped addresses, masked values - behind predicated
// loads the compiler waits with vmcnt(0)): 20 - 24 us.  More requests in flight through ONE CU make it slower: the chunk's ~290 KB
// of L arrive at ~20 GB/s, the request rate of a single CU - the bound the chunking itself (several CUs) was built against.)
__global__ void __launch_bounds__(1024) k_chol_back_chunk(const double *A, int ld, int n, const double *Linv, const double *zsrc,
                                                         double *y, int lo_blk, int hi_blk, StepArgs S) {
  __shared__ double sz[kBackChunk * kNB];
  __shared__ double sP[kNB][kNB + 1];
  __shared__ double sQ[4][kBackChunk * kNB];
  __shared__ double syk[kNB];
  const int tid = threadIdx.x, m = tid >> 5, c = tid & 31;
  const int base = lo_blk * kNB, span = (hi_blk - lo_blk) * kNB;
  if (tid < span) sz[tid] = (base + tid < n) ? zsrc[base + tid] : 0.0;
  lds_barrier();
  // L and Linv of a step do not depend on y: they are requested one step ahead (registers li / lv hold step kb while
  // the loads of step kb - 1 are in flight behind the barriers, which only wait for LDS)
  auto fetch = [&](int kb, double &li_out, double (&lv_out)[8]) {
    const int loc = (kb - lo_blk) * kNB;
    const int rows = min(kNB, n - kb * kNB);
    li_out = Linv[(size_t)kb * kNB * kNB + tid];
    const bool upd = tid < 4 * loc;
    const int j = upd ? tid % loc : 0, qr = upd ? tid / loc : 0;
    const double *col = A + (size_t)(kb * kNB + qr * 8) * ld + base + j;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) lv_out[rr] = (upd && qr * 8 + rr < rows) ? col[(size_t)rr * ld] : 0.0;
  };
  double li, lv[8];
  fetch(hi_blk - 1, li, lv);
  for (int kb = hi_blk - 1; kb >= lo_blk; --kb) {
    const int loc = (kb - lo_blk) * kNB;  // local index of block kb's first row = number of chunk columns left of it
    // thread = (column j < loc, quarter of the block's rows): 8 independent loads, the quarters are added in fixed order
    const bool upd = tid < 4 * loc;
    const int j = upd ? tid % loc : 0, qr = upd ? tid / loc : 0;
    double li_n = 0.0, lv_n[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (kb > lo_blk) fetch(kb - 1, li_n, lv_n);
    sP[m][c] = (m >= c) ? li * sz[loc + m] : 0.0;  // Linv[m][c] z[m]
    lds_barrier();
    if (tid < kNB) {
      double acc = 0.0;
#pragma unroll
      for (int q = 0; q < kNB; ++q) acc += sP[q][tid];
      acc = (kb * kNB + tid < n) ? acc : 0.0;
      syk[tid] = acc;
      sz[loc + tid] = acc;
    }
    lds_barrier();
    if (upd) {
      double acc = 0.0;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) acc += lv[rr] * syk[qr * 8 + rr];
      sQ[qr][j] = acc;
    }
    lds_barrier();
    if (tid < loc) sz[tid] -= (sQ[0][tid] + sQ[1][tid]) + (sQ[2][tid] + sQ[3][tid]);
    lds_barrier();
    li = li_n;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) lv[rr] = lv_n[rr];
  }
  if (tid < span && base + tid < n) y[base + tid] = sz[tid];
  if (S.on) {  // the last chunk: y is complete (the other chunks' parts were written by the launches before this one)
    __syncthreads();
    lm_step(S.x, y, S.scale, S.g, S.diag, n, S.xc, S.mail, S.host_xc, S.lin_cost);
  }
}

// y[j] = zsrc[j] - sum_{r in chunk} L[r][j] y[r] for the 32 columns j of this workgroup (all left of the chunk):
// thread = (column, 1/32 slice of the chunk's rows); slice sums combined in fixed order
__global__ void __launch_bounds__(1024) k_chol_back_gemv(const double *A, int ld, int n, const double *zsrc, double *y, int lo_blk,
                                                        int hi_blk) {
  __shared__ double sP[kNB][kNB + 1];
  const int tid = threadIdx.x, sl = tid >> 5, c = tid & 31;
  const int j = blockIdx.x * kNB + c;
  const int per = hi_blk - lo_blk;  // rows per slice ((hi - lo) * 32 rows / 32 slices), <= kBackChunk
  const int r0 = lo_blk * kNB + sl * per;
  double lv[kBackChunk], yv[kBackChunk];
#pragma unroll
  for (int q = 0; q < kBackChunk; ++q) {
    const bool ok = q < per && r0 + q < n;
    lv[q] = ok ? A[(size_t)(r0 + q) * ld + j] : 0.0;
    yv[q] = ok ? y[r0 + q] : 0.0;
  }
  double acc = 0.0;
#pragma unroll
  for (int q = 0; q < kBackChunk; ++q) acc += lv[q] * yv[q];
  sP[sl][c] = acc;
  __syncthreads();
  if (tid < kNB) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < kNB; ++q) t += sP[q][tid];
    const int jj = blockIdx.x * kNB + tid;
    y[jj] = zsrc[jj] - t;
  }
}

// y = L^-T z from the appended identity rows (chol_extra_tile): row r of E holds L^-T[r][.] in the panels 32 (r / 32) .. of Lmat's
// appended rows - written panel by panel by the steps - except the LAST panel, which has no step of its own: there E's updated
// values (A's appended rows; the identity for rows of the last block itself) still want Linv_last^T, applied to z's last block
// once (w) instead.  A workgroup = eight rows, a row = 32 lanes, one column of every panel per lane; sums in a fixed order.
__global__ void __launch_bounds__(256) k_back_mul(const double *__restrict__ A, const double *__restrict__ Lmat, int ld, int nblk, int n,
                                                  const double *__restrict__ Linv, double *__restrict__ y) {
  __shared__ double sw[kNB], swp[8][kNB];
  const int tid = threadIdx.x, row = tid >> 5, c = tid & 31;
  const int nrow = nblk * kNB, last = nblk - 1;
  const double *z = Lmat + (size_t)n * ld;
  const int nl = n - last * kNB;  // unknowns in the last block (0: it holds the augmented row and padding only)
  {  // w = Linv_last^T z_last over the block's unknowns: thread = (column, four rows), the eight groups added in order
    const double *Li = Linv + (size_t)last * kNB * kNB;
    double lv[4], zv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = 4 * row + j;
      const bool in = q >= c && q < nl;
      lv[j] = in ? Li[q * kNB + c] : 0.0;
      zv[j] = in ? z[last * kNB + q] : 0.0;
    }
    swp[row][c] = fma(lv[3], zv[3], fma(lv[2], zv[2], fma(lv[1], zv[1], lv[0] * zv[0])));
  }
  __syncthreads();
  if (tid < kNB) sw[tid] = ((swp[0][tid] + swp[1][tid]) + (swp[2][tid] + swp[3][tid])) + ((swp[4][tid] + swp[5][tid]) + (swp[6][tid] + swp[7][tid]));
  __syncthreads();
  const int r = blockIdx.x * 8 + row;
  const bool live = r < n;
  const int b = live ? r / kNB : last;
  const double *Lr = Lmat + (size_t)(nrow + (live ? r : 0)) * ld;
  double acc = 0.0;
  int k = b;
  for (; k + 4 <= last; k += 4) {
    const double l0 = Lr[k * kNB + c], l1 = Lr[(k + 1) * kNB + c], l2 = Lr[(k + 2) * kNB + c], l3 = Lr[(k + 3) * kNB + c];
    const double z0 = z[k * kNB + c], z1 = z[(k + 1) * kNB + c], z2 = z[(k + 2) * kNB + c], z3 = z[(k + 3) * kNB + c];
    acc = fma(l3, z3, fma(l2, z2, fma(l1, z1, fma(l0, z0, acc))));
  }
  for (; k < last; ++k) acc = fma(Lr[k * kNB + c], z[k * kNB + c], acc);
  if (nl > 0) {
    const double ev = (b == last) ? ((r - last * kNB == c) ? 1.0 : 0.0) : A[(size_t)(nrow + (live ? r : 0)) * ld + last * kNB + c];
    acc = fma(ev, sw[c], acc);
  }
  for (int mm = 16; mm >= 1; mm >>= 1) acc += __shfl_xor(acc, mm);
  if (live && c == 0) y[r] = acc;
}

#include "window_schur.inc"

// ---- host helpers ----------------------------------------------------------------------------------------------------
int sort_u32(wc_ctx *ctx, wc_window_state *W, uint32_t *kin, uint32_t *kout, uint32_t *vin, uint32_t *vout, size_t n,
             unsigned end_bit) {
  // (the Onesweep radix path at every size: rocPRIM's default is a merge sort of ~14 launches below 2^20 items - extract.hip's sort_pairs)
  using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;
  size_t tmp = 0;
  WC_HIP(ctx, rocprim::radix_sort_pairs<cfg>(nullptr, tmp, kin, kout, vin, vout, n, 0u, end_bit, ctx->stream));
  WC_TRY(wc_ensure(ctx, ctx->b_sorttmp, tmp));
  tmp = ctx->b_sorttmp.cap;
  WC_HIP(ctx, rocprim::radix_sort_pairs<cfg>(ctx->b_sorttmp.p, tmp, kin, kout, vin, vout, n, 0u, end_bit, ctx->stream));
  return WC_OK;
}

template <typename T>
int upload(wc_ctx *ctx, wc_buf &b, const std::vector<T> &v) {
  WC_TRY(wc_ensure(ctx, b, std::max<size_t>(v.size() * sizeof(T), 16)));
  // no synchronisation here: every vector passed in lives until the one hipStreamSynchronize at the end of wc_window_build
  if (!v.empty()) WC_HIP(ctx, hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return WC_OK;
}

struct Seg {
  uint32_t start, count, key;
};

// The segment search of both families in flight: head detection and the copy of (status words, head slots) into pinned memory are
// enqueued by build_families; collect_families waits for the event and orders the heads on the host (the host picks the IMU
// factors in the meantime).
struct FamilyJob {
  uint32_t nb = 0, nu = 0, cap = 0;  // cap = number of possible keys (ns^2 + ns + 1)
  int ns = 0;
  const uint32_t *h_st = nullptr;
  const uint32_t *h_heads = nullptr;
  hipEvent_t done = nullptr;
};

int collect_families(wc_ctx *ctx, const FamilyJob &J, std::vector<Seg> &segs_b, std::vector<Seg> &segs_u) {
  segs_b.clear(), segs_u.clear();
  if (J.nb + J.nu == 0) return WC_OK;
  WC_HIP(ctx, hipEventSynchronize(J.done));
  const uint32_t *st = J.h_st;
  if (st[1] & 2u) return wc_fail(ctx, WC_ERR_ORDER, "correspondence is not (older, newer)");
  if (st[1] & 1u) return wc_fail(ctx, WC_ERR_RANGE, "surfel timestamp outside the sample-state range");
  const uint32_t ns = (uint32_t)J.ns, ubase = ns * ns, n = J.nb + J.nu;
  std::vector<std::pair<uint32_t, uint32_t>> heads;  // (position, key) in key order = position order (the keys are sorted)
  heads.reserve(4096);
  for (uint32_t key = 0; key < J.cap; ++key) {
    const uint32_t v = J.h_heads[key];
    if (!v) continue;
    if (v > n || (!heads.empty() && v - 1u <= heads.back().first)) return wc_fail(ctx, WC_ERR_RANGE, "segment heads out of order (key %u at %u)", key, v - 1u);
    heads.push_back({v - 1u, key});
  }
  const size_t nh = heads.size();
  for (size_t i = 0; i < nh; ++i) {
    const uint32_t pos = heads[i].first, sk = heads[i].second;
    const bool unary = pos >= J.nb;  // (a key of the other family at the boundary is a head of its own: no segment straddles it)
    const uint32_t fam_end = unary ? J.nb + J.nu : J.nb;
    const uint32_t end = (i + 1 < nh && heads[i + 1].first < fam_end) ? heads[i + 1].first : fam_end;
    if (unary)
      segs_u.push_back({pos - J.nb, end - pos, sk - ubase});
    else
      segs_b.push_back({pos, end - pos, (sk / ns) | ((sk % ns) << 16)});
  }
  return WC_OK;
}

}  // namespace

void wc_window_free(wc_ctx *ctx) {
  wc_window_state *W = ctx->win;
  if (!W) return;
  wc_buf *all[] = {&W->times_d, &W->brec, &W->bkey, &W->borig, &W->urec, &W->ukey, &W->uorig, &W->lists, &W->partial,
                   &W->x, &W->xc, &W->lin, &W->lin_alt, &W->Linv, &W->Lmat, &W->reduce, &W->scale, &W->diag, &W->A, &W->y,
                   &W->mail, &W->cost_part, &W->keys_tmp[0], &W->keys_tmp[1], &W->vals_tmp[0], &W->vals_tmp[1], &W->heads, &W->status,
                   &W->pcr_D[0], &W->pcr_D[1], &W->pcr_A[0], &W->pcr_A[1], &W->pcr_R[0], &W->pcr_R[1], &W->yred};
  for (wc_buf *b : all) wc_buf_release(ctx, *b);
  (void)hipStreamSynchronize(ctx->stream);  // (the releases are stream ordered)
  if (W->h_pin) (void)hipHostFree(W->h_pin);
  if (W->h_up) (void)hipHostFree(W->h_up);
  for (hipEvent_t e : W->fam_done)
    if (e) (void)hipEventDestroy(e);
  if (W->side) {
    (void)hipStreamSynchronize(W->side);
    (void)hipStreamDestroy(W->side);
  }
  if (W->ev_side_go) (void)hipEventDestroy(W->ev_side_go);
  if (W->ev_side_done) (void)hipEventDestroy(W->ev_side_done);
  delete W;
  ctx->win = nullptr;
}

namespace {

// enqueue the surfel records of both families (binary, unary): keys -> sort -> packed records -> segment heads -> copy to the
// pinned staging area.  ONE chain for both (FamArgs): the sorted order holds the binary family first.
int build_families(wc_ctx *ctx, wc_window_state *W, const FamArgs &B0, const FamArgs &U0, FamilyJob &J) {
  J = FamilyJob{};
  const uint32_t nb = B0.n, nu = U0.n, n = nb + nu;
  if (n == 0) return WC_OK;
  const uint64_t ns = (uint64_t)W->ns, maxkey = ns * ns + ns;
  const uint32_t cap = (uint32_t)(maxkey + 1);  // one head slot per possible key
  {  // the heads' array: zero at rest (cleared behind every read-back); a fresh or longer one is cleared here
    void *before = W->heads.p;
    WC_TRY(wc_ensure(ctx, W->heads, (size_t)cap * 4));
    if (W->heads.p != before) W->heads_zero = 0;
    if (W->heads_zero < cap) WC_HIP(ctx, hipMemsetAsync(W->heads.p, 0, (size_t)cap * 4, ctx->stream));
    W->heads_zero = 0;  // (until the memset behind the read-back is enqueued)
  }
  WC_TRY(wc_ensure(ctx, W->keys_tmp[0], (size_t)n * 4));
  WC_TRY(wc_ensure(ctx, W->keys_tmp[1], (size_t)n * 4));
  WC_TRY(wc_ensure(ctx, W->vals_tmp[0], (size_t)n * 4));
  WC_TRY(wc_ensure(ctx, W->vals_tmp[1], (size_t)n * 4));
  WC_TRY(wc_ensure(ctx, W->brec, std::max<size_t>((size_t)nb * 15 * 8, 16)));
  WC_TRY(wc_ensure(ctx, W->bkey, std::max<size_t>((size_t)nb * 4, 16)));
  WC_TRY(wc_ensure(ctx, W->borig, std::max<size_t>((size_t)nb * 4, 16)));
  WC_TRY(wc_ensure(ctx, W->urec, std::max<size_t>((size_t)nu * 11 * 8, 16)));
  WC_TRY(wc_ensure(ctx, W->ukey, std::max<size_t>((size_t)nu * 4, 16)));
  WC_TRY(wc_ensure(ctx, W->uorig, std::max<size_t>((size_t)nu * 4, 16)));
  WC_TRY(wc_ensure(ctx, W->status, 64 * 4));
  if (!W->fam_done[0]) WC_HIP(ctx, hipEventCreateWithFlags(&W->fam_done[0], hipEventDisableTiming));
  FamArgs B = B0, U = U0;
  B.rec = (double *)W->brec.p, B.key_out = (uint32_t *)W->bkey.p, B.orig_out = (uint32_t *)W->borig.p;
  U.rec = (double *)W->urec.p, U.key_out = (uint32_t *)W->ukey.p, U.orig_out = (uint32_t *)W->uorig.p;
  uint32_t *d_st = (uint32_t *)W->status.p;  // (k_pair_keys: word 1 = flags, k_seg_heads: word 2 = heads)
  if (!W->status_clear) {  // zero at rest: cleared behind the read-back of every build (a memset in front of the chain was ~6 us of it)
    WC_HIP(ctx, hipMemsetAsync(d_st, 0, 16 * 4, ctx->stream));
  }
  W->status_clear = false;  // (until the memset behind the read-back is enqueued: a failure in between leaves it dirty)
  const unsigned grid = (n + 255) / 256;
  k_pair_keys<<<grid, 256, 0, ctx->stream>>>(B, U, (const double *)W->times_d.p, W->ns, (uint32_t *)W->keys_tmp[0].p, (uint32_t *)W->vals_tmp[0].p, d_st);
  // (k_pair_keys' flags are read back with the segment heads: a flagged record gets its family's key 0, so everything downstream is safe)
  unsigned bits = 1;
  while ((1ull << bits) < maxkey + 1) ++bits;
  WC_TRY(sort_u32(ctx, W, (uint32_t *)W->keys_tmp[0].p, (uint32_t *)W->keys_tmp[1].p, (uint32_t *)W->vals_tmp[0].p,
                  (uint32_t *)W->vals_tmp[1].p, n, bits));
  // heads of the SORT keys, by key (k_seg_heads); the host tells the families apart by position.  The heads leave BEFORE the records
  // are formed: the host cuts pieces and builds the gather's lists while k_build_records runs.
  uint32_t *d_heads = (uint32_t *)W->heads.p;
  k_seg_heads<<<(n + 1023) / 1024, 1024, 0, ctx->stream>>>((const uint32_t *)W->keys_tmp[1].p, n, d_heads);
  WC_HIP(ctx, hipGetLastError());
  char *h = (char *)W->h_pin;
  WC_HIP(ctx, hipMemcpyAsync(h, d_st, 16, hipMemcpyDeviceToHost, ctx->stream));
  WC_HIP(ctx, hipMemcpyAsync(h + 64, d_heads, (size_t)cap * 4, hipMemcpyDeviceToHost, ctx->stream));
  WC_HIP(ctx, hipEventRecord(W->fam_done[0], ctx->stream));
  WC_HIP(ctx, hipMemsetAsync(d_heads, 0, (size_t)cap * 4, ctx->stream));  // (zero at rest again, behind the read-back)
  W->heads_zero = cap;
  WC_HIP(ctx, hipMemsetAsync(d_st, 0, 16 * 4, ctx->stream));
  W->status_clear = true;
  k_build_records<<<grid, 256, 0, ctx->stream>>>(B, U, (const uint32_t *)W->vals_tmp[1].p, (const uint32_t *)W->keys_tmp[1].p,
                                                (const double *)W->times_d.p, W->ns, W->wp.sigma0_sq);
  WC_HIP(ctx, hipGetLastError());
  J.nb = nb, J.nu = nu, J.cap = cap, J.ns = W->ns, J.h_st = (const uint32_t *)h, J.h_heads = (const uint32_t *)(h + 64), J.done = W->fam_done[0];
  return WC_OK;
}

}  // namespace

static int window_build_impl(wc_ctx *ctx, const wc_surfel *d_sld_surf, const wc_pose *d_sld_pose, const wc_pair *d_pairs_sld,
                             uint64_t n_pairs_sld, const wc_surfel *d_fix_surf, const wc_pose *d_fix_pose,
                             const wc_pair *d_pairs_fix, uint64_t n_pairs_fix, const wc_imu_state *h_imu, uint64_t n_imu,
                             const double *h_sample_times, uint64_t ns_, const double *h_grav, int fix_first_pos, bool sharded) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !h_sample_times || ns_ < 2 || ns_ > 340 || !h_grav) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);  // 12 ns <= 4096 unknowns: dense H (134 MB), the largest window the solve has been exercised on; the reference's default window has 82 sample states
  if (n_pairs_sld >= (1ull << 31) || n_pairs_fix >= (1ull << 31)) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  WC_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->win) {
    ctx->win = new wc_window_state;
    ctx->win->reduce.plain = ctx->win->mail.plain = true;  // (the all-reduce's buffers: ordinary hipMalloc memory for RCCL)
  }
  wc_window_state *W = ctx->win;
  W->built = false;
  W->sharded = sharded;
  W->two_coll = false;  // (wc_window_build_sharded sets it behind a successful build)
  const wc_params &P = ctx->P;
  const int ns = (int)ns_;
  W->ns = ns;
  W->n = 12 * ns;
  W->np = ((W->n + 1 + kNB - 1) / kNB) * kNB;
  W->ld = W->np;
  W->times.assign(h_sample_times, h_sample_times + ns);
  WinParams &wp = W->wp;
  wp.sigma0_sq = P.surfel_sigma0 * P.surfel_sigma0;
  wp.cauchy_b = P.cauchy_a * P.cauchy_a;
  wp.inv_cauchy_b = 1.0 / wp.cauchy_b;
  wp.w_gyr = P.w_gyr, wp.w_acc = P.w_acc, wp.w_bg = P.w_bg, wp.w_ba = P.w_ba, wp.dt = P.imu_dt;
  for (int i = 0; i < 3; ++i) wp.grav[i] = h_grav[i];
  wp.quirks = P.reference_quirks;
  wp.ns = ns;
  wp.fix_first = fix_first_pos ? 1 : 0;
  // host arrays the asynchronous uploads below read; the guard (destroyed first) waits for the stream on every way out
  std::vector<ImuRec> irecs;
  std::vector<Piece> pieces;
  std::vector<uint32_t> src_begin, gsrc_begin, heavy, pair_off;
  // Nothing enqueued below reads a vector of this scope: what the host builds leaves through the pinned staging buffers (h_up), so the
  // call returns with its last copies still in flight (the solve is enqueued behind them on the same stream; the closing
  // hipStreamSynchronize of rounds 1 - 3 was 30 - 40 us of every build).  The next build waits for fam_done[1] before it writes staging.
  if (W->fam_done[1]) WC_HIP(ctx, hipEventSynchronize(W->fam_done[1]));
  WC_TRY(upload(ctx, W->times_d, W->times));  // (W->times outlives the copy)

  std::vector<Seg> segs_b, segs_u;
  W->nb = (uint32_t)n_pairs_sld;
  W->nu = (uint32_t)n_pairs_fix;
  static const bool tdbg = wc_log_env("WC_WIN_DEBUG");  // (debug knobs: read once per process, never on a later call)
  static const bool gdbg = wc_log_env("WC_DEBUG_GATHER");
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto t_a = tnow();
  {  // staging for both families' (status, heads); heads of both in one device buffer
    const size_t half = 64 + ((size_t)ns * ns + 1) * 8, need = 2 * ((half + 127) / 128 * 128);
    if (W->h_pin_cap < need) {
      if (W->h_pin) (void)hipHostFree(W->h_pin);
      W->h_pin = nullptr, W->h_pin_cap = 0;
      // (sized for the largest window wc_window_build accepts, once: a pinned allocation is 0.3 - 0.5 ms, and a window that gains
      // sample states from sweep to sweep - the facade's first seconds - asked for a larger one on every call)
      const size_t half_max = 64 + ((size_t)340 * 340 + 1) * 8, want = std::max(need, 2 * ((half_max + 127) / 128 * 128));
      WC_HIP(ctx, hipHostMalloc(&W->h_pin, want));
      W->h_pin_cap = want;
    }
  }
  // (Round 3, WC_WIN_DEBUG: of the odometry step's 0.40 ms build the host waited ~110 us for the binary family's chain - keys, three
  // sort passes, records, segment heads, then the same for the unary family: ~240 us of small launches - and computed for ~65 us.  Round 4:
  // one chain for both families, the heads leave before the records are formed, the host's lists go out in one copy from pinned staging,
  // no closing stream wait: 0.22 - 0.25 ms, of which the host now sees ~85 us of enqueueing, ~25 of waiting for the heads and ~65 of its own
  // work - WC_WIN_DEBUG prints them per phase.)
  FamilyJob job;
  {
    const FamArgs fb{d_sld_surf, d_sld_surf, d_sld_pose, d_sld_pose, d_pairs_sld, W->nb, nullptr, nullptr, nullptr};
    const FamArgs fu{d_fix_surf, d_sld_surf, d_fix_pose, d_sld_pose, d_pairs_fix, W->nu, nullptr, nullptr, nullptr};
    WC_TRY(build_families(ctx, W, fb, fu, job));
  }

  // IMU factors (BuildImuResiduals, lidar_odometry.cc:319-363), selected on the host: a few thousand records
  std::vector<Seg> segs_i;
  if (h_imu && n_imu >= 3) {
    for (uint64_t i = 0; i + 2 < n_imu; ++i) {
      if (h_imu[i].t < W->times.front()) continue;
      if (h_imu[i + 2].t > W->times.back()) break;
      const int it = (int)(std::upper_bound(W->times.begin(), W->times.end(), h_imu[i].t) - W->times.begin());
      if (it == 0 || it == ns) return wc_fail(ctx, WC_ERR_RANGE, "IMU state outside the sample-state range");
      ImuRec r;
      r.i1 = h_imu[i], r.i2 = h_imu[i + 1], r.i3 = h_imu[i + 2];
      r.sp1 = it - 1;
      r.mode = (it == ns - 1) ? 1 : 0;
      if (!segs_i.empty() && segs_i.back().key == (uint32_t)r.sp1 && segs_i.back().count < (uint32_t)kImuMax)
        segs_i.back().count++;
      else
        segs_i.push_back({(uint32_t)irecs.size(), 1, (uint32_t)r.sp1});
      irecs.push_back(r);
    }
  }
  W->ni = (uint32_t)irecs.size();

  auto t_b = tnow();
  // pieces + the CSR source lists of the gather
  uint32_t off = 0;
  auto cut = [&](const std::vector<Seg> &segs, uint32_t T, bool split, uint32_t piece_max = kPiece) {
    for (const Seg &s : segs) {
      for (uint32_t b = 0; b < s.count; b += split ? piece_max : s.count) {
        const uint32_t c = split ? std::min<uint32_t>(piece_max, s.count - b) : s.count;
        pieces.push_back({s.start + b, c, s.key, off});
        off += T * (T + 1) / 2;
      }
    }
  };
  // (longest pieces first inside a family: workgroups are dispatched in piece order, and a full piece started last would
  // run alone at the end of the launch)
  auto by_size = [&](size_t first, uint32_t piece_max = kPiece) {  // stable counting sort of pieces[first..) by count, descending (counts are 1..piece_max)
    std::vector<uint32_t> pos(piece_max + 2, 0);
    for (size_t i = first; i < pieces.size(); ++i) pos[piece_max - pieces[i].count + 1]++;
    for (uint32_t c = 0; c <= piece_max; ++c) pos[c + 1] += pos[c];
    std::vector<Piece> out(pieces.size() - first);
    for (size_t i = first; i < pieces.size(); ++i) out[pos[piece_max - pieces[i].count]++] = pieces[i];
    std::copy(out.begin(), out.end(), pieces.begin() + first);
  };
  WC_TRY(collect_families(ctx, job, segs_b, segs_u));
  auto t_b1 = tnow();
  cut(segs_b, 25, true);
  W->npiece_b = (uint32_t)pieces.size();
  by_size(0);
  W->npiece_b_big = 0;  // (sorted by size, largest first: the pieces of more than half a workgroup's records)
  while (W->npiece_b_big < W->npiece_b && pieces[W->npiece_b_big].count > (uint32_t)kPiece / 2u) ++W->npiece_b_big;
  const uint32_t npairs = (uint32_t)(ns * (ns + 1) / 2);
  W->npairs = npairs;
  src_begin.assign(npairs + 1, 0), gsrc_begin.assign(ns + 1, 0);
  std::vector<uint32_t> imu_cnt(npairs, 0);
  auto pair_id = [&](int I, int J) { return (uint32_t)(I * ns - I * (I - 1) / 2 + (J - I)); };
  for (size_t pi = 0; pi < pieces.size(); ++pi) {  // (binary pieces only: blocks_of below, first branch)
    const Piece &pc = pieces[pi];
    const int sp1l = pc.key & 0xFFFF, sp2l = pc.key >> 16;
    int blk[4] = {sp1l, sp1l + 1, 0, 0}, nblk = 2;
    if (sp2l > sp1l + 1)
      blk[2] = sp2l, blk[3] = sp2l + 1, nblk = 4;
    else if (sp2l == sp1l + 1)
      blk[2] = sp2l + 1, nblk = 3;
    for (int p = 0; p < nblk; ++p) {
      gsrc_begin[blk[p] + 1]++;
      for (int q = p; q < nblk; ++q) src_begin[pair_id(blk[p], blk[q]) + 1]++;
    }
  }
  // (unary pieces of up to kLinChunksU chunks of kPiece records - lin_surfel_body; development option lin_unary_chunks: 1 = rounds 2 - 5's
  // pieces of one chunk)
  // Measured (profiles/dev/ab_lin2.py, device time of one linearisation): C4's 12 299 pieces 0.124 -> 0.110 ms with four chunks (9 329
  // pieces; two chunks 0.115, eight 0.118: a piece of 2 048 records is the launch's tail), the step-sized window's 3 492 pieces 0.044
  // either way, a window of 1 336 pieces 0.0230 -> 0.0238: the long pieces are for windows whose launch is several rounds of the chip's
  // workgroup slots.
  uint32_t nrec_u = 0;
  for (const Seg &sg : segs_u) nrec_u += sg.count;
  const bool many = pieces.size() + nrec_u / kPiece > 6000u;
  // In the LM loop (x != 0, the solve's kernels between two linearisations) the long pieces LOSE: they only fit the register file as a launch
  // of their own (k_lin_surfel<12, true, true>; inside k_lin_fused the chunk loop spilled), and k_lin_fused without the unary family +
  // that launch last 99 us + a boundary against 87 fused - C4 0.406 -> 0.417 ms per LM iteration.  Default: one chunk; the option keeps
  // the long pieces for A/B runs.
  (void)many;
  const int want_ch = ctx->dev.lin_unary_chunks > 0 ? ctx->dev.lin_unary_chunks : 1;
  const uint32_t upiece = (uint32_t)kPiece * (uint32_t)std::max(1, std::min(kLinChunksU, want_ch));
  W->unary_multi = upiece > (uint32_t)kPiece;
  cut(segs_u, 13, true, upiece);
  W->npiece_u = (uint32_t)pieces.size() - W->npiece_b;
  by_size(W->npiece_b, upiece);
  cut(segs_i, 37, false);
  W->npiece_i = (uint32_t)pieces.size() - W->npiece_b - W->npiece_u;
  W->npart_doubles = off;
  auto t_b2 = tnow();
  if (tdbg) {  // piece sizes per family
    uint32_t hist[3][6] = {{0}};
    for (size_t i = 0; i < pieces.size(); ++i) {
      const int fam = i < W->npiece_b ? 0 : (i < W->npiece_b + W->npiece_u ? 1 : 2);
      const uint32_t c = pieces[i].count;
      hist[fam][c <= 16 ? 0 : c <= 32 ? 1 : c <= 64 ? 2 : c <= 128 ? 3 : c < (uint32_t)kPiece ? 4 : 5]++;
    }
    for (int fam = 0; fam < 3; ++fam)
      fprintf(stderr, "[win] family %d pieces by count <=16 %u, <=32 %u, <=64 %u, <=128 %u, <%d %u, full %u\n", fam, hist[fam][0], hist[fam][1], hist[fam][2],
              hist[fam][3], kPiece, hist[fam][4], hist[fam][5]);
  }

  // CSR source lists in two passes over the pieces (count, fill): sources of a pair / block in piece order
  auto blocks_of = [&](size_t pi, int blk[4], uint8_t &w, uint8_t &T) -> int {
    const Piece &pc = pieces[pi];
    if (pi < W->npiece_b) {
      const int sp1l = pc.key & 0xFFFF, sp2l = pc.key >> 16;
      w = 6, T = 25;
      blk[0] = sp1l, blk[1] = sp1l + 1;
      if (sp2l > sp1l + 1) {
        blk[2] = sp2l, blk[3] = sp2l + 1;
        return 4;
      }
      if (sp2l == sp1l + 1) {
        blk[2] = sp2l + 1;
        return 3;
      }
      return 2;
    }
    if (pi < W->npiece_b + W->npiece_u) {
      w = 6, T = 13, blk[0] = (int)pc.key, blk[1] = (int)pc.key + 1;
      return 2;
    }
    w = 12, T = 37;
    blk[0] = (int)pc.key, blk[1] = (int)pc.key + 1, blk[2] = (int)pc.key + 2;
    return ((int)pc.key + 1 == ns - 1) ? 2 : 3;
  };
  for (size_t pi = W->npiece_b; pi < pieces.size(); ++pi) {  // (the binary pieces have been counted above)
    int blk[4];
    uint8_t w, T;
    const int nblk = blocks_of(pi, blk, w, T);
    for (int p = 0; p < nblk; ++p) {
      gsrc_begin[blk[p] + 1]++;
      for (int q = p; q < nblk; ++q) {
        src_begin[pair_id(blk[p], blk[q]) + 1]++;
        if (w == 12) imu_cnt[pair_id(blk[p], blk[q])]++;  // (a pair's sources are in piece order: the IMU pieces' come last)
      }
    }
  }
  for (uint32_t i = 0; i < npairs; ++i) src_begin[i + 1] += src_begin[i];
  for (int i = 0; i < ns; ++i) gsrc_begin[i + 1] += gsrc_begin[i];
  pair_off.assign(npairs + 1, 0);
  {
    uint32_t o = 0, pid = 0;
    for (int I = 0; I < ns; ++I)
      for (int J = I; J < ns; ++J) {
        pair_off[pid++] = o;
        o += (J - I <= 2) ? 144u : 36u;
      }
    pair_off[npairs] = o;
    W->red_H = o;
  }
  // who sums which pair in k_gather: heavy (long source lists), far (more than two sample blocks apart, surfel sources only: the
  // 6 x 6 pose corner; a far pair without sources is only written into the multi-GPU reduction buffer - in H its entries are zero
  // from the build), near (the rest)
  std::vector<uint32_t> near_list;
  std::vector<FarJob> far_list;
  {
    uint32_t pid = 0;
    for (int I = 0; I < ns; ++I)
      for (int J = I; J < ns; ++J, ++pid) {
        const uint32_t cnt = src_begin[pid + 1] - src_begin[pid];
        if (cnt > kHeavySrc)
          heavy.push_back(pid), heavy.push_back(src_begin[pid + 1] - imu_cnt[pid]);  // {pair, where its IMU sources begin}
        else if (J - I > 2) {
          far_list.push_back({src_begin[pid], src_begin[pid + 1], pid, (uint16_t)I, (uint16_t)J});
        } else
          near_list.push_back(pid);
      }
  }
  W->nheavy = (uint32_t)heavy.size() / 2, W->nnear = (uint32_t)near_list.size(), W->nfar = (uint32_t)far_list.size();
  // Every list the host builds lives in ONE device arena and leaves in ONE copy out of ONE pinned staging buffer, in which the two
  // large ones (the gather's source lists, ~0.5 MB in the odometry step) are written in place.  (Rounds 3 - 4: eight buffers, eight
  // copies - out of pageable vectors at first: the runtime stages such a copy and returns when it is through, 10 - 35 us each -,
  // then out of a staging buffer the vectors were copied into; a kernel trace of the odometry step showed the device idle for
  // ~190 us between k_build_records and the solve's first kernel while the host copied and enqueued.)
  auto t_b3 = tnow();
  const size_t n_src = src_begin[npairs], n_gsrc = gsrc_begin[ns];
  struct Part {
    wc_buf *b;
    const void *h;  // nullptr: written in place below
    size_t bytes, off;
  };
  Part parts[] = {{&W->irec, irecs.data(), irecs.size() * sizeof(ImuRec), 0}, {&W->pair_off, pair_off.data(), pair_off.size() * 4, 0},
                  {&W->heavy, heavy.data(), heavy.size() * 4, 0},             {&W->pieces, pieces.data(), pieces.size() * sizeof(Piece), 0},
                  {&W->src, nullptr, n_src * sizeof(Src), 0},                 {&W->src_begin, src_begin.data(), src_begin.size() * 4, 0},
                  {&W->gsrc, nullptr, n_gsrc * sizeof(GSrc), 0},              {&W->gsrc_begin, gsrc_begin.data(), gsrc_begin.size() * 4, 0},
                  {&W->near_l, near_list.data(), near_list.size() * 4, 0},    {&W->far_l, far_list.data(), far_list.size() * sizeof(FarJob), 0}};
  size_t total = 0;
  for (Part &pt : parts) {
    pt.off = total;
    total += (std::max<size_t>(pt.bytes, 16) + 255) / 256 * 256;
  }
  if (W->h_up_cap < total) {
    WC_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (copies of an earlier build out of the old buffer)
    if (W->h_up) (void)hipHostFree(W->h_up);
    W->h_up = nullptr, W->h_up_cap = 0;
    // (never below 32 MB: pinning pages costs ~1 ms per MB, and with a 4 MB floor the facade's growing window re-pinned 8 MB inside
    // a sweep - a build of 8 ms among builds of 0.25, round 5)
    const size_t want = std::max<size_t>(2 * total, (size_t)32 << 20);
    WC_HIP(ctx, hipHostMalloc(&W->h_up, want));
    W->h_up_cap = want;
  }
  WC_TRY(wc_ensure(ctx, W->lists, total));
  for (Part &pt : parts) {  // the lists' buffers are views into the arena (not freed on their own)
    pt.b->p = (char *)W->lists.p + pt.off, pt.b->cap = pt.bytes;
    if (pt.h && pt.bytes) std::memcpy((char *)W->h_up + pt.off, pt.h, pt.bytes);
  }
  auto t_b4 = tnow();
  {  // CSR source lists, second pass (fill): sources of a pair / block in piece order, straight into the staging buffer
    Src *src = (Src *)((char *)W->h_up + parts[4].off);
    GSrc *gsrc = (GSrc *)((char *)W->h_up + parts[6].off);
    std::vector<uint32_t> cur(src_begin.begin(), src_begin.end() - 1), gcur(gsrc_begin.begin(), gsrc_begin.end() - 1);
    for (size_t pi = 0; pi < pieces.size(); ++pi) {
      int blk[4];
      uint8_t w, T;
      const int nblk = blocks_of(pi, blk, w, T);
      const uint32_t po = pieces[pi].part_off;
      for (int p = 0; p < nblk; ++p) {
        gsrc[gcur[blk[p]]++] = {po, (uint8_t)p, w, T, 0};
        for (int q = p; q < nblk; ++q) src[cur[pair_id(blk[p], blk[q])]++] = {po, (uint8_t)p, (uint8_t)q, w, T};
      }
    }
  }
  if (gdbg) {
    uint32_t mx = 0;
    for (uint32_t i = 0; i < npairs; ++i) mx = std::max(mx, src_begin[i + 1] - src_begin[i]);
    fprintf(stderr, "gather: %u pairs, %u heavy, %zu sources (max %u per pair), %zu g-sources, %zu pieces\n", npairs, W->nheavy, n_src, mx, n_gsrc, pieces.size());
  }
  auto t_c = tnow();
  WC_HIP(ctx, hipMemcpyAsync(W->lists.p, W->h_up, total, hipMemcpyHostToDevice, ctx->stream));

  // (the buffers whose size goes with the SQUARE of the sample states are allocated for at least 96 of them - the reference's default
  // window has 82 -: a window that gains ten sample states per sweep, the facade's first seconds, otherwise re-allocates four of them on
  // most calls, 0.1 - 0.7 ms of hipFree / hipMalloc each time; 4 x 10.6 MB)
  const size_t n = W->n, n_al = std::max<size_t>(n, 12 * 96), np_al = ((n_al + 1 + kNB - 1) / kNB) * kNB;
  WC_TRY(wc_ensure(ctx, W->partial, ((size_t)off + pieces.size()) * 8 + 64));  // the pieces' partials, then one cost per piece
  WC_TRY(wc_ensure(ctx, W->x, n_al * 8));
  WC_TRY(wc_ensure(ctx, W->xc, n_al * 8));
  WC_TRY(wc_ensure(ctx, W->lin, (n_al * n_al + np_al + 2) * 8));
  WC_TRY(wc_ensure(ctx, W->lin_alt, (n_al * n_al + np_al + 2) * 8));
  // (k_gather writes a far pair's 6 x 6 pose corner only: the rest of those blocks is zero from here on)
  WC_HIP(ctx, hipMemsetAsync(W->lin.p, 0, n * n * 8, ctx->stream));
  WC_HIP(ctx, hipMemsetAsync(W->lin_alt.p, 0, n * n * 8, ctx->stream));
  W->lin_sel = 0;
  WC_TRY(wc_ensure(ctx, W->Linv, np_al * kNB * 8));
  WC_TRY(wc_ensure(ctx, W->scale, n_al * 8));
  WC_TRY(wc_ensure(ctx, W->diag, n_al * 8));
  WC_TRY(wc_ensure(ctx, W->A, np_al * np_al * 8));
  WC_TRY(wc_ensure(ctx, W->Lmat, np_al * np_al * 8));
  WC_TRY(wc_ensure(ctx, W->y, np_al * 8));
  WC_TRY(wc_ensure(ctx, W->mail, 64 * 8));
  WC_HIP(ctx, hipMemsetAsync((double *)W->mail.p + 60, 0, 24, ctx->stream));  // k_gather's count of finished g / cost workgroups, their maximum of |g|; [62]: k_schur_bias_y_step's count
  const size_t ncb = (W->nb + 255) / 256 + (W->nu + 255) / 256 + (W->ni + 255) / 256 + 8;
  WC_TRY(wc_ensure(ctx, W->cost_part, ncb * 8));
  if (!W->fam_done[1]) WC_HIP(ctx, hipEventCreateWithFlags(&W->fam_done[1], hipEventDisableTiming));
  WC_HIP(ctx, hipEventRecord(W->fam_done[1], ctx->stream));
  if (tdbg) {
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    fprintf(stderr, "[win] build: families + imu %.0f us, pieces + source lists (host) %.0f us (wait for the heads %.0f, cut %.0f, count %.0f, layout %.0f, fill %.0f), uploads %.0f us\n", us(t_a, t_b),
            us(t_b, t_c), us(t_b, t_b1), us(t_b1, t_b2), us(t_b2, t_b3), us(t_b3, t_b4), us(t_b4, t_c), us(t_c, tnow()));
  }
  W->built = true;
  return WC_OK;
}

extern "C" int wc_window_build(wc_ctx *ctx, const wc_surfel *d_sld_surf, const wc_pose *d_sld_pose, const wc_pair *d_pairs_sld,
                               uint64_t n_pairs_sld, const wc_surfel *d_fix_surf, const wc_pose *d_fix_pose,
                               const wc_pair *d_pairs_fix, uint64_t n_pairs_fix, const wc_imu_state *h_imu, uint64_t n_imu,
                               const double *h_sample_times, uint64_t ns_, const double *h_grav, int fix_first_pos) {
  return window_build_impl(ctx, d_sld_surf, d_sld_pose, d_pairs_sld, n_pairs_sld, d_fix_surf, d_fix_pose, d_pairs_fix, n_pairs_fix, h_imu,
                           n_imu, h_sample_times, ns_, h_grav, fix_first_pos, false);
}

namespace {
int do_allreduce(wc_ctx *ctx, wc_window_state *W, double *d_buf, size_t count);
}

// The multi-GPU form (see include/wildcat_hip.h): the SAME replicated arguments on every rank; the library takes this rank's
// contiguous share of both correspondence lists and of the IMU state triples (factor i = states i, i + 1, i + 2: a rank's share
// of the factors is its states plus the two that follow), and checks with one small all-reduce that the ranks' shares add up
// to the whole problem - ranks that were handed different lists, or a world in which not every rank made this call, fail here
// instead of summing H, g and the cost a wrong number of times.
extern "C" int wc_window_build_sharded(wc_ctx *ctx, const wc_surfel *d_sld_surf, const wc_pose *d_sld_pose, const wc_pair *d_pairs_sld,
                                       uint64_t n_pairs_sld, const wc_surfel *d_fix_surf, const wc_pose *d_fix_pose,
                                       const wc_pair *d_pairs_fix, uint64_t n_pairs_fix, const wc_imu_state *h_imu, uint64_t n_imu,
                                       const double *h_sample_times, uint64_t ns_, const double *h_grav, int fix_first_pos) {
  if (!ctx) return WC_ERR_ARG;
  if (!ctx->have_comm || ctx->comm.world <= 1)  // one rank: the whole problem, no collective
    return window_build_impl(ctx, d_sld_surf, d_sld_pose, d_pairs_sld, n_pairs_sld, d_fix_surf, d_fix_pose, d_pairs_fix, n_pairs_fix, h_imu,
                             n_imu, h_sample_times, ns_, h_grav, fix_first_pos, false);
  if (!ctx->comm.allreduce_f64) return wc_fail(ctx, WC_ERR_ARG, "%s: the communicator has no all-reduce", __func__);
  const uint64_t w = (uint64_t)ctx->comm.world, r = (uint64_t)ctx->comm.rank;
  auto share = [&](uint64_t n, uint64_t &lo, uint64_t &cnt) {
    lo = (n * r) / w;
    cnt = (n * (r + 1)) / w - lo;
  };
  uint64_t lo_b, n_b, lo_u, n_u, lo_i = 0, n_i = 0;
  share(n_pairs_sld, lo_b, n_b);
  share(n_pairs_fix, lo_u, n_u);
  const uint64_t n_fac = (h_imu && n_imu >= 3) ? n_imu - 2 : 0;
  // Round 6: the IMU factors are REPLICATED - a few hundred to two thousand factors whose family is a latency chain of ~15 us
  // whatever the window - so that everything the bias elimination reads (bias x bias, bias x pose, the bias half of g) is complete on every
  // rank without a collective, and the ranks only sum the surfel factors' pose corners.  (development option lm_one_collective: rounds 3 - 5's
  // form - the IMU triples sharded too, ONE all-reduce of {upper block pairs, g, cost} per linearisation.)
  const bool two = ctx->dev.lm_one_collective == 0;
  if (two)
    lo_i = 0, n_i = n_fac;
  else
    share(n_fac, lo_i, n_i);
  const int rc_local = window_build_impl(ctx, d_sld_surf, d_sld_pose, d_pairs_sld ? d_pairs_sld + lo_b : nullptr, n_b, d_fix_surf, d_fix_pose,
                                         d_pairs_fix ? d_pairs_fix + lo_u : nullptr, n_u, n_i ? h_imu + lo_i : nullptr, n_i ? n_i + 2 : 0,
                                         h_sample_times, ns_, h_grav, fix_first_pos, true);
  wc_dev_guard dg_(ctx);
  if (rc_local != WC_OK) {
    // A rank whose local build failed (bad argument, out of memory, too few sample states) still takes part in the share check,
    // with a poisoned share: the others then fail the check instead of waiting in the collective for a rank that has left
    // (ADVICE r3).  The message of the local failure is kept.
    const std::string why = ctx->err;
    const double poison[4] = {std::nan(""), std::nan(""), std::nan(""), std::nan("")};
    if (wc_ensure(ctx, ctx->b_status, 64 * 4) == WC_OK && hipMemcpyAsync(ctx->b_status.p, poison, sizeof(poison), hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
        hipStreamSynchronize(ctx->stream) == hipSuccess)
      (void)ctx->comm.allreduce_f64(ctx->comm.user, (double *)ctx->b_status.p, 4);
    ctx->err = why;
    return rc_local;
  }
  wc_window_state *W = ctx->win;
  // the whole problem's IMU factor count, from the replicated states (the selection rule of BuildImuResiduals, cc:324-329)
  uint64_t ni_all = 0;
  for (uint64_t i = 0; i + 2 < n_imu && h_imu; ++i) {
    if (h_imu[i].t < W->times.front()) continue;
    if (h_imu[i + 2].t > W->times.back()) break;
    ++ni_all;
  }
  W->two_coll = two;
  const double mine[4] = {(double)W->nb, (double)W->nu, (double)W->ni, 1.0};
  double *chk = (double *)W->mail.p + 48;
  WC_HIP(ctx, hipMemcpyAsync(chk, mine, sizeof(mine), hipMemcpyHostToDevice, ctx->stream));
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (`mine` is a stack array)
  WC_TRY(do_allreduce(ctx, W, chk, 4));
  double all[4];
  WC_HIP(ctx, hipMemcpyAsync(all, chk, sizeof(all), hipMemcpyDeviceToHost, ctx->stream));
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const double imu_expected = two ? (double)(w * ni_all) : (double)ni_all;  // (replicated: every rank holds all of them)
  if (all[0] != (double)n_pairs_sld || all[1] != (double)n_pairs_fix || all[2] != imu_expected || all[3] != (double)w) {
    W->built = false;
    return wc_fail(ctx, WC_ERR_ARG,
                   "wc_window_build_sharded: the ranks' shares do not add up to the problem this rank was given (binary %.0f of %llu, unary %.0f of "
                   "%llu, imu %.0f of %llu, ranks %.0f of %llu): every rank must pass the same replicated arguments",
                   all[0], (unsigned long long)n_pairs_sld, all[1], (unsigned long long)n_pairs_fix, all[2], (unsigned long long)ni_all, all[3],
                   (unsigned long long)w);
  }
  return WC_OK;
}

namespace {

// lin = [H (n*n) | g (np) | cost, spare]; `other` = the buffer that does NOT hold the current point's linearisation
inline double *lin_H(wc_window_state *W, bool other = false) { return (double *)(((W->lin_sel != 0) != other) ? W->lin_alt.p : W->lin.p); }
inline double *lin_g(wc_window_state *W, bool other = false) { return lin_H(W, other) + (size_t)W->n * W->n; }
inline double *lin_cost(wc_window_state *W, bool other = false) { return lin_g(W, other) + W->np; }

// ticket != 0: stored to host_mail[48] behind the mailbox (system-scope fence in between) - the host waits for it by reading
// pinned memory instead of waiting for the stream (wait_mail)
__global__ void __launch_bounds__(1024) k_post_reduce(const double *g, const double *cost, int n, double *mail, int slot, double *host_mail = nullptr,
                                                     unsigned long long ticket = 0) {
  __shared__ double s[1024];
  const int tid = threadIdx.x;
  double mx = 0.0;
  for (int i = tid; i < n; i += 1024) mx = fmax(mx, fabs(g[i]));
  s[tid] = mx;
  __syncthreads();
  for (int st = 512; st > 0; st >>= 1) {
    if (tid < st) s[tid] = fmax(s[tid], s[tid + st]);
    __syncthreads();
  }
  if (tid == 0) {
    mail[slot] = cost[0];
    mail[slot + 1] = s[0];
    if (host_mail) {  // the whole mailbox to pinned host memory (as k_sum_blocks does on the evaluation path)
      for (int i = 0; i < 40; ++i) host_mail[i] = (i == slot) ? cost[0] : (i == slot + 1 ? s[0] : mail[i]);
      if (ticket) {
        __threadfence_system();
        __hip_atomic_store((unsigned long long *)(host_mail + 48), ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// multi-GPU hook: sum a device buffer over all ranks (RCCL through the caller); no-op on one GPU
// (the callback installed with wc_window_set_allreduce, else the ctx's communicator - the in-library RCCL binding of comm.hip
// enqueues ncclAllReduce on the ctx stream: no host synchronisation on that path)
bool multi_gpu(const wc_ctx *ctx, const wc_window_state *W) {
  return W->allreduce != nullptr || (W->sharded && ctx->have_comm && ctx->comm.world > 1 && ctx->comm.allreduce_f64 != nullptr);
}
int do_allreduce(wc_ctx *ctx, wc_window_state *W, double *d_buf, size_t count) {
  if (W->allreduce) {
    WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (W->allreduce(W->allreduce_user, d_buf, (uint64_t)count) != 0) return wc_fail(ctx, WC_ERR_HIP, "all-reduce callback failed");
    return WC_OK;
  }
  if (!multi_gpu(ctx, W)) return WC_OK;
  if (!ctx->comm.stream_ordered) WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->comm.allreduce_f64(ctx->comm.user, d_buf, (uint64_t)count) != 0) return wc_fail(ctx, WC_ERR_HIP, "all-reduce failed");
  return WC_OK;
}

// the ctx stream waits for the side stream's last expand (two-collective form); a no-op when nothing is in flight there
int join_side(wc_ctx *ctx, wc_window_state *W) {
  if (!W->side_pending) return WC_OK;
  WC_HIP(ctx, hipStreamWaitEvent(ctx->stream, W->ev_side_done, 0));
  W->side_pending = false;
  return WC_OK;
}

// all kernels of one linearisation at x (device): partials -> H, g ; cost -> mail[slot], max|g| -> mail[slot+1]
int enqueue_linearize(wc_ctx *ctx, wc_window_state *W, const double *d_x, int mail_slot, bool post = true, bool other = false,
                      double *host_mail = nullptr, unsigned long long ticket = 0) {
  hipStream_t st = ctx->stream;
  WC_TRY(join_side(ctx, W));  // (the reduction buffer and the linearisation buffers are written again below)
  const Piece *pcs = (const Piece *)W->pieces.p;
  double *partial = (double *)W->partial.p;
  // (Round 3, tried: k_lin_imu - 14.5 us of dependent fp64 chains in a few hundred workgroups - on a stream of its own beside the
  // surfel families, fork / join by events: 1 650 LM it/s against 1 790 at C4, odometry-step solve 2.58 against 2.39 ms - the two
  // cross-stream waits cost more than the launch they hide.  One stream.)
  const bool imu_apart = ctx->dev.lin_imu_apart != 0;  // (A/B: the IMU family as a launch of its own)
  const bool unary_in = ctx->dev.lin_unary_apart == 0 && !W->unary_multi;  // (A/B: the unary family as a launch of its own; multi-chunk pieces: always)
  const bool fused = W->npiece_b && W->npiece_i && !imu_apart && kPiece == 256;
  const bool fused_u = fused && unary_in && W->npiece_u;
#ifndef WC_LIN_FEW
#define WC_LIN_FEW (6u * 256u)  // two rounds of the chip's slots at three per CU (facade, 244 linearisations: 13.4 us on average with OCC = 4 throughout, 12.2 with 768, 11.9 - 12.4 with 1 536 / 3 072; the step's 2 989 pieces: 26.8 us with OCC = 4, 29.1 with 3)
#endif
  const bool few = W->npiece_b + W->npiece_u <= WC_LIN_FEW;
  // (binary pieces of at most 128 records go two to a workgroup - lin_binary_pair_body; development option lin_pair = 0: one each)
  const uint32_t n_big = ctx->dev.lin_pair != 0 ? W->npiece_b_big : W->npiece_b;
  const uint32_t n_bwg = n_big + (W->npiece_b - n_big + 1u) / 2u;
  auto fused_launch = [&](auto kern, uint32_t grid) {
    kern<<<grid, 256, 0, st>>>(W->wp, pcs, W->npiece_i, W->npiece_b, W->npiece_u, (const double *)W->brec.p, W->nb, (const double *)W->urec.p, W->nu,
                               (const ImuRec *)W->irec.p, (const double *)W->times_d.p, d_x, partial, W->npart_doubles, n_big, n_bwg);
  };
  if (fused_u) {
    if (few)
      fused_launch(k_lin_fused<true, 3>, W->npiece_i + n_bwg + W->npiece_u);
    else
      fused_launch(k_lin_fused<true, WC_LIN_WG_PER_CU>, W->npiece_i + n_bwg + W->npiece_u);
  } else if (fused) {
    if (few)
      fused_launch(k_lin_fused<false, 3>, W->npiece_i + n_bwg);
    else
      fused_launch(k_lin_fused<false, WC_LIN_WG_PER_CU>, W->npiece_i + n_bwg);
  }
  else if (W->npiece_b)
    k_lin_surfel<24, false><<<W->npiece_b, kPiece, 0, st>>>(W->wp, pcs, (const double *)W->brec.p, (const uint32_t *)W->bkey.p, W->nb,
                                                          d_x, partial, W->npart_doubles);
  if (W->npiece_u && !fused_u) {
    if (W->unary_multi)
      k_lin_surfel<12, true, true><<<W->npiece_u, kPiece, 0, st>>>(W->wp, pcs + W->npiece_b, (const double *)W->urec.p,
                                                                 (const uint32_t *)W->ukey.p, W->nu, d_x, partial, W->npart_doubles + W->npiece_b);
    else
      k_lin_surfel<12, true><<<W->npiece_u, kPiece, 0, st>>>(W->wp, pcs + W->npiece_b, (const double *)W->urec.p,
                                                           (const uint32_t *)W->ukey.p, W->nu, d_x, partial, W->npart_doubles + W->npiece_b);
  }
  if (W->npiece_i && !fused)
    k_lin_imu<<<W->npiece_i, 256, 0, st>>>(W->wp, pcs + W->npiece_b + W->npiece_u, (const ImuRec *)W->irec.p, d_x,
                                          (const double *)W->times_d.p, partial, W->npart_doubles + W->npiece_b + W->npiece_u);
  GatherArgs ga;
  ga.src = (const Src *)W->src.p, ga.src_begin = (const uint32_t *)W->src_begin.p;
  ga.gsrc = (const GSrc *)W->gsrc.p, ga.gsrc_begin = (const uint32_t *)W->gsrc_begin.p;
  ga.pieces = pcs, ga.heavy = (const uint32_t *)W->heavy.p, ga.partial = partial;
  ga.near = (const uint32_t *)W->near_l.p, ga.far = (const FarJob *)W->far_l.p, ga.nnear = W->nnear, ga.nfar = W->nfar;
  // multi-GPU: the ranks reduce the upper block triangle only (pair order, 144 doubles per block pair, then g and the cost):
  // half the bytes of the dense matrix on the wire; one more kernel spreads the sum into both triangles
  const bool packed = multi_gpu(ctx, W);
  const bool two = packed && W->two_coll && !W->allreduce;  // (the two-collective form: IMU factors replicated by wc_window_build_sharded)
  double *red = nullptr;
  const size_t red_count = two ? (size_t)36 * W->npairs + 6 * (size_t)W->ns : (size_t)W->red_H + W->np + 2;
  if (packed) {
    WC_TRY(wc_ensure(ctx, W->reduce, red_count * 8));
    red = (double *)W->reduce.p;
  }
  ga.H = packed ? red : lin_H(W, other);
  ga.g = packed ? red + (size_t)W->red_H : lin_g(W, other);
  ga.pair_off = (const uint32_t *)W->pair_off.p;
  ga.cost = ga.g + W->np;
  ga.packed = packed ? 1 : 0;
  ga.nheavy = W->nheavy, ga.npairs = W->npairs, ga.npieces = W->npiece_b + W->npiece_u + W->npiece_i;
  ga.nb_pieces = W->npiece_b, ga.nu_pieces = W->npiece_u, ga.ns = W->ns, ga.fix_first = W->wp.fix_first;
  ga.cost_base = W->npart_doubles;
  // max |g| + the mailbox by the last of k_gather's g / cost workgroups (one GPU; behind an all-reduce k_post_reduce stays a launch)
  const bool post_apart = ctx->dev.lin_post_apart != 0;
  ga.post = (post && !packed && !post_apart) ? 1 : 0;
  ga.mail_slot = mail_slot, ga.done = (uint32_t *)((double *)W->mail.p + 60), ga.mail = (double *)W->mail.p, ga.host_mail = host_mail, ga.ticket = ticket;
  const uint32_t gather_grid = W->nheavy + (W->nnear + kGG * kLightSets - 1) / (kGG * kLightSets) + (W->nfar + kFarGroups * kFarSets - 1) / (kFarGroups * kFarSets) + W->ns + 1;
  ga.only = 0;
  if (two) {
    // Two gathers, two collectives: the surfel factors' sums into the compact buffer + their cost; the (replicated) IMU factors'
    // sums straight into H / g + their cost.  The 16-byte sum of the costs goes first and k_post_cost sends the linearisation's cost to the
    // host: the trust-region decision does not wait for the 0.6 - 2.3 MB of pose corners, which only k_schur_form needs.
    double *mail = (double *)W->mail.p;
    ga.only = 1, ga.packed = 1, ga.H = red, ga.g = red + (size_t)36 * W->npairs, ga.cost = mail + 52, ga.post = 0;
    k_gather<<<gather_grid, 144 * kGG, 0, st>>>(ga);
    ga.only = 2, ga.packed = 0, ga.H = lin_H(W, other), ga.g = lin_g(W, other), ga.cost = mail + 54;
    k_gather<<<gather_grid, 144 * kGG, 0, st>>>(ga);
    WC_HIP(ctx, hipGetLastError());
    WC_TRY(do_allreduce(ctx, W, mail + 52, 2));  // collective 1: {cost of the surfel factors, spare}
    k_post_cost<<<1, 64, 0, st>>>(mail, mail_slot, lin_cost(W, other), post ? host_mail : nullptr, post ? ticket : 0ull);
    double *late = nullptr;
    if (host_mail) late = host_mail + 40 + (W->lin_count & 1u);
    ++W->lin_count;
    // collective 2 (pose corners + the pose half of g) and the expansion: on the side stream when the communicator enqueues (the in-library
    // RCCL binding) - the ctx stream goes on with the bias elimination, which reads nothing of it, and waits in front of k_schur_form
    // (join_side).  Development option lm_side_stream: 0 = on the ctx stream, 2 = the side stream's choreography also with a communicator
    // of callbacks (tests: the callback then runs with both streams drained).
    const bool rccl_side = ctx->comm.stream_ordered && ctx->rccl && ctx->comm.user == ctx->rccl;
    const bool side = ctx->dev.lm_side_stream == 2 || (ctx->dev.lm_side_stream == 1 && rccl_side);
    if (side) {
      if (!W->side) {
        WC_HIP(ctx, hipStreamCreateWithFlags(&W->side, hipStreamNonBlocking));
        WC_HIP(ctx, hipEventCreateWithFlags(&W->ev_side_go, hipEventDisableTiming));
        WC_HIP(ctx, hipEventCreateWithFlags(&W->ev_side_done, hipEventDisableTiming));
      }
      WC_HIP(ctx, hipEventRecord(W->ev_side_go, st));
      WC_HIP(ctx, hipStreamWaitEvent(W->side, W->ev_side_go, 0));
      if (rccl_side) {
        if (wc_rccl_allreduce_on(ctx, red, (uint64_t)red_count, W->side) != 0) return wc_fail(ctx, WC_ERR_HIP, "all-reduce (side stream) failed");
      } else {
        WC_HIP(ctx, hipStreamSynchronize(W->side));
        WC_TRY(do_allreduce(ctx, W, red, red_count));
      }
      k_expand_corners<<<W->npairs + 1, 64, 0, W->side>>>(red, W->npairs, W->ns, lin_H(W, other), lin_g(W, other), mail, mail_slot + 1, late);
      WC_HIP(ctx, hipGetLastError());
      WC_HIP(ctx, hipEventRecord(W->ev_side_done, W->side));
      W->side_pending = true;
      return WC_OK;
    }
    WC_TRY(do_allreduce(ctx, W, red, red_count));
    k_expand_corners<<<W->npairs + 1, 64, 0, st>>>(red, W->npairs, W->ns, lin_H(W, other), lin_g(W, other), mail, mail_slot + 1, late);
    WC_HIP(ctx, hipGetLastError());
    return WC_OK;
  }
  k_gather<<<gather_grid, 144 * kGG, 0, st>>>(ga);
  WC_HIP(ctx, hipGetLastError());
  if (packed) {
    WC_TRY(do_allreduce(ctx, W, red, red_count));  // the ONE collective of a linearisation
    k_expand_pairs<<<W->npairs + 1, 144, 0, st>>>(red, W->npairs, W->ns, W->np, lin_H(W, other), lin_g(W, other), (const uint32_t *)W->pair_off.p, W->red_H);
  }
  // (inside the LM loop the next lm_step forms cost / max |g| of this linearisation itself: post = false)
  if (post && !ga.post) k_post_reduce<<<1, 1024, 0, st>>>(lin_g(W, other), lin_cost(W, other), W->n, (double *)W->mail.p, mail_slot, host_mail, ticket);
  WC_HIP(ctx, hipGetLastError());
  return WC_OK;
}

// The host's wait for an iteration's mailbox: k_post_reduce stores a ticket to pinned memory behind the mailbox, the host reads
// that word until it shows up - hipStreamSynchronize adds the end-of-kernel signal and the runtime's wake-up to every iteration
// (~33 us between k_post_reduce's end and the next k_pcr_init's start in a kernel trace of the odometry step).  Falls back to the
// stream wait after 50 ms (a faulted kernel never stores its ticket: the stream wait then reports the error).
int wait_mail(wc_ctx *ctx, unsigned long long ticket) {
  if (ticket) {
    const volatile unsigned long long *w = (const volatile unsigned long long *)(ctx->h_mail + 48);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0;; ++spin) {
      if (__atomic_load_n((const unsigned long long *)w, __ATOMIC_ACQUIRE) == ticket) return WC_OK;
      if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
    }
  }
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WC_OK;
}

// cost (and optionally residuals in the reference's block order: binary, unary, imu x 12) at x -> mail[slot]
int enqueue_evaluate(wc_ctx *ctx, wc_window_state *W, const double *d_x, double *d_res, int mail_slot, double *host_mail = nullptr) {
  hipStream_t st = ctx->stream;
  double *cp = (double *)W->cost_part.p;
  const uint32_t gb = (W->nb + 255) / 256, gu = (W->nu + 255) / 256, gi = (W->ni + 255) / 256;
  if (gb + gu) {
    const EvalArgs B{(const double *)W->brec.p, (const uint32_t *)W->bkey.p, (const uint32_t *)W->borig.p, W->nb, d_res};
    const EvalArgs U{(const double *)W->urec.p, (const uint32_t *)W->ukey.p, (const uint32_t *)W->uorig.p, W->nu,
                     d_res ? d_res + W->nb : nullptr};
    k_eval_surfel<<<gb + gu, 256, 0, st>>>(W->wp, B, U, gb, d_x, cp);
  }
  if (gi)
    k_eval_imu<<<gi, 256, 0, st>>>(W->wp, (const ImuRec *)W->irec.p, W->ni, d_x, (const double *)W->times_d.p,
                                  d_res ? d_res + W->nb + W->nu : nullptr, cp + gb + gu);
  if (multi_gpu(ctx, W) && W->two_coll && !W->allreduce) {  // (the IMU factors are on every rank: their cost is added behind the sum of the surfel factors')
    double *mail = (double *)W->mail.p;
    k_sum_blocks<<<1, 1024, 0, st>>>(cp, gb + gu, mail, 52, nullptr);
    k_sum_blocks<<<1, 1024, 0, st>>>(cp + gb + gu, gi, mail, 54, nullptr);
    WC_HIP(ctx, hipGetLastError());
    WC_TRY(do_allreduce(ctx, W, mail + 52, 2));
    k_post_cost<<<1, 64, 0, st>>>(mail, mail_slot, mail + 55, nullptr, 0ull);
    WC_HIP(ctx, hipGetLastError());
    return WC_OK;
  }
  k_sum_blocks<<<1, 1024, 0, st>>>(cp, gb + gu + gi, (double *)W->mail.p, mail_slot, multi_gpu(ctx, W) ? nullptr : host_mail);
  WC_HIP(ctx, hipGetLastError());
  WC_TRY(do_allreduce(ctx, W, (double *)W->mail.p + mail_slot, 1));
  return WC_OK;
}

int read_mail(wc_ctx *ctx, wc_window_state *W, int count) {
  WC_HIP(ctx, hipMemcpyAsync(ctx->h_mail, W->mail.p, (size_t)count * 8, hipMemcpyDeviceToHost, ctx->stream));
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WC_OK;
}

}  // namespace

extern "C" int wc_window_counts(wc_ctx *ctx, uint64_t counts[4]) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !ctx->win || !ctx->win->built) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  counts[0] = ctx->win->nb, counts[1] = ctx->win->nu, counts[2] = ctx->win->ni;
  counts[3] = ctx->win->npiece_b + ctx->win->npiece_u + ctx->win->npiece_i;
  return WC_OK;
}

extern "C" uint64_t wc_window_reduce_bytes(wc_ctx *ctx) {
  if (!ctx || !ctx->win || !ctx->win->built || !multi_gpu(ctx, ctx->win)) return 0;
  if (ctx->win->two_coll && !ctx->win->allreduce) return ((uint64_t)36 * ctx->win->npairs + 6 * (uint64_t)ctx->win->ns + 2) * 8;  // (+ the 16-byte cost collective)
  return ((uint64_t)ctx->win->red_H + ctx->win->np + 2) * 8;
}

extern "C" int wc_window_evaluate(wc_ctx *ctx, const double *h_x, double *h_cost, double *d_residuals) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !ctx->win || !ctx->win->built || !h_x || !h_cost) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  wc_window_state *W = ctx->win;
  WC_HIP(ctx, hipMemcpyAsync(W->x.p, h_x, (size_t)W->n * 8, hipMemcpyHostToDevice, ctx->stream));
  WC_TRY(enqueue_evaluate(ctx, W, (const double *)W->x.p, d_residuals, 0));
  WC_TRY(read_mail(ctx, W, 1));
  *h_cost = ctx->h_mail[0];
  return WC_OK;
}

extern "C" int wc_window_linearize(wc_ctx *ctx, const double *h_x, double *d_H, double *d_g, double *h_cost) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !ctx->win || !ctx->win->built || !h_x) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  wc_window_state *W = ctx->win;
  WC_HIP(ctx, hipMemcpyAsync(W->x.p, h_x, (size_t)W->n * 8, hipMemcpyHostToDevice, ctx->stream));
  WC_TRY(enqueue_linearize(ctx, W, (const double *)W->x.p, 0));
  WC_TRY(join_side(ctx, W));
  if (d_H) WC_HIP(ctx, hipMemcpyAsync(d_H, lin_H(W), (size_t)W->n * W->n * 8, hipMemcpyDeviceToDevice, ctx->stream));
  if (d_g) WC_HIP(ctx, hipMemcpyAsync(d_g, lin_g(W), (size_t)W->n * 8, hipMemcpyDeviceToDevice, ctx->stream));
  WC_TRY(read_mail(ctx, W, 2));
  if (h_cost) *h_cost = ctx->h_mail[0];
  return WC_OK;
}

// Measurement: `reps` linearisations at h_x enqueued back to back on the ctx stream between two HIP events - the device time of
// one linearisation (k_lin_fused + k_gather and the gap between them).  wc_window_linearize called in a loop also times its
// upload of x out of pageable memory, the wait for the mailbox and the caller's interpreter between two calls (~15 us of 125 at C4).
extern "C" int wc_window_linearize_timed(wc_ctx *ctx, const double *h_x, int reps, float *h_ms_per_linearisation) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !ctx->win || !ctx->win->built || !h_x || reps < 1 || !h_ms_per_linearisation)
    return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  wc_window_state *W = ctx->win;
  WC_HIP(ctx, hipMemcpyAsync(W->x.p, h_x, (size_t)W->n * 8, hipMemcpyHostToDevice, ctx->stream));
  WC_TRY(enqueue_linearize(ctx, W, (const double *)W->x.p, 0));
  WC_TRY(wc_timer_start(ctx));
  for (int r = 0; r < reps; ++r) WC_TRY(enqueue_linearize(ctx, W, (const double *)W->x.p, 0));
  WC_TRY(join_side(ctx, W));
  float ms = 0.f;
  WC_TRY(wc_timer_stop_ms(ctx, &ms));
  WC_TRY(read_mail(ctx, W, 2));
  *h_ms_per_linearisation = ms / (float)reps;
  return WC_OK;
}

// Multi-GPU: correspondences are sharded over ranks, the unknowns stay replicated.  The caller installs a callback that
// sums a device buffer of doubles over all ranks (RCCL all-reduce over xGMI via torch.distributed or rccl directly);
// it is invoked once per linearisation on the packed buffer {H, g, cost} and once per candidate-cost evaluation on one
// double.  Every rank then runs the identical, deterministic LM logic on identical numbers.
extern "C" int wc_window_set_allreduce(wc_ctx *ctx, int (*fn)(void *user, double *d_buf, uint64_t count), void *user) {
  wc_dev_guard dg_(ctx);
  if (!ctx) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  if (!ctx->win) {
    ctx->win = new wc_window_state;
    ctx->win->reduce.plain = ctx->win->mail.plain = true;
  }
  ctx->win->allreduce = fn;
  ctx->win->allreduce_user = user;
  return WC_OK;
}

// ceres::Solve with the reference's options (lidar_odometry.cc:551-561): Ceres-default trust-region LM restated
// (upstream semantics, see oracle/window.cc for the per-rule citations).
extern "C" int wc_window_solve(wc_ctx *ctx, double *h_x_inout, wc_solve_summary *summary, double *h_first_step) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !ctx->win || !ctx->win->built || !h_x_inout || !summary) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  wc_window_state *W = ctx->win;
  hipStream_t st = ctx->stream;
  const int n = W->n, np = W->np, ld = W->ld, nblk = np / kNB;
  std::memset(summary, 0, sizeof(*summary));
  double *x = (double *)W->x.p, *xc = (double *)W->xc.p, *H = lin_H(W), *g = lin_g(W);
  double *scale = (double *)W->scale.p, *diag = (double *)W->diag.p, *A = (double *)W->A.p, *y = (double *)W->y.p;
  double *Lmat = (double *)W->Lmat.p;
  double *mail = (double *)W->mail.p;
  int *fail = (int *)((double *)W->mail.p + 32);
  std::vector<double> best(h_x_inout, h_x_inout + n), cur(best);
  const bool lm_dense = ctx->dev.lm_dense != 0;
  const bool use_schur = !lm_dense && W->ns >= 4;  // (two super-blocks at least: the reduction has a level)
  if (use_schur) {
    const int ns_al = std::max(W->ns, 96), M_al = (ns_al + 1) / 2, ldr_al = ((6 * ns_al + 1 + 63) / 64) * 64;  // (as in wc_window_build: no re-allocation while a window grows)
    for (int b = 0; b < 2; ++b) {
      WC_TRY(wc_ensure(ctx, W->pcr_D[b], (size_t)M_al * 144 * 8));
      WC_TRY(wc_ensure(ctx, W->pcr_A[b], (size_t)M_al * 144 * 8));
      WC_TRY(wc_ensure(ctx, W->pcr_R[b], (size_t)M_al * kSB * ldr_al * 8));
    }
    WC_TRY(wc_ensure(ctx, W->yred, (size_t)(12 * ns_al + 2 * kNB + 128) * 8));  // (y of the pose half; behind it the fused tail's partial sums: four per workgroup)
  }

  const bool poll_mail = ctx->dev.lm_sync == 0;  // (development option lm_sync: wait for the stream instead of the ticket)
  double *h_mail_dev = nullptr, *h_stage_dev = nullptr;  // device addresses of the pinned mailbox and of its staging area
  {
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, ctx->h_mail, 0) != hipSuccess || !dp) return wc_fail(ctx, WC_ERR_HIP, "pinned mailbox is not device-mapped");
    h_mail_dev = (double *)dp;
    h_stage_dev = h_mail_dev + 64;
  }
  // the two-collective form of a sharded window: the cost of a linearisation reaches the host behind the 16-byte collective (ticket), its
  // max |g| behind the large one - in a late slot of the pinned mailbox the loop reads with the NEXT ticket (two_late)
  const bool two = multi_gpu(ctx, W) && W->two_coll && !W->allreduce;
  WC_HIP(ctx, hipMemcpyAsync(x, cur.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
  if (poll_mail && !multi_gpu(ctx, W)) {
    // the first linearisation's cost / max |g| through the pinned mailbox and its ticket, like every later one (the stream
    // wait of rounds 1 - 3 added the end-of-kernel signal and the runtime's wake-up, ~20 us per solve; k_scale_init runs behind it)
    const unsigned long long ticket = ++ctx->mail_ticket;
    WC_TRY(enqueue_linearize(ctx, W, x, 0, /*post=*/true, /*other=*/false, h_mail_dev, ticket));
    k_scale_init<<<(n + 255) / 256, 256, 0, st>>>(H, n, scale);
    WC_HIP(ctx, hipGetLastError());
    WC_TRY(wait_mail(ctx, ticket));
  } else {
    WC_TRY(enqueue_linearize(ctx, W, x, 0));
    WC_TRY(join_side(ctx, W));
    k_scale_init<<<(n + 255) / 256, 256, 0, st>>>(H, n, scale);
    WC_TRY(read_mail(ctx, W, 2));
  }
  summary->n_linearizations = 1;
  double cost = ctx->h_mail[0], gmax = ctx->h_mail[1];
  summary->initial_cost = cost;
  double min_cost = cost, radius = 1e4, decrease = 2.0, x_norm = 0.0;
  for (int i = 0; i < n; ++i) x_norm += cur[i] * cur[i];
  x_norm = std::sqrt(x_norm);
  int iter = 0, consecutive_invalid = 0;
  bool first_recorded = false;
  // ONE host round trip per iteration: the linearisation at an accepted point is enqueued without waiting for it; its
  // cost, max |g| and the accepted x are read back with the NEXT iteration's mailbox (the next step only needs the trust
  // region radius, which is known).  If that read-back shows a gradient below tolerance, the iteration that was enqueued
  // on top of it is discarded - the reference stops before it.
  bool lin_pending = false;
  bool two_late = false;  // (two-collective form) max |g| of the accepted point has not been looked at yet: it sits in h_mail[40 + two_at]
  int two_slot = 0, two_at = 0;
  // Round 3: the candidate's cost comes from a LINEARISATION at the candidate (into the other {H, g, cost} buffer) instead of a
  // cost-only pass over the same records: an accepted step - the rule - then needs no second pass (one pass over the factors
  // per iteration instead of two, -0.05 ms of 0.6 at C4), a rejected one has formed an H nobody uses (+0.1 ms).  Its cost and
  // max |g| arrive with the iteration's mailbox, so nothing is pending between iterations.  Development option lm_eval_pass: round 2's flow.
  const bool cand_lin = ctx->dev.lm_eval_pass == 0;
  const bool spec_ok = ctx->dev.pcr_ahead != 0;  // (development option pcr_ahead: 0 = level 0 formed at the start of every iteration, as in rounds 3 - 4)
  const void *pcr_ready_for = nullptr;           // the H whose undamped level-0 blocks sit in pcr_D / A / R [0]
  auto resolve_pending = [&]() {
    cost = ctx->h_mail[0];
    gmax = ctx->h_mail[1];
    if (cost < min_cost) {
      min_cost = cost;
      best = cur;
    }
    lin_pending = false;
  };
  summary->termination = 1;
  if (gmax <= 1e-10) {
    summary->termination = 0;
  } else {
    while (true) {
      if (iter >= ctx->P.max_iterations) {
        summary->termination = 1;
        break;
      }
      if ((!lin_pending && !two_late && gmax <= 1e-10) || radius <= 1e-32) {
        summary->termination = 0;
        break;
      }
      ++iter;
      // One attempt of the iteration: the damped step (bias elimination + dense factor of the pose half, or - schur = false - round 2's
      // dense factor of all unknowns), the candidate, its linearisation, the mailbox.
      auto attempt = [&](bool schur) -> int {
        // LevenbergMarquardtStrategy::ComputeStep on the device
        if (schur) {
          // bias unknowns first, by parallel cyclic reduction (window_schur.inc); the dense panel steps run on the pose half
          const int ns = W->ns, npz = 6 * ns, M = (ns + 1) / 2;
          const int ldr = ((npz + 1 + 63) / 64) * 64, nch = (ldr + 255) / 256;
          const int np2 = ((npz + 1 + kNB - 1) / kNB) * kNB, ld2 = np2, nblk2 = np2 / kNB;
          double *Dp[2] = {(double *)W->pcr_D[0].p, (double *)W->pcr_D[1].p}, *Ap[2] = {(double *)W->pcr_A[0].p, (double *)W->pcr_A[1].p};
          double *Rp[2] = {(double *)W->pcr_R[0].p, (double *)W->pcr_R[1].p}, *yred = (double *)W->yred.p;
          int cur = 0, nlev = 0;
          for (int s = 1; s < M; s *= 2) ++nlev;
          double *X = nullptr;
          // level 0 from H, UNDAMPED (radius 0) - unless it is there already: enqueued behind the linearisation this H came from,
          // before the host knew that its step would be accepted (pcr_ahead below)
          if (pcr_ready_for != (const void *)H || !spec_ok) {
            const PcrSrc src{H, g, scale, n, ns, 0.0, diag};
            k_pcr_init<<<dim3(M, nch), 256, 0, st>>>(src, Dp[0], Ap[0], Rp[0], ldr, fail);
          }
          pcr_ready_for = nullptr;  // (the levels overwrite level 0)
          const PcrDamp damp{radius, ns, diag}, nodamp{0.0, ns, diag};
          for (int s = 1, lev = 0; lev < nlev; s *= 2, ++lev) {  // (ns >= 4: at least one level)
            const bool last = lev == nlev - 1, first = lev == 0;
            // (the levels of small stride work on banded right-hand sides: fewer column chunks per row - k_pcr_level; development option
            // pcr_full_width: every level over all columns, as in rounds 3 - 5)
            const int band_chunks = (kSB * (4 * s + 1) + 255) / 256;
            const int limited = (ctx->dev.pcr_full_width == 0 && band_chunks < nch) ? band_chunks : 0;
            const dim3 grid(M * (limited ? limited : nch));
            if (last && first)
              k_pcr_level<true, true><<<grid, 256, 0, st>>>(s, M, Dp[cur], Ap[cur], Rp[cur], Dp[cur ^ 1], Ap[cur ^ 1], Rp[cur ^ 1], ldr, fail, damp, npz, limited);
            else if (last)
              k_pcr_level<true, false><<<grid, 256, 0, st>>>(s, M, Dp[cur], Ap[cur], Rp[cur], Dp[cur ^ 1], Ap[cur ^ 1], Rp[cur ^ 1], ldr, fail, nodamp, npz, limited);
            else if (first)
              k_pcr_level<false, true><<<grid, 256, 0, st>>>(s, M, Dp[cur], Ap[cur], Rp[cur], Dp[cur ^ 1], Ap[cur ^ 1], Rp[cur ^ 1], ldr, fail, damp, npz, limited);
            else
              k_pcr_level<false, false><<<grid, 256, 0, st>>>(s, M, Dp[cur], Ap[cur], Rp[cur], Dp[cur ^ 1], Ap[cur ^ 1], Rp[cur ^ 1], ldr, fail, nodamp, npz, limited);
            cur ^= 1;
          }
          X = Rp[cur];  // the last level wrote X = T^-1 [C | bB] where the others write R'
          WC_TRY(join_side(ctx, W));  // (the pose blocks and the pose half of g: summed over the ranks on the side stream)
          k_schur_form<<<((np2 + 255) / 256) * (1 + ns + (np2 - npz)), 256, 0, st>>>(H, g, scale, X, n, ns, ldr, np2, ld2, radius, A, diag, Lmat, (double *)W->Linv.p, fail, (np2 + 255) / 256);
          // (development option lm_back_chunks: rounds 2 - 5's back substitution - chunk solves + products - and k_schur_bias_y / k_lm_step
          // as launches of their own, for A/B runs; default: identity rows appended to the factorisation, k_back_mul, one fused tail)
          const bool back_mul = ctx->dev.lm_back_chunks == 0;
#ifdef WC_DEV_KNOBS
          if (ctx->dev.dbg_lm & 8) WC_TRY(wc_ensure(ctx, W->reduce, (size_t)nblk2 * 1024 * 8));
#endif
          for (int k = 0; k + 1 < nblk2; ++k) {
            const int tiles = ((nblk2 - k - 1) * kNB + 63) / 64;
            const int n_tri = 1 + tiles * (tiles + 1) / 2;
            const int n_extra = back_mul ? (((k + 1) * kNB + 63) / 64) * tiles : 0;
            k_chol_step<<<n_tri + n_extra, 256, 0, st>>>(A, ld2, k, nblk2, Lmat, (double *)W->Linv.p, fail, npz, n_tri, tiles, ctx->dev.dbg_lm,
                                                         (long long *)W->reduce.p);
          }
#ifdef WC_DEV_KNOBS
          if (ctx->dev.dbg_lm & 8) {  // (development session: wall-clock stamps of every workgroup of every step, 100 MHz)
            static int calls = 0;
            if (++calls == 3) {
              std::vector<long long> hb((size_t)nblk2 * 1024);
              (void)hipStreamSynchronize(st);
              (void)hipMemcpy(hb.data(), W->reduce.p, hb.size() * 8, hipMemcpyDeviceToHost);
              for (int k = 0; k + 1 < nblk2; ++k) {
                const int tiles = ((nblk2 - k - 1) * kNB + 63) / 64, n_tri = 1 + tiles * (tiles + 1) / 2;
                const int n_extra = back_mul ? (((k + 1) * kNB + 63) / 64) * tiles : 0;
                long long t0 = hb[(size_t)k * 1024], le = hb[(size_t)k * 1024 + 1], ts = 0, te = 0, xs = 0, xe = 0, tl = 0, xl = 0;
                for (int b = 1; b < n_tri + n_extra; ++b) {
                  const long long a = hb[((size_t)k * 512 + b) * 2], e = hb[((size_t)k * 512 + b) * 2 + 1];
                  if (b < n_tri) ts = std::max(ts, a - t0), te = std::max(te, e - t0), tl = std::max(tl, e - a);
                  else xs = std::max(xs, a - t0), xe = std::max(xe, e - t0), xl = std::max(xl, e - a);
                }
                fprintf(stderr, "step %2d: lead %5.2f us | tiles: last start %5.2f longest %5.2f last end %5.2f | extra: last start %5.2f longest %5.2f last end %5.2f\n", k,
                        (le - t0) * 0.01, ts * 0.01, tl * 0.01, te * 0.01, xs * 0.01, xl * 0.01, xe * 0.01);
              }
            }
          }
#endif
          const StepArgs sa{x, scale, g, diag, lin_cost(W), xc, mail, h_stage_dev, 0};
          if (back_mul) {
            k_back_mul<<<(npz + 7) / 8, 256, 0, st>>>(A, Lmat, ld2, nblk2, npz, (const double *)W->Linv.p, yred);
            k_schur_bias_y_step<<<(npz + 3) / 4, 256, 0, st>>>(X, yred, ns, ldr, y, sa, n, (uint32_t *)(mail + 62), yred + npz + kNB);
          } else {
            const double *zsrc = Lmat + (size_t)npz * ld2;
            for (int hi = (npz + kNB - 1) / kNB; hi > 0;) {
              const int lo = std::max(0, hi - kBackChunk);
              k_chol_back_chunk<<<1, 1024, 0, st>>>(Lmat, ld2, npz, (const double *)W->Linv.p, zsrc, yred, lo, hi, sa);
              if (lo > 0) k_chol_back_gemv<<<lo, 1024, 0, st>>>(Lmat, ld2, npz, zsrc, yred, lo, hi);
              zsrc = yred;
              hi = lo;
            }
            k_schur_bias_y<<<(npz + 3) / 4, 256, 0, st>>>(X, yred, ns, ldr, y);
            k_lm_step<<<1, 1024, 0, st>>>(sa, y, n);
          }
        } else {
          {
            dim3 grid((np + 255) / 256, np);
            grid.y += 1;  // (+ the workgroup of the first diagonal block)
            WC_TRY(join_side(ctx, W));
            k_damp_first<<<grid, 256, 0, st>>>(H, g, scale, n, np, ld, radius, A, diag, Lmat, (double *)W->Linv.p, fail);
          }
          for (int k = 0; k + 1 < nblk; ++k) {
            const int tiles = ((nblk - k - 1) * kNB + 63) / 64;
            k_chol_step<<<1 + tiles * (tiles + 1) / 2, 256, 0, st>>>(A, ld, k, nblk, Lmat, (double *)W->Linv.p, fail, n, 1 + tiles * (tiles + 1) / 2, tiles);
          }
          {  // back substitution, chunk by chunk from the last block row
            const double *zsrc = Lmat + (size_t)n * ld;
            for (int hi = (n + kNB - 1) / kNB; hi > 0;) {
              const int lo = std::max(0, hi - kBackChunk);
              const StepArgs sa{x, scale, g, diag, lin_cost(W), xc, mail, h_stage_dev, lo == 0 ? 1 : 0};
              k_chol_back_chunk<<<1, 1024, 0, st>>>(Lmat, ld, n, (const double *)W->Linv.p, zsrc, y, lo, hi, sa);
              if (lo > 0) k_chol_back_gemv<<<lo, 1024, 0, st>>>(Lmat, ld, n, zsrc, y, lo, hi);
              zsrc = y;
              hi = lo;
            }
          }
        }
        unsigned long long ticket = 0;
        if (cand_lin && poll_mail && (!multi_gpu(ctx, W) || two) && h_mail_dev) ticket = ++ctx->mail_ticket;
        if (cand_lin && two) two_slot = (int)(W->lin_count & 1u);  // (where this linearisation's max |g| will land)
        if (cand_lin)
          WC_TRY(enqueue_linearize(ctx, W, xc, 5, /*post=*/true, /*other=*/true, (multi_gpu(ctx, W) && !two) ? nullptr : h_mail_dev, ticket));  // mail[5] = cost, [6] = max |g| at the candidate
        else
          WC_TRY(enqueue_evaluate(ctx, W, xc, nullptr, 5, h_mail_dev));
        WC_HIP(ctx, hipGetLastError());
        // (one GPU: k_sum_blocks has stored the mailbox to pinned host memory itself; with an all-reduce behind it, copy)
        if ((multi_gpu(ctx, W) && !(two && cand_lin)) || !h_mail_dev) WC_HIP(ctx, hipMemcpyAsync(ctx->h_mail, mail, 40 * 8, hipMemcpyDeviceToHost, st));
        if (cand_lin && use_schur && spec_ok) {
          // pcr_ahead: the next iteration's level 0 from the CANDIDATE's H, while the host waits for this iteration's mailbox and decides
          // (the blocks do not depend on the radius).  An accepted step - the rule - finds them there; a rejected one forms its own.
          // (Behind the mailbox's copy: k_pcr_init clears the factorisation's fail flag, which that copy carries.)
          const int ns = W->ns, npz = 6 * ns, M = (ns + 1) / 2, ldr = ((npz + 1 + 63) / 64) * 64, nch = (ldr + 255) / 256;
          const PcrSrc src{lin_H(W, true), lin_g(W, true), scale, n, ns, 0.0, diag};
          k_pcr_init<<<dim3(M, nch), 256, 0, st>>>(src, (double *)W->pcr_D[0].p, (double *)W->pcr_A[0].p, (double *)W->pcr_R[0].p, ldr, fail);
          pcr_ready_for = (const void *)lin_H(W, true);
        }
        WC_TRY(wait_mail(ctx, ticket));
        return WC_OK;
      };
      // Above a radius of 1e10 - a dozen very successful steps in a row - the damping has all but left the bias block T, whose own
      // conditioning (random-walk factors tying neighbouring biases, weak absolute information) then shows: the cyclic reduction's explicit
      // 12 x 12 inverses lose digits a direct factorisation keeps.  Seen in profiles/stress_facade.py on sparse streams (a sweep of 38
      // iterations for the oracle's 35, states 2.5e-4 apart; with the dense step from 1e10 on: 35 iterations, 1e-5).  Such iterations
      // take round 2's dense step (development option lm_dense_radius: the exponent, 0 = never).
      const bool schur_now = use_schur && !(ctx->dev.lm_dense_radius > 0 && radius > std::pow(10.0, (double)ctx->dev.lm_dense_radius));
      WC_TRY(attempt(schur_now));
      if (two_late) {  // max |g| at the point this iteration started from has arrived with the iteration's ticket (its kernels ran behind that expand)
        gmax = ctx->h_mail[40 + two_at];
        two_late = false;
        if (gmax <= 1e-10) {  // GradientToleranceReached there: the reference stops before this iteration
          --iter;
          summary->termination = 0;
          break;
        }
      }
      if (lin_pending) {
        resolve_pending();
        if (gmax <= 1e-10) {  // GradientToleranceReached at the point this iteration started from
          --iter;
          summary->termination = 0;
          break;
        }
      }
      summary->n_cost_evaluations++;
      int hfail;
      std::memcpy(&hfail, &ctx->h_mail[32], 4);
      double model_change = ctx->h_mail[2], step_norm = ctx->h_mail[3], cand_cost = ctx->h_mail[5];
      // A step the trust region is about to REJECT (or an invalid one) that came from the bias elimination is formed again by the dense
      // factorisation of all unknowns before the decision is taken (round 5).  The elimination computes S = P - C^T T^-1 C through a
      // parallel cyclic reduction without pivoting; on ill-conditioned windows (a free gauge held by the IMU factors alone, a large
      // radius, dozens of iterations) its step can be worse than a direct factorisation's - profiles/stress_window.py found solves that
      // rejected steps the oracle (and round 2's dense path) accepted and then needed 38 iterations for the oracle's 27, or hit the
      // iteration limit.  Accepted steps - the rule: all of them in the bench's windows, the odometry step and the facade's stream - cost
      // nothing extra; a genuine rejection costs one dense step.
      if (schur_now && cand_lin) {
        const bool invalid = hfail || !(model_change > 0) || !std::isfinite(step_norm);
        const double dc = cost - cand_cost;
        const bool rejected = !invalid && step_norm > 1e-8 * (x_norm + 1e-8) && std::fabs(dc) > 1e-6 * cost && !(dc / model_change > 1e-3);
        if (invalid || rejected) {
          WC_TRY(attempt(false));
          summary->n_cost_evaluations++;
          summary->first_step[1] += 1.0;  // (dense re-tries of this solve)
          std::memcpy(&hfail, &ctx->h_mail[32], 4);
          model_change = ctx->h_mail[2], step_norm = ctx->h_mail[3], cand_cost = ctx->h_mail[5];
        }
      }
      if (hfail || !(model_change > 0) || !std::isfinite(step_norm)) {  // HandleInvalidStep
        if (++consecutive_invalid >= 5) {
          summary->termination = 2;
          break;
        }
        radius *= 0.5;
        summary->unsuccessful_steps++;
        continue;
      }
      consecutive_invalid = 0;
      if (!first_recorded) {
        first_recorded = true;
        summary->first_step[0] = step_norm;
        if (h_first_step)
          for (int i = 0; i < n; ++i) h_first_step[i] = ctx->h_mail[64 + i] - cur[i];  // (the candidate lm_step staged)
      }
      if (step_norm <= 1e-8 * (x_norm + 1e-8)) {  // ParameterToleranceReached
        summary->termination = 0;
        break;
      }
      const double cost_change = cost - cand_cost;
      if (std::fabs(cost_change) <= 1e-6 * cost) {  // FunctionToleranceReached
        summary->termination = 0;
        break;
      }
      const double rho = cost_change / model_change;
      if (rho > 1e-3) {  // HandleSuccessfulStep
        std::swap(W->x, W->xc);
        x = (double *)W->x.p, xc = (double *)W->xc.p;
        // the accepted point = the candidate lm_step staged in pinned memory (|x| and the best point are host state)
        std::memcpy(cur.data(), ctx->h_mail + 64, (size_t)n * 8);
        x_norm = 0.0;
        for (int i = 0; i < n; ++i) x_norm += cur[i] * cur[i];
        x_norm = std::sqrt(x_norm);
        if (cand_lin) {  // the candidate's linearisation IS the new point's
          W->lin_sel ^= 1;
          H = lin_H(W), g = lin_g(W);
          cost = cand_cost, gmax = ctx->h_mail[6];
          if (two) gmax = 1.0, two_late = true, two_at = two_slot;  // (not there yet: looked at behind the next iteration's ticket)
          if (cost < min_cost) {
            min_cost = cost;
            best = cur;
          }
        } else {
          WC_TRY(enqueue_linearize(ctx, W, x, 0, /*post=*/false));
          lin_pending = true;
        }
        summary->n_linearizations++;
        radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3));
        radius = std::min(1e16, radius);
        decrease = 2.0;
        summary->successful_steps++;
      } else {  // HandleUnsuccessfulStep
        radius = radius / decrease;
        decrease *= 2;
        summary->unsuccessful_steps++;
      }
    }
  }
  if (lin_pending) {  // the last accepted point's linearisation is still in flight (and no lm_step follows it)
    k_post_reduce<<<1, 1024, 0, st>>>(lin_g(W), lin_cost(W), W->n, mail, 0);
    WC_HIP(ctx, hipMemcpyAsync(ctx->h_mail, mail, 2 * 8, hipMemcpyDeviceToHost, st));
    WC_HIP(ctx, hipStreamSynchronize(st));
    resolve_pending();
  }
  if (W->side_pending) {  // (the last linearisation's expansion may still run on the side stream: nothing of this solve outlives the call)
    WC_TRY(join_side(ctx, W));
    WC_HIP(ctx, hipStreamSynchronize(st));
  }
  summary->iterations = iter;
  summary->final_cost = min_cost;
  std::memcpy(h_x_inout, best.data(), (size_t)n * 8);
  return WC_OK;
}

int wc_touch_window() {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, (const void *)k_pair_keys) == hipSuccess ? WC_OK : WC_ERR_HIP;
}

// ---- self-test / micro-benchmark of the diagonal-block factor (tests/test_kat_gpu.py, profiles/dev/factor32.py) -------------------------
namespace {
__global__ void __launch_bounds__(256) k_selftest_factor(const double *Ain, int variant, int reps, double *Lout, double *Xout, long long *clk) {
  __shared__ double sB[kNB][kNB + 1];
  __shared__ double sXi[kNB][kNB + 1];
  long long best = 0x7fffffffffffffffll;
  bool ok = true;
  for (int r = 0; r < reps; ++r) {
    for (int e = threadIdx.x; e < kNB * kNB; e += 256) sB[e / kNB][e % kNB] = (e % kNB <= e / kNB) ? Ain[e] : 0.0;
    __syncthreads();
    const long long t0 = clock64();
    ok = variant == 1 ? factor_inv32_roles(sB, sXi) : factor_inv32_blk(sB, sXi);
    __syncthreads();
    const long long t1 = clock64();
    best = t1 - t0 < best ? t1 - t0 : best;
  }
  for (int e = threadIdx.x; e < kNB * kNB; e += 256) Lout[e] = sB[e / kNB][e % kNB], Xout[e] = sXi[e / kNB][e % kNB];
  if (threadIdx.x == 0) clk[0] = best, clk[1] = ok ? 1 : 0;
}
}  // namespace

// h_A: 32 x 32 row-major SPD (lower part read); h_L, h_X: L and L^-1 (lower); h_clk[0]: shader clocks of the fastest of `reps` runs,
// h_clk[1]: 1 = all pivots positive.  (variant: 0 = factor_inv32_blk, the only form in the library; kept as an argument for experiments)
extern "C" int wc_selftest_factor32(wc_ctx *ctx, int variant, int reps, const double *h_A, double *h_L, double *h_X, long long *h_clk) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !h_A || !h_L || !h_X || !h_clk || reps < 1) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  WC_TRY(wc_ensure(ctx, ctx->b_route[0], (size_t)(3 * kNB * kNB + 2) * 8));
  double *d = (double *)ctx->b_route[0].p;
  WC_HIP(ctx, hipMemcpyAsync(d, h_A, (size_t)kNB * kNB * 8, hipMemcpyHostToDevice, ctx->stream));
  k_selftest_factor<<<1, 256, 0, ctx->stream>>>(d, variant, reps, d + kNB * kNB, d + 2 * kNB * kNB, (long long *)(d + 3 * kNB * kNB));
  WC_HIP(ctx, hipGetLastError());
  WC_HIP(ctx, hipMemcpyAsync(h_L, d + kNB * kNB, (size_t)kNB * kNB * 8, hipMemcpyDeviceToHost, ctx->stream));
  WC_HIP(ctx, hipMemcpyAsync(h_X, d + 2 * kNB * kNB, (size_t)kNB * kNB * 8, hipMemcpyDeviceToHost, ctx->stream));
  WC_HIP(ctx, hipMemcpyAsync(h_clk, d + 3 * kNB * kNB, 16, hipMemcpyDeviceToHost, ctx->stream));
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WC_OK;
}
