// selftest.hip — known-answer hooks for the PRODUCT math header (dmath.h), host and device instantiation.
//
// The reference pins its SO(3) helpers with src/common/utils_test.cc:5-21 (Jl_inv(v) Jl(v) = I and Jl(v) = Jr(-v) at
// v = (1,2,3)).  Those tests must hold for the code the kernels and the facade actually run, not only for the oracle's
// copy of the formulas: wc_selftest_so3 evaluates every helper of dmath.h either in a one-thread kernel on the ctx's
// device (on_device != 0) or with the host instantiation of the same header (on_device == 0; ctx may be NULL, so the
// CPU test-suite can run it without a GPU).
#include "ctx.h"
#include "dmath.h"
#include "so3_fused.h"

using namespace wc;

namespace {

struct So3Out {  // 48 doubles
  double exp_q[4];     // so3_exp(v), (w, x, y, z)
  double log_exp[3];   // so3_log(so3_exp(v))
  double jl[9], jl_inv[9], jr[9], jr_inv[9];  // row-major
  double hat[9];
};
static_assert(sizeof(So3Out) == 52 * 8, "layout");

WC_HD void store9(const M3 &m, double *o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o[3 * i + j] = m.m[i][j];
}

WC_HD void so3_all(const double *v3, So3Out *o) {
  const V3 v = mk3(v3[0], v3[1], v3[2]);
  const Q4 q = so3_exp(v);
  o->exp_q[0] = q.w, o->exp_q[1] = q.x, o->exp_q[2] = q.y, o->exp_q[3] = q.z;
  const V3 l = so3_log(q);
  o->log_exp[0] = l.x, o->log_exp[1] = l.y, o->log_exp[2] = l.z;
  store9(so3_Jl(v), o->jl);
  store9(so3_Jl_inv(v), o->jl_inv);
  store9(so3_Jr(v), o->jr);
  store9(so3_Jr_inv(v), o->jr_inv);
  store9(hat(v), o->hat);
}

__global__ void k_selftest_so3(const double *v3, So3Out *o) { so3_all(v3, o); }

struct EigOut {
  double ev[3];
  double vec[9];  // columns = eigenvectors, row-major storage
};

WC_HD void eig_all(const double *a9, EigOut *o) {
  M3 a, v;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a.m[i][j] = a9[3 * i + j];
  eig3_sym(a, o->ev, v);
  store9(v, o->vec);
}

__global__ void k_selftest_eig3(const double *a9, EigOut *o) { eig_all(a9, o); }

struct QuatOut {  // 11 doubles
  double slerp[4];  // qslerp(a, f, b)
  double rot[3];    // qrot(a, p)
  double mul[4];    // qmul(a, b)
};

WC_HD void quat_all(const double *in12, QuatOut *o) {  // in = a(4), b(4), f, p(3)
  const Q4 a{in12[0], in12[1], in12[2], in12[3]}, b{in12[4], in12[5], in12[6], in12[7]};
  const Q4 s = qslerp(a, in12[8], b);
  o->slerp[0] = s.w, o->slerp[1] = s.x, o->slerp[2] = s.y, o->slerp[3] = s.z;
  const V3 r = qrot(a, mk3(in12[9], in12[10], in12[11]));
  o->rot[0] = r.x, o->rot[1] = r.y, o->rot[2] = r.z;
  const Q4 m = qmul(a, b);
  o->mul[0] = m.w, o->mul[1] = m.x, o->mul[2] = m.y, o->mul[3] = m.z;
}

__global__ void k_selftest_quat(const double *in12, QuatOut *o) { quat_all(in12, o); }

struct FusedOut {  // 25 doubles: the fused device forms of so3_fused.h (what the IMU / surfel factor kernels evaluate)
  double exp_q[4], jr[9];          // exp_jr(v)
  double log_exp[3], jr_inv[9];    // log_jr_inv(exp_jr(v).E)
};
__global__ void k_selftest_fused(const double *v3, FusedOut *o) {
  const ExpJr X = exp_jr(mk3(v3[0], v3[1], v3[2]));
  o->exp_q[0] = X.E.w, o->exp_q[1] = X.E.x, o->exp_q[2] = X.E.y, o->exp_q[3] = X.E.z;
  store9(X.Jr, o->jr);
  M3 Jri;
  const V3 l = log_jr_inv(X.E, &Jri);
  o->log_exp[0] = l.x, o->log_exp[1] = l.y, o->log_exp[2] = l.z;
  store9(Jri, o->jr_inv);
}

template <typename Out, typename Kern, typename HostFn>
int run_selftest(wc_ctx *ctx, const double *h_in, size_t n_in, int on_device, double *h_out, Kern kern, HostFn host_fn) {
  if (!h_in || !h_out) return wc_fail(ctx, WC_ERR_ARG, "wc_selftest: null argument");
  Out o;
  std::memset(&o, 0, sizeof(o));
  if (!on_device) {
    host_fn(h_in, &o);
  } else {
    if (!ctx) return WC_ERR_ARG;
    wc_dev_guard dg_(ctx);
    WC_TRY(wc_ensure(ctx, ctx->b_status, 1024));
    double *d_in = (double *)ctx->b_status.p;
    Out *d_out = (Out *)((char *)ctx->b_status.p + 128);
    WC_HIP(ctx, hipMemcpyAsync(d_in, h_in, n_in * 8, hipMemcpyHostToDevice, ctx->stream));
    kern<<<1, 1, 0, ctx->stream>>>(d_in, d_out);
    WC_HIP(ctx, hipGetLastError());
    WC_HIP(ctx, hipMemcpyAsync(&o, d_out, sizeof(o), hipMemcpyDeviceToHost, ctx->stream));
    WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  std::memcpy(h_out, &o, sizeof(o));
  return WC_OK;
}

}  // namespace

extern "C" int wc_selftest_so3(wc_ctx *ctx, const double v[3], int on_device, double out52[52]) {
  return run_selftest<So3Out>(ctx, v, 3, on_device, out52, k_selftest_so3, so3_all);
}

extern "C" int wc_selftest_so3_fused(wc_ctx *ctx, const double v[3], double out25[25]) {
  if (!ctx) return WC_ERR_ARG;
  return run_selftest<FusedOut>(ctx, v, 3, 1, out25, k_selftest_fused, [](const double *, FusedOut *) {});
}

extern "C" int wc_selftest_eig3(wc_ctx *ctx, const double a9[9], int on_device, double out12[12]) {
  return run_selftest<EigOut>(ctx, a9, 9, on_device, out12, k_selftest_eig3, eig_all);
}

extern "C" int wc_selftest_quat(wc_ctx *ctx, const double in12[12], int on_device, double out11[11]) {
  return run_selftest<QuatOut>(ctx, in12, 12, on_device, out11, k_selftest_quat, quat_all);
}
