// sweep.hip — the per-point stages in front of the hot path, same bandwidth-bound shape as
// extraction (48 B in / 48 B out per point, the IMU table is L2 resident):
//   point pre-filter  lidar_odometry.cc:489-496 : lidar->imu extrinsic (double math, cast to float), range and blind-box
//                     test, ORDER-PRESERVING compaction of the survivors
//   UndistortSweep    lidar_odometry.cc:143-158 : lower_bound over the IMU states, lerp + slerp, transform to the world
//                     frame, cast to float
// Both work on the reference's 48-byte hilti_ros::Point record in place (src/common/common.h:12-28).
#include <hip/hip_runtime.h>

#include <cstring>
#include <rocprim/rocprim.hpp>

#include "ctx.h"
#include "dmath.h"

using namespace wc;

namespace {

struct Pt48 {  // 48-byte record moved as three 16-byte pieces
  uint4 a, b, c;
};
static_assert(sizeof(Pt48) == 48, "record size");

struct FilterParams {
  double q[4], t[3], min_range, max_range, bmin[3], bmax[3];
};

__global__ void __launch_bounds__(256) k_prefilter_flags(const Pt48 *in, uint64_t n, FilterParams F, float *xyz_out, uint32_t *flags) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *f = (const float *)(in + i);
  const V3 p = qrot(Q4{F.q[0], F.q[1], F.q[2], F.q[3]}, mk3((double)f[0], (double)f[1], (double)f[2])) + mk3(F.t[0], F.t[1], F.t[2]);
  const float x = (float)p.x, y = (float)p.y, z = (float)p.z;
  const float nrm = sqrtf(x * x + y * y + z * z);
  const bool blind = (double)x >= F.bmin[0] && (double)x <= F.bmax[0] && (double)y >= F.bmin[1] && (double)y <= F.bmax[1] &&
                     (double)z >= F.bmin[2] && (double)z <= F.bmax[2];
  const bool keep = !((double)nrm < F.min_range || (double)nrm > F.max_range || blind);
  flags[i] = keep ? 1u : 0u;
  xyz_out[3 * i + 0] = x, xyz_out[3 * i + 1] = y, xyz_out[3 * i + 2] = z;
}

__global__ void __launch_bounds__(256) k_prefilter_scatter(const Pt48 *in, uint64_t n, const float *xyz, const uint32_t *flags,
                                                          const uint32_t *offsets, Pt48 *out, uint64_t cap, uint32_t *status,
                                                          double *kept_times) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (i == n - 1) status[0] = offsets[i] + flags[i];
  if (!flags[i]) return;
  const uint32_t o = offsets[i];
  if (o >= cap) return;
  Pt48 r = in[i];
  float *f = (float *)&r;
  f[0] = xyz[3 * i + 0], f[1] = xyz[3 * i + 1], f[2] = xyz[3 * i + 2];
  out[o] = r;
  if (kept_times) memcpy(&kept_times[o], (const char *)&r + 24, 8);
}

// CHECK(points_buff_.empty() || pt.time >= points_buff_.back().time) (lidar_odometry.cc:491) for EVERY incoming point, kept
// or not, against the last point BUFFERED at that moment: the kept point in front of it in this message (the one whose output
// index is offsets[i] - 1), or the last point buffered before the message (prev_time; -inf when the buffer is empty)
__global__ void __launch_bounds__(256) k_prefilter_monotonic(const Pt48 *in, uint64_t n, const uint32_t *offsets, const double *kept_times,
                                                            uint64_t cap, double prev_time, uint32_t *status) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double t;
  memcpy(&t, (const char *)(in + i) + 24, 8);
  const uint32_t o = offsets[i];
  const double prev = o == 0u ? prev_time : (o - 1u < cap ? kept_times[o - 1u] : t);
  if (!(t >= prev)) status[2] = 1u;
}

// PACKED: the 20 bytes of a point the extraction reads (float xyz, 12 bytes apart | double time) instead of the 48-byte record
template <bool PACKED>
__global__ void __launch_bounds__(256) k_undistort(const Pt48 *in, uint64_t n, const wc_imu_state *__restrict__ imu, uint32_t n_imu,
                                                  Pt48 *out, float *xyz_out, double *time_out, uint32_t *status) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Pt48 r = in[i];
  const float *f = (const float *)&r;
  double t;
  memcpy(&t, (const char *)&r + 24, 8);
  uint32_t lo = 0, hi = n_imu;  // std::lower_bound (cc:147)
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (imu[mid].t < t)
      lo = mid + 1;
    else
      hi = mid;
  }
  if (!(lo >= 1 && lo < n_imu)) {  // CHECK(idx >= 1 && idx < size) (cc:149)
    atomicOr(&status[1], 1u);
    return;
  }
  const wc_imu_state a = imu[lo - 1], b = imu[lo];
  const double fac = (t - a.t) / (b.t - a.t);
  const V3 pos = mk3(a.pos[0], a.pos[1], a.pos[2]) * (1 - fac) + mk3(b.pos[0], b.pos[1], b.pos[2]) * fac;
  const Q4 rot = qslerp(Q4{a.quat[0], a.quat[1], a.quat[2], a.quat[3]}, fac, Q4{b.quat[0], b.quat[1], b.quat[2], b.quat[3]});
  const V3 w = qrot(rot, mk3((double)f[0], (double)f[1], (double)f[2])) + pos;
  if (PACKED) {
    xyz_out[3 * i + 0] = (float)w.x, xyz_out[3 * i + 1] = (float)w.y, xyz_out[3 * i + 2] = (float)w.z;
    time_out[i] = t;
  } else {
    float *g = (float *)&r;
    g[0] = (float)w.x, g[1] = (float)w.y, g[2] = (float)w.z;
    out[i] = r;
  }
}

}  // namespace

static int prefilter_impl(wc_ctx *ctx, const void *d_pts_in, uint64_t n, const double ext_quat[4], const double ext_t[3], double min_range,
                          double max_range, const double blind_min[3], const double blind_max[3], void *d_pts_out, uint64_t cap,
                          uint64_t *h_n_out, bool check_time, double prev_time, double *d_kept_times, int *h_monotonic) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !h_n_out || (n && (!d_pts_in || !d_pts_out))) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  *h_n_out = 0;
  if (n == 0) return WC_OK;
  if (n >= (1ull << 32)) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  hipStream_t st = ctx->stream;
  FilterParams F;
  for (int i = 0; i < 4; ++i) F.q[i] = ext_quat[i];
  for (int i = 0; i < 3; ++i) F.t[i] = ext_t[i], F.bmin[i] = blind_min[i], F.bmax[i] = blind_max[i];
  F.min_range = min_range, F.max_range = max_range;
  WC_TRY(wc_ensure(ctx, ctx->b_misc[5], n * 12));
  WC_TRY(wc_ensure(ctx, ctx->b_misc[6], n * 8));
  WC_TRY(wc_ensure(ctx, ctx->b_status, 64 * 4));
  uint32_t *status = (uint32_t *)ctx->b_status.p;
  uint32_t *flags = (uint32_t *)ctx->b_misc[6].p, *offsets = flags + n;
  WC_HIP(ctx, hipMemsetAsync(status, 0, 64 * 4, st));
  const unsigned grid = (unsigned)((n + 255) / 256);
  k_prefilter_flags<<<grid, 256, 0, st>>>((const Pt48 *)d_pts_in, n, F, (float *)ctx->b_misc[5].p, flags);
  size_t tmp = 0;
  WC_HIP(ctx, rocprim::exclusive_scan(nullptr, tmp, flags, offsets, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
  WC_TRY(wc_ensure(ctx, ctx->b_misc[7], tmp + 16));
  tmp = ctx->b_misc[7].cap;
  WC_HIP(ctx, rocprim::exclusive_scan(ctx->b_misc[7].p, tmp, flags, offsets, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
  if (check_time && !d_kept_times) {  // the check reads the kept stamps: a scratch array when the caller wants none
    WC_TRY(wc_ensure(ctx, ctx->b_misc[4], std::min<uint64_t>(n, cap) * 8 + 8));
    d_kept_times = (double *)ctx->b_misc[4].p;
  }
  k_prefilter_scatter<<<grid, 256, 0, st>>>((const Pt48 *)d_pts_in, n, (const float *)ctx->b_misc[5].p, flags, offsets, (Pt48 *)d_pts_out,
                                           cap, status, d_kept_times);
  if (check_time) k_prefilter_monotonic<<<grid, 256, 0, st>>>((const Pt48 *)d_pts_in, n, offsets, d_kept_times, cap, prev_time, status);
  WC_HIP(ctx, hipGetLastError());
  WC_HIP(ctx, hipMemcpyAsync(ctx->h_status, status, 12, hipMemcpyDeviceToHost, st));
  WC_HIP(ctx, hipStreamSynchronize(st));
  *h_n_out = ctx->h_status[0];
  if (h_monotonic) *h_monotonic = ctx->h_status[2] ? 0 : 1;
  if (ctx->h_status[0] > cap) return wc_fail(ctx, WC_ERR_CAPACITY, "prefilter output capacity %llu < %u", (unsigned long long)cap, ctx->h_status[0]);
  return WC_OK;
}

extern "C" int wc_prefilter_points(wc_ctx *ctx, const void *d_pts_in, uint64_t n, const double ext_quat[4], const double ext_t[3],
                                   double min_range, double max_range, const double blind_min[3], const double blind_max[3],
                                   void *d_pts_out, uint64_t cap, uint64_t *h_n_out) {
  return prefilter_impl(ctx, d_pts_in, n, ext_quat, ext_t, min_range, max_range, blind_min, blind_max, d_pts_out, cap, h_n_out, false, 0.0,
                        nullptr, nullptr);
}

extern "C" int wc_prefilter_points_checked(wc_ctx *ctx, const void *d_pts_in, uint64_t n, const double ext_quat[4], const double ext_t[3],
                                           double min_range, double max_range, const double blind_min[3], const double blind_max[3],
                                           void *d_pts_out, uint64_t cap, uint64_t *h_n_out, double prev_time, double *d_kept_times,
                                           int *h_monotonic) {
  if (!h_monotonic) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  *h_monotonic = 1;
  return prefilter_impl(ctx, d_pts_in, n, ext_quat, ext_t, min_range, max_range, blind_min, blind_max, d_pts_out, cap, h_n_out, true, prev_time,
                        d_kept_times, h_monotonic);
}

static int undistort_impl(wc_ctx *ctx, const void *d_pts_in, uint64_t n, const wc_imu_state *d_imu, uint64_t n_imu, void *d_pts_out,
                          float *d_xyz_out, double *d_time_out) {
  wc_dev_guard dg_(ctx);
  const bool packed = d_pts_out == nullptr;
  if (!ctx || (n && (!d_pts_in || !d_imu || (packed ? (!d_xyz_out || !d_time_out) : false))))
    return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  if (n == 0) return WC_OK;
  hipStream_t st = ctx->stream;
  WC_TRY(wc_ensure(ctx, ctx->b_status, 64 * 4));
  uint32_t *status = (uint32_t *)ctx->b_status.p;
  WC_HIP(ctx, hipMemsetAsync(status, 0, 64 * 4, st));
  const unsigned grid = (unsigned)((n + 255) / 256);
  if (packed)
    k_undistort<true><<<grid, 256, 0, st>>>((const Pt48 *)d_pts_in, n, d_imu, (uint32_t)n_imu, nullptr, d_xyz_out, d_time_out, status);
  else
    k_undistort<false><<<grid, 256, 0, st>>>((const Pt48 *)d_pts_in, n, d_imu, (uint32_t)n_imu, (Pt48 *)d_pts_out, nullptr, nullptr, status);
  WC_HIP(ctx, hipGetLastError());
  WC_HIP(ctx, hipMemcpyAsync(ctx->h_status, status, 8, hipMemcpyDeviceToHost, st));
  WC_HIP(ctx, hipStreamSynchronize(st));
  if (ctx->h_status[1]) return wc_fail(ctx, WC_ERR_RANGE, "point timestamp outside the IMU state range");
  return WC_OK;
}

extern "C" int wc_undistort_sweep(wc_ctx *ctx, const void *d_pts_in, uint64_t n, const wc_imu_state *d_imu, uint64_t n_imu, void *d_pts_out) {
  if (n && !d_pts_out) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  return undistort_impl(ctx, d_pts_in, n, d_imu, n_imu, d_pts_out, nullptr, nullptr);
}

extern "C" int wc_undistort_sweep_packed(wc_ctx *ctx, const void *d_pts_in, uint64_t n, const wc_imu_state *d_imu, uint64_t n_imu,
                                         float *d_xyz_out, double *d_time_out) {
  return undistort_impl(ctx, d_pts_in, n, d_imu, n_imu, nullptr, d_xyz_out, d_time_out);
}

int wc_touch_sweep() {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, (const void *)k_prefilter_flags) == hipSuccess ? WC_OK : WC_ERR_HIP;
}
