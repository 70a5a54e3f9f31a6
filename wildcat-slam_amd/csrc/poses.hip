// poses.hip — surfel pose update on gfx950.  Replaces UpdateSurfelPoses (src/odometry/lidar_odometry.cc:160-170)
// and Surfel::UpdatePose (src/odometry/surfel.h:48-58): per surfel a lower_bound over the IMU-state timestamps,
// lerp of position, slerp of rotation, and on the first update the world->body conversion of centre, normal and
// covariance.  One thread per surfel; the IMU table (<= a few thousand x 112 B) stays in L2.
#include <hip/hip_runtime.h>

#include "ctx.h"
#include "dmath.h"

namespace {
using namespace wc;

__global__ void __launch_bounds__(256) k_update_poses(const wc_imu_state *__restrict__ imu, uint32_t n_imu, wc_surfel *surf,
                                                     wc_pose *pose, uint8_t *in_body, uint64_t n, uint32_t *status) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const double t = surf[s].t;
  uint32_t lo = 0, hi = n_imu;  // std::lower_bound: first state with timestamp >= t (cc:162)
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (imu[mid].t < t)
      lo = mid + 1;
    else
      hi = mid;
  }
  if (lo == 0 || lo == n_imu) {  // CHECK(idx != 0 && idx != size) (cc:164)
    atomicOr(&status[1], 1u);
    return;
  }
  const wc_imu_state a = imu[lo - 1], b = imu[lo];
  const double f = (t - a.t) / (b.t - a.t);
  const V3 pos = mk3(a.pos[0], a.pos[1], a.pos[2]) * (1 - f) + mk3(b.pos[0], b.pos[1], b.pos[2]) * f;
  const Q4 rot = qslerp(Q4{a.quat[0], a.quat[1], a.quat[2], a.quat[3]}, f, Q4{b.quat[0], b.quat[1], b.quat[2], b.quat[3]});
  wc_pose p;
  p.pos[0] = pos.x, p.pos[1] = pos.y, p.pos[2] = pos.z;
  p.quat[0] = rot.w, p.quat[1] = rot.x, p.quat[2] = rot.y, p.quat[3] = rot.z;
  pose[s] = p;
  if (!in_body[s]) {  // surfel.h:52-57
    in_body[s] = 1;
    wc_surfel sf = surf[s];
    const Q4 rc = qconj(rot);
    const V3 c = qrot(rc, mk3(sf.center[0], sf.center[1], sf.center[2]) - pos);
    const V3 nn = qrot(rc, mk3(sf.normal[0], sf.normal[1], sf.normal[2]));
    M3 C;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) C.m[i][j] = sf.cov[3 * i + j];
    const M3 R = qmat(rot);
    const M3 Cb = (transpose(R) * C) * R;
    sf.center[0] = c.x, sf.center[1] = c.y, sf.center[2] = c.z;
    sf.normal[0] = nn.x, sf.normal[1] = nn.y, sf.normal[2] = nn.z;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) sf.cov[3 * i + j] = Cb.m[i][j];
    surf[s] = sf;
  }
}
// dst[j] = src[n - 1 - j]: a batch of surfels leaving the sliding window oldest-first, each pushed to the FRONT of the fixed
// window (ShrinkToFit, lidar_odometry.cc:243-246), ends up newest-first
__global__ void __launch_bounds__(256) k_reverse_copy(const wc_surfel *ss, const wc_pose *sp, uint64_t n, wc_surfel *ds, wc_pose *dp) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  ds[j] = ss[n - 1 - j];
  dp[j] = sp[n - 1 - j];
}
}  // namespace

extern "C" int wc_reverse_copy_surfels(wc_ctx *ctx, const wc_surfel *d_src_surf, const wc_pose *d_src_pose, uint64_t n,
                                       wc_surfel *d_dst_surf, wc_pose *d_dst_pose) {
  wc_dev_guard dg_(ctx);
  if (!ctx || (n && (!d_src_surf || !d_src_pose || !d_dst_surf || !d_dst_pose)))
    return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  if (n == 0) return WC_OK;
  k_reverse_copy<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(d_src_surf, d_src_pose, n, d_dst_surf, d_dst_pose);
  WC_HIP(ctx, hipGetLastError());
  return WC_OK;
}

extern "C" int wc_update_surfel_poses(wc_ctx *ctx, const wc_imu_state *d_imu, uint64_t n_imu, wc_surfel *d_surf,
                                      wc_pose *d_pose, uint8_t *d_in_body, uint64_t n) {
  wc_dev_guard dg_(ctx);
  if (!ctx || (n && (!d_imu || !d_surf || !d_pose || !d_in_body))) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  if (n == 0) return WC_OK;
  WC_TRY(wc_ensure(ctx, ctx->b_status, 64 * 4));
  uint32_t *status = (uint32_t *)ctx->b_status.p;
  WC_HIP(ctx, hipMemsetAsync(status, 0, 64 * 4, ctx->stream));
  k_update_poses<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(d_imu, (uint32_t)n_imu, d_surf, d_pose, d_in_body, n, status);
  WC_HIP(ctx, hipGetLastError());
  WC_HIP(ctx, hipMemcpyAsync(ctx->h_status, status, 8, hipMemcpyDeviceToHost, ctx->stream));
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->h_status[1]) return wc_fail(ctx, WC_ERR_RANGE, "surfel timestamp outside the IMU state range");
  return WC_OK;
}

int wc_touch_poses() {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, (const void *)k_reverse_copy) == hipSuccess ? WC_OK : WC_ERR_HIP;
}
