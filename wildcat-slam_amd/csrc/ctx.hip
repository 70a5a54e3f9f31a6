// ctx.hip — context lifetime, default parameters, device-memory helpers and HIP-event timer of the C-ABI.
#include "ctx.h"

#include <cmath>
#include <cstring>
#include <mutex>

extern "C" const char *wc_version(void) { return "wildcat_hip 0.1 (gfx950)"; }

extern "C" int wc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" void wc_params_default(wc_params *p) {
  std::memset(p, 0, sizeof(*p));
  // BuildVoxelMap(points, Vector3d::Zero(), 0.8, 2, {20,20,20,20}, 0.01, 0.1, ...)  surfel_extraction.cc:327
  p->voxel_size = 0.8f;
  p->max_layer = 2;
  p->min_points = 20;
  p->planer_threshold = 0.01f;
  p->min_plane_likeness = 0.1;
  p->cluster_gap = 0.05;       // surfel_extraction.cc:24
  p->cluster_min_points = 20;  // surfel_extraction.cc:33
  // knn_surfel_matcher.h:37-41
  p->center_scale = 1.0;
  p->angular_scale = 5.0 * M_PI / 180.0;
  p->surfel_dist_max = 0.1;
  p->knn_k = 10;
  p->time_diff_min = 0.06;
  p->surfel_sigma0 = 0.05 / 6;  // cost_functor.h:24
  p->cauchy_a = 0.4;            // lidar_odometry.cc:270
  // lio_config.h:10-14,32,42-45
  const double gn = 0.00015198973532354657, an = 0.006308226052016165;
  const double gw = 0.00011673723527962174, aw = 2.664506559330434e-06;
  const double rate = 200, k = 0.01;
  p->w_gyr = 1 / (gn * std::sqrt(rate)) * k;
  p->w_acc = 1 / (an * std::sqrt(rate)) * k;
  p->w_bg = 1 / (gw / std::sqrt(rate)) * k;
  p->w_ba = 1 / (aw / std::sqrt(rate)) * k;
  p->imu_dt = 1 / rate;
  p->max_iterations = 100;
  p->reference_quirks = 1;
  p->exact_sums = 0;
}

static int check_params(wc_ctx *ctx, const wc_params *p) {
  if (p->max_layer < 0 || p->max_layer > 2) return wc_fail(ctx, WC_ERR_ARG, "max_layer must be 0..2");
  if (!(p->voxel_size > 0)) return wc_fail(ctx, WC_ERR_ARG, "voxel_size must be positive");
  if (p->cluster_min_points < 1 || p->min_points < 0) return wc_fail(ctx, WC_ERR_ARG, "bad point thresholds");
  if (p->knn_k < 1 || p->knn_k > 16) return wc_fail(ctx, WC_ERR_ARG, "knn_k must be 1..16");
  return WC_OK;
}

// the development options by name (wc_ctx_set_dev_option) and - in a -DWC_DEV_KNOBS build - by environment variable
namespace {
struct DevOpt {
  const char *name, *env;
  int wc_dev_opts::*field;
  bool flag_only;  // the variable's presence means 1
};
const DevOpt kDevOpts[] = {
    {"exact_sums", "WC_EXACT_SUMS", &wc_dev_opts::exact_sums, true},
    {"debug_skip", "WC_DEBUG_SKIP", &wc_dev_opts::debug_skip, false},
    {"fx_merge_min", "WC_FX_MERGE_MIN", &wc_dev_opts::fx_merge_min, false},
    {"fx_split", "WC_FX_SPLIT", &wc_dev_opts::fx_split, false},
    {"no_bucket_sort", "WC_NO_BUCKET_SORT", &wc_dev_opts::no_bucket_sort, true},
    {"ex_sync", "WC_EX_SYNC", &wc_dev_opts::ex_sync, true},
    {"kd_leaf", "WC_KD_LEAF", &wc_dev_opts::kd_leaf, false},
    {"knn_group", "WC_KNN_GROUP", &wc_dev_opts::knn_group, false},
    {"knn_early", "WC_KNN_EARLY", &wc_dev_opts::knn_early, false},
    {"knn_sort", "WC_KNN_SORT", &wc_dev_opts::knn_sort, false},
    {"match_pair_serial", "WC_MATCH_PAIR_SERIAL", &wc_dev_opts::match_pair_serial, true},
    {"match_pair_swap", "WC_MATCH_PAIR_SWAP", &wc_dev_opts::match_pair_swap, true},
    {"match_pair_hold", "WC_MATCH_PAIR_HOLD", &wc_dev_opts::match_pair_hold, false},
    {"lin_imu_apart", "WC_LIN_IMU_APART", &wc_dev_opts::lin_imu_apart, true},
    {"lin_unary_apart", "WC_LIN_UNARY_APART", &wc_dev_opts::lin_unary_apart, true},
    {"lin_post_apart", "WC_LIN_POST_APART", &wc_dev_opts::lin_post_apart, true},
    {"lm_dense", "WC_LM_DENSE", &wc_dev_opts::lm_dense, true},
    {"lm_back_chunks", "WC_LM_BACK_CHUNKS", &wc_dev_opts::lm_back_chunks, true},
    {"dbg_lm", "WC_DBG_LM", &wc_dev_opts::dbg_lm, false},
    {"lm_one_collective", "WC_LM_ONE_COLLECTIVE", &wc_dev_opts::lm_one_collective, true},
    {"lm_side_stream", "WC_LM_SIDE_STREAM", &wc_dev_opts::lm_side_stream, false},
    {"pcr_full_width", "WC_PCR_FULL_WIDTH", &wc_dev_opts::pcr_full_width, true},
    {"lin_unary_chunks", "WC_LIN_UNARY_CHUNKS", &wc_dev_opts::lin_unary_chunks, false},
    {"lin_pair", "WC_LIN_PAIR", &wc_dev_opts::lin_pair, false},
    {"lm_sync", "WC_LM_SYNC", &wc_dev_opts::lm_sync, true},
    {"lm_eval_pass", "WC_LM_EVAL_PASS", &wc_dev_opts::lm_eval_pass, true},
    {"pcr_ahead", "WC_PCR_AHEAD", &wc_dev_opts::pcr_ahead, true},
    {"lm_dense_radius", "WC_LM_DENSE_RADIUS", &wc_dev_opts::lm_dense_radius, false},
};
}  // namespace

namespace {
struct DevPool {
  hipMemPool_t pool = nullptr;
  int refs = 0;
};
std::mutex g_pool_mu;
DevPool g_pools[64];
}  // namespace
hipMemPool_t wc_pool_acquire(int device) {
  if (device < 0 || device >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  DevPool &d = g_pools[device];
  if (!d.pool) {
    int supported = 0;
    if (hipDeviceGetAttribute(&supported, hipDeviceAttributeMemoryPoolsSupported, device) != hipSuccess || !supported) {
      (void)hipGetLastError();
      return nullptr;
    }
    hipMemPoolProps props;
    std::memset(&props, 0, sizeof(props));
    props.allocType = hipMemAllocationTypePinned;
    props.handleTypes = hipMemHandleTypeNone;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = device;
    hipMemPool_t p = nullptr;
    if (hipMemPoolCreate(&p, &props) != hipSuccess || !p) {
      (void)hipGetLastError();
      return nullptr;
    }
    uint64_t keep = ~0ull;
    if (hipMemPoolSetAttribute(p, hipMemPoolAttrReleaseThreshold, &keep) != hipSuccess) (void)hipGetLastError();
    d.pool = p;
  }
  ++d.refs;
  return d.pool;
}
void wc_pool_release(int device, hipMemPool_t pool) {
  if (!pool || device < 0 || device >= 64) return;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  DevPool &d = g_pools[device];
  if (d.pool == pool && --d.refs == 0) {
    (void)hipMemPoolDestroy(d.pool);  // (every block was released stream-ordered and the streams were synchronised: the memory returns to the driver)
    d.pool = nullptr;
  }
}

extern "C" int wc_ctx_create(const wc_params *params, int device, wc_ctx **out) {
  if (!out) return WC_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return WC_ERR_NOGPU;
  wc_ctx *ctx = new wc_ctx;
  ctx->device = device;
  if (params)
    ctx->P = *params;
  else
    wc_params_default(&ctx->P);
  int rc = check_params(ctx, &ctx->P);
  if (rc != WC_OK) {
    delete ctx;
    return rc;
  }
  int prev_dev = -1;
  (void)hipGetDevice(&prev_dev);
  struct Restore {
    int d;
    ~Restore() {
      if (d >= 0) (void)hipSetDevice(d);
    }
  } restore{prev_dev};
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
      hipHostMalloc((void **)&ctx->h_status, 128 * sizeof(uint32_t)) != hipSuccess ||
      hipHostMalloc((void **)&ctx->h_mail, (64 + 4096) * sizeof(double)) != hipSuccess) {
    delete ctx;
    return WC_ERR_HIP;
  }
  ctx->stream = ctx->own_stream;
  {
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, ctx->h_status, 0) == hipSuccess) ctx->h_status_dev = (uint32_t *)dp;
    for (int q = 0; q < 128; ++q) ctx->h_status[q] = 0;
  }
  for (wc_buf &b : ctx->b_route) b.plain = true;  // (all-to-all / all-gather buffers of the sharded extraction and matcher)
  ctx->b_status.plain = true;  // (handed to collectives too: gather_counts, the count exchange, the poison all-reduce - ADVICE r5)
  // a PRIVATE stream-ordered memory pool per (process, device), shared by the contexts on that device and destroyed with the last of
  // them (rounds 4 - 5 raised the release threshold of the device's DEFAULT pool - a process-global side effect on every
  // other hipMallocAsync user, and the memory parked by wc_ctx_warmup outlived wc_ctx_destroy).  The pool keeps what is freed.
  ctx->pool = wc_pool_acquire(device);
  ctx->pool_ok = ctx->pool != nullptr;
#ifdef WC_DEV_KNOBS  // development build only (profiles/dev): the WC_* variables of DESIGN 5.1 seed the context's options
  for (const DevOpt &o : kDevOpts)
    if (const char *v = getenv(o.env)) ctx->dev.*(o.field) = o.flag_only ? 1 : atoi(v);
#endif
  *out = ctx;
  return WC_OK;
}

extern "C" int wc_ctx_set_dev_option(wc_ctx *ctx, const char *name, int value) {
  if (!ctx || !name) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  for (const DevOpt &o : kDevOpts)
    if (strcmp(o.name, name) == 0) {
      ctx->dev.*(o.field) = value;
      ctx->ex.fx_backoff = ctx->ex.fx_skip_calls = 0;
      if (ctx->aux) ctx->aux->dev = ctx->dev;
      for (wc_ctx *sub : ctx->batch_subs) sub->dev = ctx->dev, sub->ex.fx_backoff = sub->ex.fx_skip_calls = 0;
      return WC_OK;
    }
  return wc_fail(ctx, WC_ERR_ARG, "wc_ctx_set_dev_option: unknown option '%s'", name);
}

void wc_window_free(wc_ctx *ctx);  // window.hip
extern "C" int wc_comm_rccl_destroy(wc_ctx *ctx);  // comm.hip

extern "C" void wc_ctx_destroy(wc_ctx *ctx) {
  wc_dev_guard dg_(ctx);
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->pair_worker && ctx->pair_worker_free) ctx->pair_worker_free(ctx->pair_worker);  // (joins the helper thread first)
  ctx->pair_worker = nullptr;
  if (ctx->aux) {
    wc_ctx_destroy(ctx->aux);
    ctx->aux = nullptr;
  }
  if (ctx->ev_aux) (void)hipEventDestroy(ctx->ev_aux);
  if (ctx->ev_pair) (void)hipEventDestroy(ctx->ev_pair);
  for (hipEvent_t e : ctx->ev_knn)
    if (e) (void)hipEventDestroy(e);
  (void)wc_comm_rccl_destroy(ctx);
  wc_window_free(ctx);
  wc_buf *all[] = {&ctx->b_keys[0],      &ctx->b_keys[1],     &ctx->b_vals[0],     &ctx->b_vals[1],      &ctx->b_sorttmp,
                   &ctx->b_slots,        &ctx->b_slot_ids,    &ctx->b_slot_keys[0], &ctx->b_slot_keys[1], &ctx->b_slot_idx[0],
                   &ctx->b_slot_idx[1],  &ctx->b_cand,        &ctx->b_cand_meta,   &ctx->b_status,
                   &ctx->b_ex_ctrl};
  for (wc_buf *b : all) wc_buf_release(ctx, *b);
  for (wc_buf &b : ctx->b_misc) wc_buf_release(ctx, b);
  for (wc_buf &b : ctx->b_route) wc_buf_release(ctx, b);
  wc_buf_release(ctx, ctx->b_batch);
  for (wc_buf &b : ctx->b_kd) wc_buf_release(ctx, b);
  wc_buf_release(ctx, ctx->b_match_half);
  for (wc_ctx *sub : ctx->batch_subs) wc_ctx_destroy(sub);
  ctx->batch_subs.clear();
  for (wc_buf &b : ctx->b_fx) wc_buf_release(ctx, b);
  if (ctx->h_status) (void)hipHostFree(ctx->h_status);
  if (ctx->h_mail) (void)hipHostFree(ctx->h_mail);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  wc_buf_release(ctx, ctx->b_stage);
  (void)hipStreamSynchronize(ctx->stream);  // (the releases are stream ordered)
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  for (hipEvent_t e : ctx->ex_ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  wc_pool_release(ctx->device, ctx->pool);
  delete ctx;
}

extern "C" const char *wc_last_error(const wc_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }

extern "C" int wc_ctx_set_stream(wc_ctx *ctx, void *hip_stream) {
  wc_dev_guard dg_(ctx);
  if (!ctx) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return WC_OK;
}

extern "C" int wc_ctx_set_params(wc_ctx *ctx, const wc_params *params) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !params) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  WC_TRY(check_params(ctx, params));
  ctx->P = *params;
  ctx->ex.fx_backoff = ctx->ex.fx_skip_calls = 0;  // new parameters: the adaptive path choice of the extraction starts afresh
  return WC_OK;
}

extern "C" int wc_dev_alloc(wc_ctx *ctx, size_t bytes, void **d_ptr) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !d_ptr) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  WC_HIP(ctx, hipSetDevice(ctx->device));
  static const bool alloc_dbg = wc_log_env("WC_ALLOC_DEBUG");  // (read once per process)
  if (alloc_dbg) fprintf(stderr, "[alloc] wc_dev_alloc %zu bytes\n", bytes);
  WC_HIP(ctx, hipMalloc(d_ptr, bytes ? bytes : 1));
  return WC_OK;
}
extern "C" int wc_dev_free(wc_ctx *ctx, void *d_ptr) {
  wc_dev_guard dg_(ctx);
  if (!ctx) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  WC_HIP(ctx, hipFree(d_ptr));
  return WC_OK;
}
extern "C" int wc_h2d(wc_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
  wc_dev_guard dg_(ctx);
  if (!ctx) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  if (!bytes) return WC_OK;
  WC_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WC_OK;
}
extern "C" int wc_d2h(wc_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
  wc_dev_guard dg_(ctx);
  if (!ctx) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  if (!bytes) return WC_OK;
  WC_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WC_OK;
}
extern "C" int wc_d2d(wc_ctx *ctx, void *d_dst, const void *d_src, size_t bytes) {
  wc_dev_guard dg_(ctx);
  if (!ctx) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  if (!bytes) return WC_OK;
  WC_HIP(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WC_OK;
}
// one field out of every record of an array (e.g. the timestamps of fresh surfels): n elements of elem_bytes, src_stride apart
__global__ void __launch_bounds__(256) k_pack_strided(const uint32_t *src, uint32_t ew, uint32_t sw, uint64_t words, uint32_t *dst) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w < words) dst[w] = src[(w / ew) * sw + (w % ew)];
}
extern "C" int wc_d2h_strided(wc_ctx *ctx, void *h_dst, const void *d_src, size_t elem_bytes, size_t src_stride, size_t n) {
  wc_dev_guard dg_(ctx);
  if (!ctx || (n && (!h_dst || !d_src)) || elem_bytes == 0 || src_stride < elem_bytes)
    return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  if (!n) return WC_OK;
  // word-sized elements (the facade's read-back of the new surfels' stamps, 8 of 144 bytes each): packed by a kernel, ONE contiguous
  // copy into pinned memory, a host memcpy - a 2-D copy into pageable memory took ~80 us for 7 000 elements
  if (elem_bytes % 4 == 0 && src_stride % 4 == 0 && ((uintptr_t)d_src % 4) == 0 && n * elem_bytes <= ((size_t)64 << 20)) {
    const size_t bytes = n * elem_bytes;
    WC_TRY(wc_ensure(ctx, ctx->b_stage, bytes));
    if (ctx->h_stage_cap < bytes) {
      if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
      ctx->h_stage = nullptr, ctx->h_stage_cap = 0;
      const size_t want = std::max<size_t>(2 * bytes, (size_t)1 << 20);
      WC_HIP(ctx, hipHostMalloc(&ctx->h_stage, want));
      ctx->h_stage_cap = want;
    }
    const uint32_t ew = (uint32_t)(elem_bytes / 4), sw = (uint32_t)(src_stride / 4);
    const uint64_t words = (uint64_t)n * ew;
    k_pack_strided<<<(unsigned)((words + 255) / 256), 256, 0, ctx->stream>>>((const uint32_t *)d_src, ew, sw, words, (uint32_t *)ctx->b_stage.p);
    WC_HIP(ctx, hipGetLastError());
    WC_HIP(ctx, hipMemcpyAsync(ctx->h_stage, ctx->b_stage.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(h_dst, ctx->h_stage, bytes);
    return WC_OK;
  }
  WC_HIP(ctx, hipMemcpy2DAsync(h_dst, elem_bytes, d_src, src_stride, elem_bytes, n, hipMemcpyDeviceToHost, ctx->stream));
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WC_OK;
}
extern "C" int wc_memset(wc_ctx *ctx, void *d_dst, int value, size_t bytes) {
  wc_dev_guard dg_(ctx);
  if (!ctx) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  if (!bytes) return WC_OK;
  WC_HIP(ctx, hipMemsetAsync(d_dst, value, bytes, ctx->stream));
  return WC_OK;
}
extern "C" int wc_sync(wc_ctx *ctx) {
  wc_dev_guard dg_(ctx);
  if (!ctx) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WC_OK;
}
extern "C" int wc_timer_start(wc_ctx *ctx) {
  wc_dev_guard dg_(ctx);
  if (!ctx) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  WC_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  return WC_OK;
}
extern "C" int wc_timer_stop_ms(wc_ctx *ctx, float *h_ms) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !h_ms) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  WC_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  WC_HIP(ctx, hipEventSynchronize(ctx->ev1));
  WC_HIP(ctx, hipEventElapsedTime(h_ms, ctx->ev0, ctx->ev1));
  return WC_OK;
}

int wc_touch_extract();
int wc_touch_match();
int wc_touch_poses();
int wc_touch_sweep();
int wc_touch_route();
int wc_touch_window();
int wc_match_pair_prepare(wc_ctx *ctx);  // match.hip

// Everything a long-running caller wants out of its first sweeps (include/wildcat_hip.h): the code objects of all translation units
// loaded, wc_match_pair's helper context and thread created, `reserve_bytes` of HBM taken into the stream-ordered pool the scratch
// buffers grow from.
extern "C" int wc_ctx_warmup(wc_ctx *ctx, size_t reserve_bytes) {
  wc_dev_guard dg_(ctx);
  if (!ctx) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  if (wc_touch_extract() != WC_OK || wc_touch_match() != WC_OK || wc_touch_poses() != WC_OK || wc_touch_sweep() != WC_OK || wc_touch_route() != WC_OK ||
      wc_touch_window() != WC_OK)
    return wc_fail(ctx, WC_ERR_HIP, "wc_ctx_warmup: a code object could not be loaded");
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, (const void *)k_pack_strided);
  WC_TRY(wc_match_pair_prepare(ctx));
  {  // the copy engines: a copy above the runtime's small-copy limit takes another path than a small one, and the first use of a path
    // creates its queue (~10 ms, seen as a window build of 12 ms when the segment heads' read-back crossed 16 KB - round 5); every
    // direction and both size classes once, on the ctx stream and on the helper's
    void *hp = nullptr, *dp = nullptr;
    const size_t big = (size_t)1 << 20;
    if (hipHostMalloc(&hp, big) == hipSuccess && hipMalloc(&dp, 2 * big) == hipSuccess) {
      std::memset(hp, 0, big);
      hipStream_t sts[2] = {ctx->stream, ctx->aux ? ctx->aux->stream : ctx->stream};
      for (hipStream_t st : sts)
        for (size_t bytes : {(size_t)256, (size_t)12 << 10, (size_t)48 << 10, big}) {
          (void)hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, st);
          (void)hipMemcpyAsync((char *)dp + big, dp, bytes, hipMemcpyDeviceToDevice, st);
          (void)hipMemcpyAsync(hp, (char *)dp + big, bytes, hipMemcpyDeviceToHost, st);
          (void)hipMemsetAsync(dp, 0, bytes, st);
        }
      for (hipStream_t st : sts) (void)hipStreamSynchronize(st);
    }
    if (dp) (void)hipFree(dp);
    if (hp) (void)hipHostFree(hp);
    (void)hipGetLastError();
  }
  if (reserve_bytes && ctx->pool_ok) {
    void *p = nullptr;
    if (hipMallocFromPoolAsync(&p, reserve_bytes, ctx->pool, ctx->stream) == hipSuccess) (void)hipFreeAsync(p, ctx->stream);
    (void)hipGetLastError();
  }
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WC_OK;
}

