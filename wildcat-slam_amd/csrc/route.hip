// route.hip — ONE point cloud over several GPUs (BASELINE config 5: a 10 M-point accumulated
// cloud on 8 MI355X).
//
// The reference processes root voxels independently once the points are binned (BuildVoxelMap,
// src/odometry/surfel_extraction.cc:217-219 and the emission loop :330-332), but a voxel needs ALL its points, in time
// order (ClusterSurfels :22-29).  So a cloud shards by ROOT VOXEL:
//   1. every rank keys its time-contiguous slice of the cloud (VoxelLoc, surfel_extraction.h:59-64) and partitions it by
//      owner = hash(root voxel index) mod world — a STABLE partition, time order survives inside a segment (k_route_*),
//   2. ONE exchange step (all-to-all of 24-byte {xyz, t} records; xGMI) hands every rank the points of the voxels it owns,
//      segments concatenated in source-rank order = global time order,
//   3. the rank runs the ordinary extraction (wc_extract_surfels) on what it received: its voxels are complete, so the
//      surfels are the ones the unsharded call produces for those voxels, bit for bit,
//   4. the surfel lists are disjoint by voxel; an all-gather + k-way merge by (timestamp, id) reproduces the unsharded
//      output order (wc_merge_surfels) where the replicated window needs it.
// The collectives go through the ctx's wc_comm (callbacks: tests, torch.distributed; or the in-library RCCL binding of
// comm.hip).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "ctx.h"

namespace {

constexpr int kRouteTile = 2048;  // points per workgroup
constexpr int kRouteThreads = 256;
constexpr int kRouteIters = kRouteTile / kRouteThreads;
constexpr int kMaxWorld = 64;

__host__ __device__ inline uint32_t route_hash(int32_t kx, int32_t ky, int32_t kz) {
  uint64_t h = (uint64_t)(uint32_t)kx * 0x9E3779B97F4A7C15ull;
  h ^= (uint64_t)(uint32_t)ky * 0xC2B2AE3D27D4EB4Full;
  h ^= (uint64_t)(uint32_t)kz * 0x165667B19E3779F9ull;
  h ^= h >> 29;
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 32;
  return (uint32_t)h;
}

__device__ __forceinline__ int route_vox(double p, double vs) { return (int)floor(p / vs); }  // true fp64 division (h:59-64)

__device__ __forceinline__ void route_load(const wc_points &pts, uint64_t i, float &x, float &y, float &z, double &t) {
  const float *f = (const float *)((const char *)pts.xyz + i * pts.xyz_stride);
  x = f[0], y = f[1], z = f[2];
  t = *(const double *)((const char *)pts.time + i * pts.time_stride);
}

__device__ __forceinline__ uint32_t route_owner(float x, float y, float z, double vs, uint32_t world) {
  return route_hash(route_vox((double)x, vs), route_vox((double)y, vs), route_vox((double)z, vs)) % world;
}

// pass 1: points per (owner, tile)
__global__ void __launch_bounds__(kRouteThreads) k_route_count(wc_points pts, double vs, uint32_t world, uint32_t tiles, uint32_t *cnt) {
  __shared__ uint32_t hist[kMaxWorld];
  if (threadIdx.x < kMaxWorld) hist[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * kRouteTile;
  for (int it = 0; it < kRouteIters; ++it) {
    const uint64_t i = base + (uint64_t)it * kRouteThreads + threadIdx.x;
    if (i < pts.n) {
      float x, y, z;
      double t;
      route_load(pts, i, x, y, z, t);
      atomicAdd(&hist[route_owner(x, y, z, vs, world)], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < world) cnt[(size_t)threadIdx.x * tiles + blockIdx.x] = hist[threadIdx.x];
}

// pass 2: stable scatter.  off = exclusive scan of cnt in (owner, tile) order: the segment of owner o starts at off[o * tiles],
// tile b's share of it at off[o * tiles + b]; inside the tile points keep their order (rank = same-owner points in front).
__global__ void __launch_bounds__(kRouteThreads) k_route_scatter(wc_points pts, double vs, uint32_t world, uint32_t tiles, const uint32_t *off,
                                                                wc_route_point *out) {
  __shared__ uint32_t run[kMaxWorld];                        // same-owner points of the earlier iterations of this tile
  __shared__ uint32_t wcnt[kRouteThreads / 64][kMaxWorld];  // per wavefront of this iteration
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < kMaxWorld) run[threadIdx.x] = 0;
  const uint64_t base = (uint64_t)blockIdx.x * kRouteTile;
  for (int it = 0; it < kRouteIters; ++it) {
    if (lane < kMaxWorld) wcnt[wave][lane] = 0;
    __syncthreads();
    const uint64_t i = base + (uint64_t)it * kRouteThreads + threadIdx.x;
    const bool live = i < pts.n;
    float x = 0, y = 0, z = 0;
    double t = 0;
    uint32_t o = 0xFFFFFFFFu, lane_rank = 0;
    if (live) {
      route_load(pts, i, x, y, z, t);
      o = route_owner(x, y, z, vs, world);
    }
    // rank inside the wavefront: one round per owner present
    unsigned long long todo = __ballot(live);
    while (todo) {
      const int first = __ffsll((long long)todo) - 1;
      const uint32_t o0 = (uint32_t)__shfl((int)o, first);
      const unsigned long long m = __ballot(live && o == o0);
      if (live && o == o0) {
        lane_rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == first) wcnt[wave][o0] = (uint32_t)__popcll(m);
      }
      todo &= ~m;
    }
    __syncthreads();
    if (live) {
      uint32_t before = run[o] + lane_rank;
      for (int w = 0; w < wave; ++w) before += wcnt[w][o];
      wc_route_point r;
      r.x = x, r.y = y, r.z = z, r.src = 0u, r.t = t;
      out[(size_t)off[(size_t)o * tiles + blockIdx.x] + before] = r;
    }
    __syncthreads();
    if (threadIdx.x < world) {
      uint32_t s = 0;
      for (int w = 0; w < kRouteThreads / 64; ++w) s += wcnt[w][threadIdx.x];
      run[threadIdx.x] += s;
    }
    __syncthreads();
  }
}

// ---- k-way merge of time-sorted surfel lists --------------------------------------------------------------------------
struct MergeLists {
  uint64_t off[kMaxWorld + 1];
  int k;
};

// the canonical surfel order of the extraction (and of the oracle): timestamp, ties by root voxel index and node id
__device__ __forceinline__ bool surfel_less(double ta, const wc_surfel_id &a, double tb, const wc_surfel_id &b, bool have_ids) {
  if (ta != tb) return ta < tb;
  if (!have_ids) return false;
  if (a.kx != b.kx) return a.kx < b.kx;
  if (a.ky != b.ky) return a.ky < b.ky;
  if (a.kz != b.kz) return a.kz < b.kz;
  return a.node < b.node;
}

__global__ void __launch_bounds__(256) k_merge_surfels(const wc_surfel *in, const wc_surfel_id *in_ids, MergeLists L, wc_surfel *out,
                                                      wc_surfel_id *out_ids) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= L.off[L.k]) return;
  int mine = 0;
  while (g >= L.off[mine + 1]) ++mine;
  const double t = in[g].t;
  const bool have_ids = in_ids != nullptr;
  wc_surfel_id id{0, 0, 0, 0u};
  if (have_ids) id = in_ids[g];
  uint64_t rank = g - L.off[mine];
  for (int j = 0; j < L.k; ++j) {
    if (j == mine) continue;
    // elements of list j in front of this one: strictly smaller, and (for lists before mine) equal ones too
    uint64_t lo = L.off[j], hi = L.off[j + 1];
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      const double tm = in[mid].t;
      wc_surfel_id im{0, 0, 0, 0u};
      if (have_ids) im = in_ids[mid];
      const bool in_front = j < mine ? !surfel_less(t, id, tm, im, have_ids) : surfel_less(tm, im, t, id, have_ids);
      if (in_front)
        lo = mid + 1;
      else
        hi = mid;
    }
    rank += lo - L.off[j];
  }
  out[rank] = in[g];
  if (have_ids && out_ids) out_ids[rank] = id;
}

}  // namespace

extern "C" int wc_route_owner(int32_t kx, int32_t ky, int32_t kz, int world) {
  if (world <= 0) return -1;
  return (int)(route_hash(kx, ky, kz) % (uint32_t)world);
}

extern "C" int wc_route_partition(wc_ctx *ctx, const wc_points *pts, int world, wc_route_point *d_send, uint64_t *h_counts) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !pts || !h_counts || world < 1 || world > kMaxWorld || (pts->n && !d_send))
    return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument (1 <= world <= %d)", __func__, kMaxWorld);
  for (int r = 0; r < world; ++r) h_counts[r] = 0;
  const uint64_t n = pts->n;
  if (n == 0) return WC_OK;
  if (n >= (1ull << 32)) return wc_fail(ctx, WC_ERR_ARG, "at most 2^32-1 points per call");
  hipStream_t st = ctx->stream;
  const uint32_t tiles = (uint32_t)((n + kRouteTile - 1) / kRouteTile);
  const size_t cells = (size_t)world * tiles + 1;
  wc_buf &b_cnt = ctx->b_misc[5], &b_off = ctx->b_misc[6], &b_tmp = ctx->b_misc[7];
  WC_TRY(wc_ensure(ctx, b_cnt, cells * 4));
  WC_TRY(wc_ensure(ctx, b_off, cells * 4));
  uint32_t *cnt = (uint32_t *)b_cnt.p, *off = (uint32_t *)b_off.p;
  WC_HIP(ctx, hipMemsetAsync(cnt + (cells - 1), 0, 4, st));
  const double vs = (double)ctx->P.voxel_size;
  k_route_count<<<tiles, kRouteThreads, 0, st>>>(*pts, vs, (uint32_t)world, tiles, cnt);
  size_t tmp = 0;
  WC_HIP(ctx, rocprim::exclusive_scan(nullptr, tmp, cnt, off, 0u, cells, rocprim::plus<uint32_t>(), st));
  WC_TRY(wc_ensure(ctx, b_tmp, tmp + 16));
  tmp = b_tmp.cap;
  WC_HIP(ctx, rocprim::exclusive_scan(b_tmp.p, tmp, cnt, off, 0u, cells, rocprim::plus<uint32_t>(), st));
  k_route_scatter<<<tiles, kRouteThreads, 0, st>>>(*pts, vs, (uint32_t)world, tiles, off, d_send);
  WC_HIP(ctx, hipGetLastError());
  // segment starts: off[o * tiles], o = 0..world (the last cell holds the total)
  std::vector<uint32_t> starts((size_t)world + 1);
  WC_HIP(ctx, hipMemcpy2DAsync(starts.data(), 4, off, (size_t)tiles * 4, 4, (size_t)world, hipMemcpyDeviceToHost, st));
  WC_HIP(ctx, hipMemcpyAsync(&starts[world], off + (cells - 1), 4, hipMemcpyDeviceToHost, st));
  WC_HIP(ctx, hipStreamSynchronize(st));
  for (int r = 0; r < world; ++r) h_counts[r] = (uint64_t)(starts[r + 1] - starts[r]);
  if (starts[world] != n) return wc_fail(ctx, WC_ERR_HIP, "internal: partition lost points (%u of %llu)", starts[world], (unsigned long long)n);
  return WC_OK;
}

extern "C" int wc_merge_surfels(wc_ctx *ctx, const wc_surfel *d_in, const wc_surfel_id *d_in_ids, const uint64_t *h_counts, int k,
                                wc_surfel *d_out, wc_surfel_id *d_out_ids) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !h_counts || k < 1 || k > kMaxWorld) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  MergeLists L;
  L.k = k;
  L.off[0] = 0;
  for (int i = 0; i < k; ++i) L.off[i + 1] = L.off[i] + h_counts[i];
  const uint64_t n = L.off[k];
  if (n == 0) return WC_OK;
  if (!d_in || !d_out || d_in == d_out) return wc_fail(ctx, WC_ERR_ARG, "%s: input and output must be distinct buffers", __func__);
  k_merge_surfels<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(d_in, d_in_ids, L, d_out, d_out_ids);
  WC_HIP(ctx, hipGetLastError());
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WC_OK;
}

// ---- the collective steps, through the ctx's communicator ---------------------------------------------------------------
extern "C" int wc_ctx_set_comm(wc_ctx *ctx, const wc_comm *comm) {
  wc_dev_guard dg_(ctx);
  if (!ctx) return wc_fail(ctx, WC_ERR_ARG, "%s: null context", __func__);
  if (comm) {
    if (comm->world < 1 || comm->world > kMaxWorld || comm->rank < 0 || comm->rank >= comm->world)
      return wc_fail(ctx, WC_ERR_ARG, "wc_ctx_set_comm: bad rank / world");
    ctx->comm = *comm;
    ctx->have_comm = true;
  } else {
    ctx->have_comm = false;
    std::memset(&ctx->comm, 0, sizeof(ctx->comm));
  }
  return WC_OK;
}

static int need_comm(wc_ctx *ctx, const char *fn, bool a2a, bool ag) {
  if (!ctx->have_comm) return wc_fail(ctx, WC_ERR_ARG, "%s: no communicator installed (wc_ctx_set_comm / wc_comm_rccl_init)", fn);
  if ((a2a && !ctx->comm.alltoallv) || (ag && !ctx->comm.allgatherv))
    return wc_fail(ctx, WC_ERR_ARG, "%s: the communicator lacks the collective this call needs", fn);
  return WC_OK;
}

// every rank learns every rank's count: an all-gather of one u64 through the byte-wise allgatherv on a device staging word
static int gather_counts(wc_ctx *ctx, uint64_t mine, uint64_t *h_all) {
  const int world = ctx->comm.world;
  WC_TRY(wc_ensure(ctx, ctx->b_status, 64 * 4 + (size_t)(world + 1) * 8));
  uint64_t *d_one = (uint64_t *)((char *)ctx->b_status.p + 256), *d_all = d_one + 1;
  WC_HIP(ctx, hipMemcpyAsync(d_one, &mine, 8, hipMemcpyHostToDevice, ctx->stream));
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<uint64_t> eight((size_t)world, 8);
  if (ctx->comm.allgatherv(ctx->comm.user, d_one, 8, d_all, eight.data()) != 0) return wc_fail(ctx, WC_ERR_HIP, "allgatherv callback failed");
  WC_HIP(ctx, hipMemcpyAsync(h_all, d_all, (size_t)world * 8, hipMemcpyDeviceToHost, ctx->stream));
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WC_OK;
}

extern "C" int wc_extract_surfels_sharded(wc_ctx *ctx, const wc_points *pts, double t_lo, double t_hi, wc_surfel *d_out, wc_surfel_id *d_ids,
                                          uint64_t cap, uint64_t *h_n_out, uint64_t *h_n_points_owned) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !pts || !h_n_out) return wc_fail(ctx, WC_ERR_ARG, "%s: null argument", __func__);
  WC_TRY(need_comm(ctx, __func__, true, false));
  if (t_lo > t_hi) return wc_fail(ctx, WC_ERR_ARG, "%s needs the GLOBAL time range of the cloud in t_lo, t_hi", __func__);
  const int world = ctx->comm.world;
  // 1. partition the local slice by owner
  wc_buf &b_send = ctx->b_route[0], &b_recv = ctx->b_route[1];
  WC_TRY(wc_ensure(ctx, b_send, std::max<uint64_t>(pts->n, 1) * sizeof(wc_route_point)));
  std::vector<uint64_t> send_cnt((size_t)world), recv_cnt((size_t)world), send_b((size_t)world), recv_b((size_t)world);
  WC_TRY(wc_route_partition(ctx, pts, world, (wc_route_point *)b_send.p, send_cnt.data()));
  // 2. counts, then the points: ONE exchange step of 24-byte records
  {
    WC_TRY(wc_ensure(ctx, ctx->b_status, 64 * 4 + (size_t)(2 * world) * 8));
    uint64_t *d_sc = (uint64_t *)((char *)ctx->b_status.p + 256), *d_rc = d_sc + world;
    WC_HIP(ctx, hipMemcpyAsync(d_sc, send_cnt.data(), (size_t)world * 8, hipMemcpyHostToDevice, ctx->stream));
    WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint64_t> eight((size_t)world, 8);
    if (ctx->comm.alltoallv(ctx->comm.user, d_sc, eight.data(), d_rc, eight.data()) != 0) return wc_fail(ctx, WC_ERR_HIP, "alltoallv callback failed");
    WC_HIP(ctx, hipMemcpyAsync(recv_cnt.data(), d_rc, (size_t)world * 8, hipMemcpyDeviceToHost, ctx->stream));
    WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  uint64_t n_recv = 0;
  for (int r = 0; r < world; ++r) {
    send_b[r] = send_cnt[r] * sizeof(wc_route_point);
    recv_b[r] = recv_cnt[r] * sizeof(wc_route_point);
    n_recv += recv_cnt[r];
  }
  if (h_n_points_owned) *h_n_points_owned = n_recv;
  WC_TRY(wc_ensure(ctx, b_recv, std::max<uint64_t>(n_recv, 1) * sizeof(wc_route_point)));
  if (ctx->comm.alltoallv(ctx->comm.user, b_send.p, send_b.data(), b_recv.p, recv_b.data()) != 0) return wc_fail(ctx, WC_ERR_HIP, "alltoallv callback failed");
  // 3. the ordinary extraction on the received records (segments arrive in source-rank order = time order)
  wc_points mine;
  mine.xyz = b_recv.p;
  mine.time = (const char *)b_recv.p + offsetof(wc_route_point, t);
  mine.xyz_stride = mine.time_stride = (uint32_t)sizeof(wc_route_point);
  mine.n = n_recv;
  return wc_extract_surfels(ctx, &mine, t_lo, t_hi, d_out, d_ids, cap, h_n_out);
}

extern "C" int wc_gather_surfels(wc_ctx *ctx, const wc_surfel *d_local, const wc_surfel_id *d_local_ids, uint64_t n_local, wc_surfel *d_out,
                                 wc_surfel_id *d_out_ids, uint64_t cap, uint64_t *h_n_out) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !h_n_out) return wc_fail(ctx, WC_ERR_ARG, "%s: null argument", __func__);
  WC_TRY(need_comm(ctx, __func__, false, true));
  const int world = ctx->comm.world;
  std::vector<uint64_t> cnt((size_t)world), bytes((size_t)world);
  WC_TRY(gather_counts(ctx, n_local, cnt.data()));
  uint64_t total = 0;
  for (int r = 0; r < world; ++r) total += cnt[r];
  *h_n_out = total;
  if (total > cap) return wc_fail(ctx, WC_ERR_CAPACITY, "output capacity %llu < %llu surfels", (unsigned long long)cap, (unsigned long long)total);
  if (total == 0) return WC_OK;
  wc_buf &b_s = ctx->b_route[2], &b_i = ctx->b_route[3];
  WC_TRY(wc_ensure(ctx, b_s, total * sizeof(wc_surfel)));
  for (int r = 0; r < world; ++r) bytes[r] = cnt[r] * sizeof(wc_surfel);
  if (ctx->comm.allgatherv(ctx->comm.user, d_local, n_local * sizeof(wc_surfel), b_s.p, bytes.data()) != 0)
    return wc_fail(ctx, WC_ERR_HIP, "allgatherv callback failed");
  const bool ids = d_local_ids != nullptr;
  if (ids) {
    WC_TRY(wc_ensure(ctx, b_i, total * sizeof(wc_surfel_id)));
    for (int r = 0; r < world; ++r) bytes[r] = cnt[r] * sizeof(wc_surfel_id);
    if (ctx->comm.allgatherv(ctx->comm.user, d_local_ids, n_local * sizeof(wc_surfel_id), b_i.p, bytes.data()) != 0)
      return wc_fail(ctx, WC_ERR_HIP, "allgatherv callback failed");
  }
  return wc_merge_surfels(ctx, (const wc_surfel *)b_s.p, ids ? (const wc_surfel_id *)b_i.p : nullptr, cnt.data(), world, d_out, d_out_ids);
}

int wc_touch_route() {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, (const void *)k_route_count) == hipSuccess ? WC_OK : WC_ERR_HIP;
}
