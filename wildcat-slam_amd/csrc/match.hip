// match.hip — surfel-to-surfel correspondence on gfx950.  Replaces KnnSurfelMatcher::BuildIndex / Match
// (src/odometry/knn_surfel_matcher.cc:3-49) and the FLANN kd-tree beneath it (cc:65-89):
//   * 6-D feature [centre_world / 1.0, normal_world / 5deg] per target (ToVector, cc:91-98)
//   * EXACT k = 10 nearest neighbours in squared L2 (FLANN SearchParams(-1, 0.0): exact, sorted), ties by index
//   * first neighbour passing |dt| >= 0.06, acos(n.n) <= 5deg, |n.(c - c')| <= 0.1 and "pair not seen yet" (cc:25-47)
// Index: a kd-tree with bounding boxes over the 6-D features, built on the device for every call (match_tree.inc); a query
// walks it three levels at a time, nearest box first, and prunes against its k-th distance in ALL six dimensions; the first
// look at a box or a point is fp32 and conservative, the distances that enter the list are summed in fp64 in
// flann::L2_Simple's order, so indices and distances are the reference's bit for bit.  The reference's order dependence
// (std::set of already-paired surfels, queries visited in order) is a recurrence choice(q) = f(choice(c) : c < q); it is
// solved by fixed-point iteration on the device.  Latency-bound gather work; no MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "ctx.h"
#include "dmath.h"

using namespace wc;

// wc_match_pair's helper thread, kept between calls (a thread per call cost its creation - 60 to 100 us before the second
// search's first launch - on every odometry step).  Owned by the ctx through ctx->pair_worker / pair_worker_free.
struct wc_pair_worker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<void()> job;
  bool has = false, done = true, quit = false;
  wc_pair_worker() {
    th = std::thread([this] {
      std::unique_lock<std::mutex> lk(m);
      for (;;) {
        cv.wait(lk, [&] { return has || quit; });
        if (quit) return;
        has = false;
        lk.unlock();
        job();  // (never throws: wc_match_pair wraps the search)
        lk.lock();
        done = true;
        cv.notify_all();
      }
    });
  }
  void start(std::function<void()> fn) {
    std::lock_guard<std::mutex> lk(m);
    job = std::move(fn), has = true, done = false;
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return done; });
  }
  ~wc_pair_worker() {
    {
      std::lock_guard<std::mutex> lk(m);
      quit = true;
      cv.notify_all();
    }
    if (th.joinable()) th.join();
  }
};

// wc_match_pair, round 6: the fixed-window search's walk is held back until the sliding-window search's TREE is built.  Without the
// leaf-order sort the fixed-window search (a small tree, built sooner) reaches its walk while the other search is still building: the
// walk's one-wavefront workgroups then fill every slot, and the build's large workgroups (1 024 threads, 50 - 100 KB of LDS) wait for
// a compute unit to drain - room surfels, 55 k queries: the pair 0.73 -> 0.84 ms.  One event, recorded behind the sliding-window
// search's build and waited for in front of the other's walk; the host side is a flag under a mutex (the waiting thread has enqueued
// its own build by then).  post() is called on EVERY way out of the sliding-window search (without an event when it failed early).
struct wc_pair_sync {
  std::mutex m;
  std::condition_variable cv;
  bool posted = false, have_event = false;
  hipEvent_t ev = nullptr;
  void post(bool with_event) {
    std::lock_guard<std::mutex> lk(m);
    if (posted) return;
    posted = true, have_event = with_event;
    cv.notify_all();
  }
  bool wait() {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return posted; });
    return have_event;
  }
};

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;

struct MatchParams {
  double cs, as;        // centre / angular scale
  double time_min, ang_max, dist_max;
  double cos_acc, plane_acc;  // EARLY walks (match_tree.inc): accept a candidate when cos >= cos_acc and the plane distance <= plane_acc - ...
  int k, same_set;
};

__device__ __forceinline__ void feature6(const wc_surfel &s, const wc_pose &p, double cs, double as, double f[6], V3 &cw, V3 &nw) {
  const Q4 q{p.quat[0], p.quat[1], p.quat[2], p.quat[3]};
  cw = qrot(q, mk3(s.center[0], s.center[1], s.center[2])) + mk3(p.pos[0], p.pos[1], p.pos[2]);  // surfel.h:67-69
  nw = qrot(q, mk3(s.normal[0], s.normal[1], s.normal[2]));                                      // surfel.h:78-80
  f[0] = cw.x / cs, f[1] = cw.y / cs, f[2] = cw.z / cs;
  f[3] = nw.x / as, f[4] = nw.y / as, f[5] = nw.z / as;
}

// features + world-frame copies of the targets; a non-finite feature is reported (status bit 4) and zeroed so that the
// traversal's comparisons stay ordered
__global__ void __launch_bounds__(256) k_features(const wc_surfel *surf, const wc_pose *pose, uint32_t n, double cs, double as,
                                                 double *feat, double *world, uint32_t *status) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double f[6];
  V3 cw, nw;
  feature6(surf[i], pose[i], cs, as, f, cw, nw);
  bool bad = false;
  for (int d = 0; d < 6; ++d) {
    if (!(fabs(f[d]) < 1e18)) bad = true, f[d] = 0.0;
    feat[(size_t)i * 6 + d] = f[d];
  }
  if (bad) atomicOr(&status[1], 4u);
  double *w = world + (size_t)i * 7;
  w[0] = cw.x, w[1] = cw.y, w[2] = cw.z, w[3] = nw.x, w[4] = nw.y, w[5] = nw.z, w[6] = surf[i].t;
}

#include "match_tree.inc"

// The searches write a query's gated list where the query stands in the CELL order (position-major: 4 k contiguous bytes per
// thread, coalesced); the resolve rounds read plane j of the lists by query index.  Round 2 wrote the planes straight from the
// search, gated[j nq + q] with q = qorder[qi]: ten scattered 4-byte stores per query, 342 MB of write traffic for 40 MB of lists at
// a million queries.  Now: ONE scattered plane (the inverse permutation) and a transposition whose reads are 4 k-byte runs and
// whose writes are coalesced.
__global__ void __launch_bounds__(256) k_inv_perm(const uint32_t *__restrict__ qorder, uint32_t nq, uint32_t *qpos) {
  const uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x;
  if (qi < nq) qpos[qorder ? qorder[qi] : qi] = qi;
}
__global__ void __launch_bounds__(256) k_gated_planes(const uint32_t *__restrict__ pos_major, const uint32_t *__restrict__ qpos, uint32_t nq, int k,
                                                     uint32_t *gated) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const uint32_t *src = pos_major + (size_t)qpos[q] * k;
  for (int j = 0; j < k; ++j) gated[(size_t)j * nq + q] = src[j];
}

// choice(q) = first gated candidate c that is not already paired with q from c's own turn (c < q and choice(c) == q).
// "Something changed" is reported with ONE plain store per workgroup (idempotent: every writer stores 1): an atomicOr per
// wavefront was 15 600 atomics on one word in the first round of a 1 M-query match - 83 us per round for 72 MB of traffic.
// first != 0: the round that starts from "nobody has chosen" - choice_in is not read (no memset of it in front of the rounds).
__global__ void __launch_bounds__(256) k_resolve(const uint32_t *gated, uint32_t nq, int k, int same_set, int first, const uint32_t *choice_in,
                                                uint32_t *choice_out, uint32_t *changed) {
  __shared__ uint32_t s_changed;
  if (threadIdx.x == 0) s_changed = 0u;
  __syncthreads();
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nq) {
    uint32_t pick = kNone;
    for (int j = 0; j < k; ++j) {
      const uint32_t c = gated[(size_t)j * nq + q];
      if (c == kNone) break;
      if (same_set && !first && c < q && choice_in[c] == q) continue;  // {c, q} is already in surfel_pairs (cc:35-38)
      pick = c;
      break;
    }
    choice_out[q] = pick;
    if (pick != (first ? kNone : choice_in[q])) s_changed = 1u;
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_changed) *changed = 1u;
}

struct ChoiceFlag {  // 1 where a query has chosen a partner: the scan of the compaction reads the choices through it (no pass, no array of flags)
  __host__ __device__ uint32_t operator()(uint32_t c) const { return c != kNone ? 1u : 0u; }
};

__global__ void __launch_bounds__(256) k_emit_pairs(const uint32_t *choice, const uint32_t *offsets, uint32_t nq, const wc_surfel *q_surf,
                                                   const double *tworld, int same_set, wc_pair *pairs, uint64_t cap, uint32_t *status) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const uint32_t c = choice[q];
  const uint32_t o = offsets[q];
  if (q == nq - 1) status[0] = o + (c != kNone ? 1u : 0u);  // total (one atomicMax per pair on this word cost 0.18 ms per 1 M queries)
  if (c == kNone) return;
  if (o >= cap) return;
  const double tq = q_surf[q].t, tc = tworld[(size_t)c * 7 + 6];
  if (same_set) {
    pairs[o] = (tq < tc) ? wc_pair{(int32_t)q, (int32_t)c} : wc_pair{(int32_t)c, (int32_t)q};  // (older, newer), cc:41-45
  } else {
    if (!(tc < tq)) atomicOr(&status[1], 2u);  // the fixed-window surfel must be the older one (CHECK_LT, cc:301)
    pairs[o] = wc_pair{(int32_t)c, (int32_t)q};
  }
}

}  // namespace

// ---- the tree of one call: plan on the host (sizes only depend on nt), build on the device --------------------------------------
struct KdPlan {
  int T = 0, Bd = 0, D = 0, first = 3;
  int stages[4] = {0, 0, 0, 0}, nstage = 0;
};
static KdPlan kd_plan(uint32_t nt, int leaf_opt) {  // leaf_opt: the development option kd_leaf (0: leaves of 4 - 8 points)
  KdPlan p;
  const double leaf = leaf_opt > 0 ? (double)leaf_opt : 8.0;
  if (nt > 1024u) p.T = std::min(kKdTMax, (int)std::ceil(std::log2((double)nt / 512.0)));
  const double avg = (double)nt / (double)(1u << p.T);
  p.Bd = std::max(0, std::min(kKdBdMax, (int)std::ceil(std::log2(std::max(avg / leaf, 1.0)))));
  p.D = p.T + p.Bd;
  p.first = p.D % kW ? p.D % kW : kW;
  for (int rem = p.T; rem > 0;) {  // top stages of at most kKdStageMax levels, as even as possible
    const int nst = (rem + kKdStageMax - 1) / kKdStageMax, ts = (rem + nst - 1) / nst;
    p.stages[p.nstage++] = ts;
    rem -= ts;
  }
  return p;
}

static int kd_build(wc_ctx *ctx, const double *d_feat, uint32_t nt, const KdPlan &pl, KdTree &tree) {
  hipStream_t st = ctx->stream;
  wc_buf *B = ctx->b_kd;
  const size_t nbk = (size_t)1 << pl.T, nnode = (size_t)2 << pl.D;
  WC_TRY(wc_ensure(ctx, B[0], nbk * 4 + 64));
  WC_TRY(wc_ensure(ctx, B[1], nbk * 8 + 64));
  WC_TRY(wc_ensure(ctx, B[2], (size_t)nt * 4));
  WC_TRY(wc_ensure(ctx, B[3], nbk * kKdPad * 4));
  WC_TRY(wc_ensure(ctx, B[4], (nbk + 1) * 4));
  WC_TRY(wc_ensure(ctx, B[5], (nbk + 1) * 4));
  WC_TRY(wc_ensure(ctx, B[6], (size_t)nt * 4));
  WC_TRY(wc_ensure(ctx, B[7], (size_t)nt * 4));
  WC_TRY(wc_ensure(ctx, B[8], nnode * 48));
  WC_TRY(wc_ensure(ctx, B[9], (((size_t)1 << pl.D) + 1) * 4));
  WC_TRY(wc_ensure(ctx, ctx->b_match_half, (size_t)nt * 32));
  WC_TRY(wc_ensure(ctx, ctx->b_misc[3], (size_t)nt * 48));
  WC_TRY(wc_ensure(ctx, ctx->b_vals[1], (size_t)nt * 4));
  int *plane_dim = (int *)B[0].p;
  double *plane_val = (double *)B[1].p;
  uint32_t *bucket = (uint32_t *)B[2].p, *count = (uint32_t *)B[3].p;
  uint32_t *starts[2] = {(uint32_t *)B[4].p, (uint32_t *)B[5].p}, *idx[2] = {(uint32_t *)B[6].p, (uint32_t *)B[7].p};
  const uint32_t *starts_prev = nullptr;
  uint32_t *idx_prev = nullptr;
  int tcum = 0;
  for (int s = 0; s < pl.nstage; ++s) {
    k_kd_top<<<1u << tcum, kKdTopNT, 0, st>>>(d_feat, idx_prev, starts_prev, nt, tcum, pl.stages[s], plane_dim, plane_val, count);  // (also clears the stage's counters)
    tcum += pl.stages[s];
    k_kd_route<<<(nt + 1023) / 1024, 1024, 0, st>>>(d_feat, nt, tcum, plane_dim, plane_val, bucket, count);
    k_kd_scan<<<1, 1024, 0, st>>>(count, 1u << tcum, starts[s & 1]);
    k_kd_scatter<<<(nt + 1023) / 1024, 1024, 0, st>>>(bucket, nt, tcum, count, idx[s & 1]);
    starts_prev = starts[s & 1], idx_prev = idx[s & 1];
  }
  static const bool kd_dbg = wc_log_env("WC_MATCH_DEBUG");
  if (kd_dbg && starts_prev) {  // (debug only: a stream wait and a read-back) sizes of the buckets the bottom levels are built on
    std::vector<uint32_t> h((size_t)nbk + 1);
    if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h.data(), starts_prev, h.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
      uint32_t mx = 0, n512 = 0, n1024 = 0, n2048 = 0;
      for (size_t b = 0; b < nbk; ++b) {
        const uint32_t c = h[b + 1] - h[b];
        mx = std::max(mx, c), n512 += c > 512u, n1024 += c > 1024u, n2048 += c > 2048u;
      }
      fprintf(stderr, "[match] kd buckets: %zu of %.0f points on average, largest %u; above 512 / 1024 / 2048 points: %u / %u / %u\n", nbk, (double)nt / (double)nbk, mx, n512, n1024, n2048);
    }
  }
  const KdOut out{(float4 *)B[8].p, (uint32_t *)B[9].p, (float4 *)ctx->b_match_half.p, (double *)ctx->b_misc[3].p, (uint32_t *)ctx->b_vals[1].p};
  // (the launch for the few larger buckets first: its workgroups need 108 KB of LDS each, and behind the other launch it can find the
  // chip filled by another stream's walk - as an EMPTY launch it then waited 490 us for slots)
  if (pl.T > 0) k_kd_bottom<2048><<<1u << pl.T, kKdBotNT, 0, st>>>(d_feat, idx_prev, starts_prev, nt, pl.T, pl.Bd, out);  // (T = 0: one bucket of <= 1024)
  k_kd_bottom<1024><<<1u << pl.T, kKdBotNT, 0, st>>>(d_feat, idx_prev, starts_prev, nt, pl.T, pl.Bd, out);
  if (pl.T > 0) k_kd_top_boxes<<<1, 1024, 0, st>>>((float4 *)B[8].p, pl.T);
  WC_HIP(ctx, hipGetLastError());
  tree.box = (const float4 *)B[8].p, tree.leaf_begin = (const uint32_t *)B[9].p, tree.pts32 = (const float4 *)ctx->b_match_half.p;
  tree.sfeat = (const double *)ctx->b_misc[3].p, tree.sorig = (const uint32_t *)ctx->b_vals[1].p;
  tree.D = pl.D, tree.first = pl.first;
  return WC_OK;
}

// A rank of a query-sharded search that fails BEFORE the all-gather (an allocation, a kernel launch, the tree's sort) must not
// leave the others waiting in the collective: while this guard is armed, leaving match_impl enters the all-gather with a
// share of poison words (0xFEFEFEFE: no neighbour index, no kNone); every rank that finds one in the gathered table fails the
// call (k_peer_poison -> flag 16 of the control block).  The local failure's message is kept.
constexpr uint32_t kPeerPoison = 0xFEFEFEFEu;
__global__ void k_peer_poison(const uint32_t *__restrict__ table, const uint32_t *__restrict__ first_word, int world, uint32_t *status) {
  const int r = threadIdx.x;
  if (r < world && table[first_word[r]] == kPeerPoison) atomicOr(&status[1], 16u);
}
struct MatchPeerGuard {
  wc_ctx *ctx;
  uint32_t nq;
  int k;
  bool armed;
  ~MatchPeerGuard() {
    if (!armed) return;
    const std::string why = ctx->err;
    const uint32_t w = (uint32_t)ctx->comm.world, r = (uint32_t)ctx->comm.rank;
    const uint32_t mine = (uint32_t)(((uint64_t)nq * (r + 1)) / w - ((uint64_t)nq * r) / w);
    std::vector<uint64_t> bytes((size_t)w);
    for (uint32_t q = 0; q < w; ++q) bytes[q] = (uint64_t)(((uint64_t)nq * (q + 1)) / w - ((uint64_t)nq * q) / w) * k * 4;
    if (wc_ensure(ctx, ctx->b_route[2], (size_t)nq * k * 4) == WC_OK && wc_ensure(ctx, ctx->b_route[3], (size_t)(mine + 1) * k * 4) == WC_OK &&
        hipMemsetAsync(ctx->b_route[3].p, 0xFE, (size_t)(mine + 1) * k * 4, ctx->stream) == hipSuccess &&
        (ctx->comm.stream_ordered || hipStreamSynchronize(ctx->stream) == hipSuccess))
      (void)ctx->comm.allgatherv(ctx->comm.user, ctx->b_route[3].p, (uint64_t)mine * k * 4, ctx->b_route[2].p, bytes.data());
    ctx->err = why;
  }
};

// scratch lives in ctx->b_misc[1..7] slots to avoid another state struct.  want_shard: the call is a collective of the ctx's
// communicator (wc_match_sharded) - every rank makes it with the same replicated arguments
static int match_impl(wc_ctx *ctx, const wc_surfel *d_q_surf, const wc_pose *d_q_pose, uint64_t nq_, const wc_surfel *d_t_surf,
                      const wc_pose *d_t_pose, uint64_t nt_, int same_set, wc_pair *d_pairs, uint64_t cap, uint64_t *h_n_pairs,
                      uint32_t *d_knn_idx, double *d_knn_d2, bool want_shard) {
  wc_dev_guard dg_(ctx);
  const auto t_entry = std::chrono::steady_clock::now();
  wc_pair_sync *psync = ctx ? (wc_pair_sync *)ctx->pair_sync : nullptr;
  struct PairPost {  // (EVERY way out of a sliding-window search of a pair - the argument checks below included - releases the other search)
    wc_pair_sync *s;
    ~PairPost() {
      if (s) s->post(false);
    }
  } pair_post{same_set ? psync : nullptr};
  if (!ctx || !h_n_pairs) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  *h_n_pairs = 0;
  if (nt_ == 0 || nq_ == 0) return WC_OK;  // knn_surfel_matcher.cc:18-20
  if (nq_ >= (1ull << 31) || nt_ >= (1ull << 30)) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  // same_set: the queries ARE the targets (the sliding-window search, knn_surfel_matcher.cc:31-38) - their order is taken from the
  // tree's target permutation, so the two sets must be one array
  if (same_set && (nq_ != nt_ || d_q_surf != d_t_surf || d_q_pose != d_t_pose))
    return wc_fail(ctx, WC_ERR_ARG, "%s: same_set needs the query set to be the target set (nq %llu, nt %llu)", __func__, (unsigned long long)nq_, (unsigned long long)nt_);
  const uint32_t nq = (uint32_t)nq_, nt = (uint32_t)nt_;
  const wc_params &P = ctx->P;
  hipStream_t st = ctx->stream;
  // several GPUs: decided from arguments every rank shares (a rank-dependent predicate would leave the others in the all-gather)
  const bool sharded = want_shard && ctx->have_comm && ctx->comm.world > 1 && ctx->comm.allgatherv && nq >= 4096;
  // (the peer-poison table holds one word per rank in the control block's eight free words - a larger world would skip the
  // check and could feed a failed rank's poison to the resolve kernels as neighbour indices.  Every rank sees the same world: all refuse.)
  if (sharded && ctx->comm.world > 8) return wc_fail(ctx, WC_ERR_ARG, "wc_match_sharded: at most 8 ranks (one node), got %d", ctx->comm.world);
  MatchPeerGuard peer{ctx, nq, P.knn_k, sharded};
  wc_buf &b_feat = ctx->b_misc[1], &b_world = ctx->b_misc[2], &b_gated = ctx->b_misc[4], &b_choice = ctx->b_misc[5],
         &b_scan = ctx->b_misc[7];
  WC_TRY(wc_ensure(ctx, b_feat, (size_t)nt * 6 * 8));
  WC_TRY(wc_ensure(ctx, b_world, (size_t)nt * 7 * 8));
  WC_TRY(wc_ensure(ctx, b_gated, (size_t)nq * P.knn_k * 4));
  WC_TRY(wc_ensure(ctx, b_choice, (size_t)nq * 4 * 4));  // choice[2], flags, offsets
  WC_TRY(wc_ensure(ctx, ctx->b_keys[0], (size_t)nq * 4));
  WC_TRY(wc_ensure(ctx, ctx->b_vals[0], (size_t)nq * 4));
  WC_TRY(wc_ensure(ctx, ctx->b_status, 64 * 4));
  // ONE control block - [0] pairs, [1] flags, [32..39] "round r changed something", [40..55] the walk's sampled counts (8 x u64) -:
  // one memset in front of the call's first kernel, one copy to pinned memory behind its last
  uint32_t *status = (uint32_t *)ctx->b_status.p;
  uint32_t *changed = status + 32;
  unsigned long long *stats = (unsigned long long *)(status + 40);

  // 1. features of the targets, 2. their tree - no host round trip: the tree's shape only depends on nt
  WC_HIP(ctx, hipMemsetAsync(status, 0, 64 * 4, st));
  k_features<<<(nt + 255) / 256, 256, 0, st>>>(d_t_surf, d_t_pose, nt, P.center_scale, P.angular_scale, (double *)b_feat.p, (double *)b_world.p, status);
  const KdPlan plan = kd_plan(nt, ctx->dev.kd_leaf);
  KdTree tree;
  WC_TRY(kd_build(ctx, (const double *)b_feat.p, nt, plan, tree));
  if (psync && same_set && psync->ev && hipEventRecord(psync->ev, st) == hipSuccess) psync->post(true);
  MatchParams M;
  M.cs = P.center_scale, M.as = P.angular_scale;
  M.time_min = P.time_diff_min, M.ang_max = P.angular_scale, M.dist_max = P.surfel_dist_max;
  M.k = P.knn_k, M.same_set = same_set ? 1 : 0;
  // (margins of the early acceptance: 1e-9 in the angle and relative 1e-9 in the distance, against ~1e-15 between the walk's operands and the gates')
  M.cos_acc = M.ang_max > 1e-8 ? std::cos(std::min(M.ang_max, 3.141592653589793) - 1e-9) + 1e-12 : 2.0;
  M.plane_acc = M.dist_max * (1.0 - 1e-9) - 1e-12;
  // 3. exact k-NN + gates.  Queries are processed in the order of the tree's leaves, so that the lanes of a wavefront walk the same
  // nodes: same-set queries through the sorted target permutation, queries of another set (sliding window against fixed window)
  // by the leaf they would be looked for in first (in time order their walks are unrelated and the loads diverge)
  const uint32_t *qorder = tree.sorig;
  // (below ~40 k queries the two passes - locate, radix sort: ~70 us - cost more than the walks gain from them: 16 k queries against 4 k
  // targets 0.38 -> 0.31 ms in query order, 64 k the same, 250 k 1.23 -> 1.36; the rule depends on the call's sizes alone)
  // (round 6: NOT with the early bound - its walks are a descent and two or three leaves, and the two passes cost more than coherent
  // descents save even for queries in RANDOM order: 250 k queries against 62 k targets 0.74 -> 0.69 ms (own order 0.77 -> 0.70), C4's 1 M
  // against 50 k 1.07 -> 0.98 (0.95), room surfels 0.57 -> 0.56; profiles/dev/ab_sort_random.py, ab_room_match.py)
  const bool early_walk = !d_knn_idx && ctx->dev.knn_early != 0;
  const bool sort_queries = !same_set && (ctx->dev.knn_sort >= 0 ? ctx->dev.knn_sort != 0 : (nq >= 40000u && !early_walk));
  if (!same_set && !sort_queries) qorder = nullptr;
  if (sort_queries) {
    uint32_t *k0 = (uint32_t *)ctx->b_keys[0].p, *v0 = (uint32_t *)ctx->b_vals[0].p;
    uint32_t *qk = (uint32_t *)b_choice.p + 2 * (size_t)nq, *qo = (uint32_t *)b_choice.p + 3 * (size_t)nq;  // (flags / offsets: free until step 5)
    k_tree_locate<<<(nq + 255) / 256, 256, 0, st>>>(d_q_surf, d_q_pose, nq, tree, M.cs, M.as, k0, v0);
    // (the Onesweep radix path at every size: rocPRIM's default below 2^20 items is a merge sort of ~19 launches)
    using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;
    const unsigned end_bit = (unsigned)std::max(1, plan.D);
    size_t tmp = 0;
    WC_HIP(ctx, rocprim::radix_sort_pairs<cfg>(nullptr, tmp, k0, qk, v0, qo, (size_t)nq, 0u, end_bit, st));
    WC_TRY(wc_ensure(ctx, ctx->b_sorttmp, tmp));
    tmp = ctx->b_sorttmp.cap;
    WC_HIP(ctx, rocprim::radix_sort_pairs<cfg>(ctx->b_sorttmp.p, tmp, k0, qk, v0, qo, (size_t)nq, 0u, end_bit, st));
    qorder = qo;
  }
  // several GPUs: the queries are independent (knn_surfel_matcher.cc:22-48), the targets are replicated;
  // every rank searches a contiguous share of the queries (in leaf order) and ONE all-gather of the gated lists (4 k bytes per
  // query) gives every rank the whole table; the order-dependent de-duplication below then runs replicated
  // (`sharded` is decided at the top, from the replicated arguments alone)
  uint32_t q_begin = 0, q_end = nq;
  if (sharded) {
    const uint32_t w = (uint32_t)ctx->comm.world, r = (uint32_t)ctx->comm.rank;
    q_begin = (uint32_t)(((uint64_t)nq * r) / w), q_end = (uint32_t)(((uint64_t)nq * (r + 1)) / w);
  }
  // the searches' output: position-major lists (this rank's share [q_begin, q_end) of the positions when sharded); all ranks'
  // lists - or simply this call's - then sit in b_route[2] and are transposed into the planes of b_gated
  WC_TRY(wc_ensure(ctx, ctx->b_route[2], (size_t)nq * P.knn_k * 4));
  WC_TRY(wc_ensure(ctx, ctx->b_route[3], (size_t)(q_end - q_begin + 1) * P.knn_k * 4));
  // (a search of few queries writes the planes of b_gated straight from the walk - ten scattered 4-byte stores per query, what the
  // transposition below exists to avoid at a million queries, are nothing at 100 k, and two launches of the call's tail go)
  const bool direct_planes = !sharded && (nq < 131072u || (early_walk && ctx->dev.knn_group != 0 && (!same_set || ctx->dev.knn_early == 1)));  // (the group walk stores what k_resolve can read, no more)
  uint32_t *gated_shard = sharded ? (uint32_t *)ctx->b_route[3].p : (direct_planes ? nullptr : (uint32_t *)ctx->b_route[2].p);
  const uint32_t nq_mine = q_end - q_begin;
  for (hipEvent_t &e : ctx->ev_knn)
    if (!e) WC_HIP(ctx, hipEventCreate(&e));
  // pending nodes of a walk: the children of the root's step, then 2^kW - 1 more per further step above the leaves
  const int wide_steps = plan.D > 0 ? 1 + (plan.D - plan.first) / kW : 0, stack_cap = std::max(1, (1 << plan.first) + (kNch - 1) * std::max(0, wide_steps - 2));
  static const bool tdbg = wc_log_env("WC_MATCH_TIMING");
  const auto t_prep = std::chrono::steady_clock::now();
  if (tdbg) WC_HIP(ctx, hipEventRecord(ctx->ev_knn[0], st));
  // Which walk: eight lanes per query (match_tree.inc: k_knn_tree_group; every round trip coalesced, eight independent walks per
  // wavefront, VALU-bound at ~87 % busy) or one lane per query (k_knn_tree).  The two searches of a step-like window at 16 k / 64 k /
  // 128 k / 250 k / 500 k queries, lane-per-query against group walk: same-set 0.59 / 0.77 / 0.88 / 1.18 / 2.09 against 0.33 / 0.48 /
  // 0.62 / 0.94 / 1.77 ms, fixed-window 0.62 / 0.96 / 1.13 / 1.35 / 2.28 against 0.31 / 0.54 / 0.80 / 1.23 / 2.15 ms; at a million
  // queries (C4) 2.06 against 2.02 and 3.50 against 3.96 (profiles/dev/time_match_sizes.py, time_match.py).  The rule depends on the
  // call's sizes and kind alone - no timing, no history.
  if (psync && !same_set && psync->wait()) WC_HIP(ctx, hipStreamWaitEvent(st, psync->ev, 0));  // (wc_pair_sync: behind the other search's build)
  const int group_opt = ctx->dev.knn_group;  // (development option: 0 / 1 pins the walk)
  // (round 6: with the early bound the group walk at every size - its short walks are a descent and two or three leaves, what the lane
  // walk's fp32 first look and four-wide steps were built to shorten is gone: C4's fixed-window search, 1 M queries, 0.95 -> 0.80 ms)
  const bool group_walk = group_opt >= 0 ? group_opt != 0 : (early_walk || nq_mine < 750000u || (same_set && nq_mine < 1500000u));
  const int first3 = plan.D % 3 ? plan.D % 3 : 3;
  // Workgroups of ONE wavefront (eight queries): the groups of a workgroup share nothing, and a 256-thread workgroup holds its four
  // wavefront slots and its LDS until the slowest of its 32 walks has ended - the next workgroup waits for all of them.  Measured
  // (profiles/dev/time_match.py pair, time_room_match.py): the step-like pair of searches 1.674 -> 1.605 ms, a room search 0.444 -> 0.439.
  constexpr int kGroupNT = 64;
  // two sets, nobody asked for the neighbour lists: the walk may stop at the nearest gate-passing candidate (match_tree.inc, EARLY)
  const bool early = !d_knn_idx && ctx->dev.knn_early != 0 && (!same_set || ctx->dev.knn_early != 2);
#define WC_KNN_LAUNCH(KK)                                                                                                                            \
  if (nq_mine && group_walk && early)                                                                                                                \
    k_knn_tree_group<KK, kGroupNT, true><<<(nq_mine + kGroupNT / 8 - 1) / (kGroupNT / 8), kGroupNT, 0, st>>>(                                        \
        d_q_surf, d_q_pose, nq, tree, first3, (const double *)b_world.p, nt, M, (uint32_t *)b_gated.p, d_knn_idx, d_knn_d2, qorder, q_begin, q_end,  \
        gated_shard, stats, status);                                                                                                                 \
  else if (nq_mine && group_walk)                                                                                                                    \
    k_knn_tree_group<KK, kGroupNT, false><<<(nq_mine + kGroupNT / 8 - 1) / (kGroupNT / 8), kGroupNT, 0, st>>>(                                       \
        d_q_surf, d_q_pose, nq, tree, first3, (const double *)b_world.p, nt, M, (uint32_t *)b_gated.p, d_knn_idx, d_knn_d2, qorder, q_begin, q_end,  \
        gated_shard, stats, status);                                                                                                                 \
  else if (nq_mine && early)                                                                                                                         \
    k_knn_tree<KK, true><<<(nq_mine + 63) / 64, 64, (size_t)(kNch + 1 + stack_cap) * 64 * 4, st>>>(d_q_surf, d_q_pose, nq, tree, (const double *)b_world.p, nt, M, (uint32_t *)b_gated.p, d_knn_idx, \
                                                       d_knn_d2, qorder, q_begin, q_end, gated_shard, stats, status, stack_cap);                     \
  else if (nq_mine)                                                                                                                                  \
    k_knn_tree<KK, false><<<(nq_mine + 63) / 64, 64, (size_t)(kNch + 1 + stack_cap) * 64 * 4, st>>>(d_q_surf, d_q_pose, nq, tree, (const double *)b_world.p, nt, M, (uint32_t *)b_gated.p, d_knn_idx, \
                                                       d_knn_d2, qorder, q_begin, q_end, gated_shard, stats, status, stack_cap);
  switch (P.knn_k) {  // the reference's k = 10 gets its own instantiation (top-k in 30 registers)
    case 10: WC_KNN_LAUNCH(10); break;
    case 1: WC_KNN_LAUNCH(1); break;
    case 2: WC_KNN_LAUNCH(2); break;
    case 3: WC_KNN_LAUNCH(3); break;
    case 4: WC_KNN_LAUNCH(4); break;
    case 5: WC_KNN_LAUNCH(5); break;
    case 6: WC_KNN_LAUNCH(6); break;
    case 7: WC_KNN_LAUNCH(7); break;
    case 8: WC_KNN_LAUNCH(8); break;
    case 9: WC_KNN_LAUNCH(9); break;
    case 11: WC_KNN_LAUNCH(11); break;
    case 12: WC_KNN_LAUNCH(12); break;
    case 13: WC_KNN_LAUNCH(13); break;
    case 14: WC_KNN_LAUNCH(14); break;
    case 15: WC_KNN_LAUNCH(15); break;
    default: WC_KNN_LAUNCH(16); break;
  }
#undef WC_KNN_LAUNCH
  const auto t_launched = std::chrono::steady_clock::now();
  WC_HIP(ctx, hipGetLastError());
  if (tdbg) WC_HIP(ctx, hipEventRecord(ctx->ev_knn[1], st));
  if (sharded) {
    const int w = ctx->comm.world;
    std::vector<uint64_t> bytes((size_t)w);
    for (int r = 0; r < w; ++r)
      bytes[r] = (uint64_t)(((uint64_t)nq * (r + 1)) / w - ((uint64_t)nq * r) / w) * P.knn_k * 4;
    if (!ctx->comm.stream_ordered) WC_HIP(ctx, hipStreamSynchronize(st));
    peer.armed = false;  // this rank enters the collective itself
    if (ctx->comm.allgatherv(ctx->comm.user, gated_shard, (uint64_t)nq_mine * P.knn_k * 4, ctx->b_route[2].p, bytes.data()) != 0)
      return wc_fail(ctx, WC_ERR_HIP, "wc_match: all-gather of the gated neighbour lists failed");
    // a peer that failed before the collective sent poison (MatchPeerGuard): first word of every rank's share
    std::vector<uint32_t> firsts((size_t)w);
    for (int r = 0; r < w; ++r) firsts[r] = (uint32_t)((((uint64_t)nq * r) / w) * P.knn_k);
    uint32_t *d_first = status + 56;  // (words 56 .. 63 of the control block: free; worlds of up to eight ranks per node)
    WC_HIP(ctx, hipMemcpyAsync(d_first, firsts.data(), (size_t)w * 4, hipMemcpyHostToDevice, st));  // (w <= 8: checked at the top)
    WC_HIP(ctx, hipStreamSynchronize(st));  // (`firsts` is a stack-lifetime vector)
    k_peer_poison<<<1, 64, 0, st>>>((const uint32_t *)ctx->b_route[2].p, d_first, w, status);
  }
  if (!direct_planes) {
    uint32_t *qpos = (uint32_t *)b_choice.p + nq;  // (choice[1]: free until the resolve rounds)
    k_inv_perm<<<(nq + 255) / 256, 256, 0, st>>>(qorder, nq, qpos);
    k_gated_planes<<<(nq + 255) / 256, 256, 0, st>>>((const uint32_t *)ctx->b_route[2].p, qpos, nq, P.knn_k, (uint32_t *)b_gated.p);
    WC_HIP(ctx, hipGetLastError());
  }
  // 4. resolve the order-dependent "pair already seen" rule by fixed-point iteration
  uint32_t *choice[2] = {(uint32_t *)b_choice.p, (uint32_t *)b_choice.p + nq};
  uint32_t *offsets = (uint32_t *)b_choice.p + 3 * (size_t)nq;
  int cur = 0;
  // rounds are issued eight at a time between host checks (a round past the fixed point changes nothing, so the extra ones
  // are harmless); round r of a batch reports into changed[r] and only the last word is read back
  // The compaction (step 5) is enqueued right behind every batch, before the host knows whether the batch reached the fixed point: the
  // rule - it did - then costs ONE host round trip for rounds + compaction instead of two (a batch that did not is followed by another
  // one, and the compaction is redone on its result).
  bool converged = false;
  // (the control block comes back in ONE copy, into PINNED memory - words 64.. of h_status -: into a stack array a copy is staged by the
  // runtime and the call returns when it is through; the launches behind it were enqueued 15 - 20 us late, a kernel trace showed)
  volatile uint32_t *h_ctl = ctx->h_status + 64;
  volatile uint32_t *hc8 = h_ctl + 32;
  volatile unsigned long long *h_stats = (volatile unsigned long long *)(h_ctl + 40);
  static const bool match_dbg = wc_log_env("WC_MATCH_DEBUG");
  {
    size_t tmp = 0;
    WC_HIP(ctx, rocprim::exclusive_scan(nullptr, tmp, rocprim::make_transform_iterator((const uint32_t *)choice[0], ChoiceFlag{}), offsets, 0u, (size_t)nq,
                                        rocprim::plus<uint32_t>(), st));
    WC_TRY(wc_ensure(ctx, b_scan, tmp + 16));
  }
  for (int batch = 0; batch < 250000 && !converged; ++batch) {
    const int rounds = same_set ? 8 : 1;
    if (batch > 0) WC_HIP(ctx, hipMemsetAsync(changed, 0, 32, st));  // (batch 0: cleared with the control block)
    for (int r = 0; r < rounds; ++r) {
      k_resolve<<<(nq + 255) / 256, 256, 0, st>>>((const uint32_t *)b_gated.p, nq, P.knn_k, same_set, batch == 0 && r == 0 ? 1 : 0, choice[cur],
                                                  choice[cur ^ 1], changed + r);
      cur ^= 1;
    }
    // 5. compact in query order
    if (batch > 0) WC_HIP(ctx, hipMemsetAsync(status, 0, 4, st));  // (the count of the previous, unconverged, attempt)
    {
      size_t tmp = b_scan.cap;
      WC_HIP(ctx, rocprim::exclusive_scan(b_scan.p, tmp, rocprim::make_transform_iterator((const uint32_t *)choice[cur], ChoiceFlag{}), offsets, 0u, (size_t)nq,
                                          rocprim::plus<uint32_t>(), st));
    }
    k_emit_pairs<<<(nq + 255) / 256, 256, 0, st>>>(choice[cur], offsets, nq, d_q_surf, (const double *)b_world.p, same_set, d_pairs, cap, status);
    WC_HIP(ctx, hipGetLastError());
    WC_HIP(ctx, hipMemcpyAsync((void *)h_ctl, status, 56 * 4, hipMemcpyDeviceToHost, st));
    WC_HIP(ctx, hipStreamSynchronize(st));
    const uint32_t hc = hc8[rounds - 1];
    if (match_dbg)
      fprintf(stderr, "[match] resolve batch %d: rounds that changed something %u%u%u%u%u%u%u%u\n", batch, hc8[0], hc8[1], hc8[2], hc8[3], hc8[4], hc8[5],
              hc8[6], hc8[7]);
    converged = !hc || !same_set;
  }
  if (!converged) return wc_fail(ctx, WC_ERR_NUMERIC, "wc_match: the pair de-duplication did not reach its fixed point");
  for (int j = 0; j < 8; ++j) ctx->match_stats[j] = (double)h_stats[j];
  ctx->match_stats[5] = (double)plan.D, ctx->match_stats[6] = (double)plan.T, ctx->match_stats[7] = (double)nt;
  if (match_dbg && h_stats[4])
    fprintf(stderr, "[match] nq %u nt %u same %d: tree depth %d (%d top levels in %d stages), per query %.1f wide nodes, %.1f leaves, %.1f points, %.1f exact distances\n",
            nq, nt, same_set, plan.D, plan.T, plan.nstage, (double)h_stats[0] / h_stats[4], (double)h_stats[1] / h_stats[4], (double)h_stats[2] / h_stats[4],
            (double)h_stats[3] / h_stats[4]);
  if (tdbg) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ctx->ev_knn[0], ctx->ev_knn[1]);
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    const auto t_end = std::chrono::steady_clock::now();
    fprintf(stderr, "[match] same %d nq %u nt %u: host enqueue of the build %.0f us, launch %.0f us, launch -> done %.0f us (k_knn_tree by events %.0f us)\n",
            same_set, nq, nt, us(t_entry, t_prep), us(t_prep, t_launched), us(t_launched, t_end), ms * 1e3);
  }
  const uint32_t n_found = h_ctl[0], fl = h_ctl[1];
  *h_n_pairs = n_found;
  if (fl & 16u) return wc_fail(ctx, WC_ERR_HIP, "wc_match_sharded: another rank failed before the all-gather of the neighbour lists");
  if (fl & 4u) return wc_fail(ctx, WC_ERR_ARG, "non-finite surfel centre or normal");
  if (fl & 8u) return wc_fail(ctx, WC_ERR_NUMERIC, "wc_match: traversal stack overflow (internal)");
  if (fl & 2u) return wc_fail(ctx, WC_ERR_ORDER, "fixed-window surfel newer than its sliding-window match");
  if (n_found > cap) return wc_fail(ctx, WC_ERR_CAPACITY, "pair capacity %llu < %u", (unsigned long long)cap, n_found);
  return WC_OK;
}

extern "C" int wc_match(wc_ctx *ctx, const wc_surfel *d_q_surf, const wc_pose *d_q_pose, uint64_t nq, const wc_surfel *d_t_surf,
                        const wc_pose *d_t_pose, uint64_t nt, int same_set, wc_pair *d_pairs, uint64_t cap, uint64_t *h_n_pairs,
                        uint32_t *d_knn_idx, double *d_knn_d2) {
  return match_impl(ctx, d_q_surf, d_q_pose, nq, d_t_surf, d_t_pose, nt, same_set, d_pairs, cap, h_n_pairs, d_knn_idx, d_knn_d2, false);
}

extern "C" int wc_match_stats(wc_ctx *ctx, double h_out[8]) {
  if (!ctx || !h_out) return wc_fail(ctx, WC_ERR_ARG, "%s: null argument", __func__);
  for (int j = 0; j < 8; ++j) h_out[j] = ctx->match_stats[j];
  return WC_OK;
}

extern "C" int wc_match_sharded(wc_ctx *ctx, const wc_surfel *d_q_surf, const wc_pose *d_q_pose, uint64_t nq, const wc_surfel *d_t_surf,
                                const wc_pose *d_t_pose, uint64_t nt, int same_set, wc_pair *d_pairs, uint64_t cap, uint64_t *h_n_pairs) {
  return match_impl(ctx, d_q_surf, d_q_pose, nq, d_t_surf, d_t_pose, nt, same_set, d_pairs, cap, h_n_pairs, nullptr, nullptr, true);
}

// both searches of an outer iteration as collectives, one after the other (their all-gathers share the ctx stream)
extern "C" int wc_match_pair_sharded(wc_ctx *ctx, const wc_surfel *d_sld_surf, const wc_pose *d_sld_pose, uint64_t n_sld,
                                     const wc_surfel *d_fix_surf, const wc_pose *d_fix_pose, uint64_t n_fix, wc_pair *d_pairs_sld,
                                     uint64_t cap_sld, uint64_t *h_n_pairs_sld, wc_pair *d_pairs_fix, uint64_t cap_fix,
                                     uint64_t *h_n_pairs_fix) {
  if (!ctx || !h_n_pairs_sld || !h_n_pairs_fix) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  WC_TRY(wc_match_sharded(ctx, d_sld_surf, d_sld_pose, n_sld, d_sld_surf, d_sld_pose, n_sld, 1, d_pairs_sld, cap_sld, h_n_pairs_sld));
  return wc_match_sharded(ctx, d_sld_surf, d_sld_pose, n_sld, d_fix_surf, d_fix_pose, n_fix, 0, d_pairs_fix, cap_fix, h_n_pairs_fix);
}

// The helper context (second stream, own scratch) and host thread of wc_match_pair, created at the first call - or ahead of time by
// wc_ctx_warmup: created inside a sweep it showed as a 19 ms search in the facade's stream (round 5).
int wc_match_pair_prepare(wc_ctx *ctx) {
  wc_dev_guard dg_(ctx);
  if (!ctx->aux) {
    const int rc = wc_ctx_create(&ctx->P, ctx->device, &ctx->aux);
    if (rc != WC_OK) return wc_fail(ctx, rc, "wc_match_pair: no helper context");
    ctx->aux->dev = ctx->dev;
    // The helper's stream gets the device's highest priority: its search is prepared (tree, locate, radix passes - small launches)
    // while the other search's walk fills the chip, and those launches wait for slots behind the walk's workgroups.  What a kernel
    // trace of the odometry step shows with it: launches of 256- and 512-thread workgroups get through (k_kd_bottom 77 us next to the
    // other tree's 127), a launch of 1 024-thread workgroups does not - a CU whose slots are refilled four wavefronts at a time never
    // has sixteen free (one Onesweep pass: 500 us with or without priority).  Measured together with the order of the two
    // k_kd_bottom launches (kd_build): the step's two searches 1.82 -> 1.76 - 1.79 ms.
    int pr_lo = 0, pr_hi = 0;
    hipStream_t hs = nullptr;
    if (hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi) == hipSuccess && pr_hi < pr_lo &&
        hipStreamCreateWithPriority(&hs, hipStreamNonBlocking, pr_hi) == hipSuccess) {
      (void)hipStreamDestroy(ctx->aux->own_stream);
      ctx->aux->own_stream = hs, ctx->aux->stream = hs;
    }
  }
  if (!ctx->pair_worker) {
    try {
      ctx->pair_worker = new wc_pair_worker();
      ctx->pair_worker_free = [](void *p) { delete (wc_pair_worker *)p; };
    } catch (...) {
      ctx->pair_worker = nullptr;  // (the second search then runs on the caller's thread)
    }
  }
  return WC_OK;
}

// The two searches of an outer iteration side by side (see include/wildcat_hip.h).  wc_match is synchronous and talks to the
// host between its launches (the fixed-point rounds of the pair rule), so the second search gets its own context AND its own
// host thread; both only read the surfels.
extern "C" int wc_match_pair(wc_ctx *ctx, const wc_surfel *d_sld_surf, const wc_pose *d_sld_pose, uint64_t n_sld,
                             const wc_surfel *d_fix_surf, const wc_pose *d_fix_pose, uint64_t n_fix, wc_pair *d_pairs_sld,
                             uint64_t cap_sld, uint64_t *h_n_pairs_sld, wc_pair *d_pairs_fix, uint64_t cap_fix,
                             uint64_t *h_n_pairs_fix) {
  if (!ctx || !h_n_pairs_sld || !h_n_pairs_fix) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  const bool serial = ctx->dev.match_pair_serial != 0;  // (development option)
  if (n_sld == 0 || n_fix == 0 || serial) {
    WC_TRY(wc_match(ctx, d_sld_surf, d_sld_pose, n_sld, d_sld_surf, d_sld_pose, n_sld, 1, d_pairs_sld, cap_sld, h_n_pairs_sld, nullptr, nullptr));
    return wc_match(ctx, d_sld_surf, d_sld_pose, n_sld, d_fix_surf, d_fix_pose, n_fix, 0, d_pairs_fix, cap_fix, h_n_pairs_fix, nullptr, nullptr);
  }
  wc_dev_guard dg_(ctx);
  WC_TRY(wc_match_pair_prepare(ctx));
  wc_ctx *aux = ctx->aux;
  WC_TRY(wc_ctx_set_params(aux, &ctx->P));
  // the helper's stream starts behind everything already enqueued on the ctx stream (the producers of the surfels and poses)
  if (!ctx->ev_aux) WC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_aux, hipEventDisableTiming));
  WC_HIP(ctx, hipEventRecord(ctx->ev_aux, ctx->stream));
  WC_HIP(ctx, hipStreamWaitEvent(aux->stream, ctx->ev_aux, 0));
  int rc_fix = WC_OK, rc_sld = WC_OK;
  wc_pair_sync psync;
  if (!ctx->ev_pair) WC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_pair, hipEventDisableTiming));
  psync.ev = ctx->ev_pair;
  struct PairScope {  // both searches see the call's wc_pair_sync, and neither a later one
    wc_ctx *a, *b;
    PairScope(wc_ctx *a_, wc_ctx *b_, wc_pair_sync *s) : a(a_), b(b_) { a->pair_sync = s, b->pair_sync = s; }
    ~PairScope() { a->pair_sync = nullptr, b->pair_sync = nullptr; }
  } pair_scope(ctx, aux, &psync);
  // (A rendezvous of the two searches in front of their walk kernels was tried in round 3 - grid kernels: 2.50 - 2.76 ms against
  // 2.45 - 2.50 - and again in round 4 with the tree: a kernel trace showed the fixed-window search's locate + radix passes waiting
  // for wavefront slots behind the other search's k_knn_tree - one Onesweep pass of 250 k keys took 446 us - and its walk starting
  // when the other was nearly over; with the first walk held back until the second preparation is through both walks start together
  // at 0.39 ms, last 1.31 / 1.47 ms instead of 1.02 / 1.03, and the pair ends at the same 2.04 ms: the walks are bound by
  // wavefront-slot time (3 906 wavefronts each for 4 096 slots), not by when they start.  Not kept.)
  // no exception may cross the C boundary: a failed thread creation runs the second search on this thread, an allocation
  // failure inside a search becomes a status code
  auto guarded = [](int &rc, auto &&fn) {
    try {
      rc = fn();
    } catch (...) {
      rc = WC_ERR_HIP;
    }
  };
  // which search runs where: the one on the ctx stream starts at once, the helper's a thread start later (WC_MATCH_PAIR_SWAP: A/B)
  const bool swap = ctx->dev.match_pair_swap != 0;  // (development option)
  wc_ctx *c_fix = swap ? ctx : aux, *c_sld = swap ? aux : ctx;
  auto search_fix = [&] { return wc_match(c_fix, d_sld_surf, d_sld_pose, n_sld, d_fix_surf, d_fix_pose, n_fix, 0, d_pairs_fix, cap_fix, h_n_pairs_fix, nullptr, nullptr); };
  auto search_sld = [&] { return wc_match(c_sld, d_sld_surf, d_sld_pose, n_sld, d_sld_surf, d_sld_pose, n_sld, 1, d_pairs_sld, cap_sld, h_n_pairs_sld, nullptr, nullptr); };
  wc_pair_worker *worker = (wc_pair_worker *)ctx->pair_worker;  // (wc_match_pair_prepare; null: no thread could be started)
  bool threaded = worker != nullptr;
  if (threaded) {
    try {
      worker->start([&] {
        if (swap)
          guarded(rc_sld, search_sld);
        else
          guarded(rc_fix, search_fix);
      });
    } catch (...) {
      threaded = false;
    }
  }
  if (!threaded || ctx->dev.match_pair_hold == 0) psync.post(false);  // (one thread, one search after the other: nothing to hold back, and nobody to wait for)
  if (swap)
    guarded(rc_fix, search_fix);
  else
    guarded(rc_sld, search_sld);
  if (threaded)
    worker->wait();
  else if (swap)
    guarded(rc_sld, search_sld);
  else
    guarded(rc_fix, search_fix);
  if (rc_sld != WC_OK) return c_sld == ctx ? rc_sld : wc_fail(ctx, rc_sld, "wc_match_pair (sliding window): %s", wc_last_error(aux));
  if (rc_fix != WC_OK) return c_fix == ctx ? rc_fix : wc_fail(ctx, rc_fix, "wc_match_pair (fixed window): %s", wc_last_error(aux));
  return WC_OK;
}


int wc_touch_match() {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, (const void *)k_inv_perm) == hipSuccess ? WC_OK : WC_ERR_HIP;
}
