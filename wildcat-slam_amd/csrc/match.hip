// match.hip — surfel-to-surfel correspondence on gfx950.  Replaces KnnSurfelMatcher::BuildIndex / Match
// (src/odometry/knn_surfel_matcher.cc:3-49) and the FLANN kd-tree beneath it (cc:65-89):
//   * 6-D feature [centre_world / 1.0, normal_world / 5deg] per target (ToVector, cc:91-98)
//   * EXACT k = 10 nearest neighbours in squared L2 (FLANN SearchParams(-1, 0.0): exact, sorted), ties by index
//   * first neighbour passing |dt| >= 0.06, acos(n.n) <= 5deg, |n.(c - c')| <= 0.1 and "pair not seen yet" (cc:25-47)
// Index: targets sorted by the 1-unit cell of their (scaled) centre, x fastest, so that a run of cells along x is one
// contiguous range found by two binary searches; the query expands a cube of cells shell by shell and stops as soon as
// the k-th best distance is inside the searched cube (the 6-D distance is bounded below by the 3-D centre distance),
// which keeps the search exact.  The reference's order dependence (std::set of already-paired surfels, queries visited
// in order) is a recurrence choice(q) = f(choice(c) : c < q); it is solved by fixed-point iteration on the device.
// Memory bound gather/scan work; no MFMA.
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

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;

struct MatchParams {
  double cs, as;        // centre / angular scale
  double time_min, ang_max, dist_max;
  int k;
  double h;             // cell size in scaled units
  double org[3];        // grid origin (scaled units)
  int dim[3];           // cells per axis (<= 1024)
  const uint32_t *cell_start;  // dense per-cell [start, end) into the sorted targets, or null (binary search fallback)

};

__device__ __forceinline__ void feature6(const wc_surfel &s, const wc_pose &p, double cs, double as, double f[6], V3 &cw, V3 &nw) {
  const Q4 q{p.quat[0], p.quat[1], p.quat[2], p.quat[3]};
  cw = qrot(q, mk3(s.center[0], s.center[1], s.center[2])) + mk3(p.pos[0], p.pos[1], p.pos[2]);  // surfel.h:67-69
  nw = qrot(q, mk3(s.normal[0], s.normal[1], s.normal[2]));                                      // surfel.h:78-80
  f[0] = cw.x / cs, f[1] = cw.y / cs, f[2] = cw.z / cs;
  f[3] = nw.x / as, f[4] = nw.y / as, f[5] = nw.z / as;
}

__device__ __forceinline__ unsigned long long enc_min(double v) {  // order-preserving encoding for atomicMin/Max
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
inline double dec_host(unsigned long long u) {
  u = (u >> 63) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u;
  double d;
  memcpy(&d, &u, 8);
  return d;
}

__global__ void __launch_bounds__(256) k_features(const wc_surfel *surf, const wc_pose *pose, uint32_t n, double cs, double as,
                                                 double *feat, double *world, unsigned long long *bbox) {
  __shared__ unsigned long long s_mm[4][6];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0ull, 0ull, 0ull};
  if (i < n) {
    double f[6];
    V3 cw, nw;
    feature6(surf[i], pose[i], cs, as, f, cw, nw);
    for (int d = 0; d < 6; ++d) feat[(size_t)i * 6 + d] = f[d];
    double *w = world + (size_t)i * 7;
    w[0] = cw.x, w[1] = cw.y, w[2] = cw.z, w[3] = nw.x, w[4] = nw.y, w[5] = nw.z, w[6] = surf[i].t;
    for (int d = 0; d < 3; ++d) lo[d] = hi[d] = enc_min(f[d]);
  }
  // bounding box: wavefront, then workgroup, then six atomics per workgroup (one per surfel on six words serialises:
  // 0.58 ms per million surfels)
  for (int d = 0; d < 3; ++d)
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned long long a = __shfl_xor(lo[d], off), b = __shfl_xor(hi[d], off);
      lo[d] = min(lo[d], a);
      hi[d] = max(hi[d], b);
    }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
    for (int d = 0; d < 3; ++d) s_mm[w][d] = lo[d], s_mm[w][3 + d] = hi[d];
  __syncthreads();
  if (threadIdx.x < 3)
    atomicMin(&bbox[threadIdx.x], min(min(s_mm[0][threadIdx.x], s_mm[1][threadIdx.x]), min(s_mm[2][threadIdx.x], s_mm[3][threadIdx.x])));
  else if (threadIdx.x < 6)
    atomicMax(&bbox[threadIdx.x], max(max(s_mm[0][threadIdx.x], s_mm[1][threadIdx.x]), max(s_mm[2][threadIdx.x], s_mm[3][threadIdx.x])));
}

// ---- the cell size of THIS call's data: a sampled look at the k-th 6-D distances ----------------------------------------------
// The grid lives on the scaled centres only, so a query has to scan every cell within its k-th 6-D distance; the right cell size
// is that distance, which depends on the window (1.4 m cells suit a sparse fixed window whose 10th neighbour is metres away, a
// room seen by 40 sweeps has 280 surfels per cubic metre and its 10th neighbour 0.4 units away - one-metre cells there mean
// 2 500 candidates per query).  kSampleQ queries, spread over the call's queries, count the targets inside 16 radii 2^(i/2 - 3)
// (0.125 .. 22.6 units) - every target against every sample, a few GFLOP at most (large target sets are strided) - and the host
// takes the median radius that holds k of them.  Only the speed of the search depends on it, never its result.
constexpr int kSampleQ = 128, kSampleR = 16;
__global__ void __launch_bounds__(128) k_sample_feats(const wc_surfel *surf, const wc_pose *pose, uint32_t n, double cs, double as, double *sf,
                                                     uint32_t *counts) {
  const int s = threadIdx.x;
  const uint32_t i = (uint32_t)(((uint64_t)n * (2u * (uint32_t)s + 1u)) / (2u * (uint32_t)kSampleQ));  // evenly spread over the call's queries
  double f[6];
  V3 cw, nw;
  feature6(surf[min(i, n - 1u)], pose[min(i, n - 1u)], cs, as, f, cw, nw);
  for (int d = 0; d < 6; ++d) sf[s * 6 + d] = f[d];
  for (int r = 0; r < kSampleR; ++r) counts[s * kSampleR + r] = 0u;
}
__global__ void __launch_bounds__(256) k_kth_sample(const double *__restrict__ feat, uint32_t nt, uint32_t stride, const double *__restrict__ sf,
                                                   uint32_t *counts) {
  __shared__ double s_f[kSampleQ][6];
  __shared__ uint32_t s_cnt[kSampleQ][kSampleR];
  for (int e = threadIdx.x; e < kSampleQ * 6; e += 256) (&s_f[0][0])[e] = sf[e];
  for (int e = threadIdx.x; e < kSampleQ * kSampleR; e += 256) (&s_cnt[0][0])[e] = 0u;
  __syncthreads();
  const uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * stride;
  if (i < nt) {
    double t[6];
    for (int d = 0; d < 6; ++d) t[d] = feat[i * 6 + d];
    for (int s = 0; s < kSampleQ; ++s) {
      double d2 = 0.0;
#pragma unroll
      for (int d = 0; d < 6; ++d) {
        const double v = t[d] - s_f[s][d];
        d2 = fma(v, v, d2);
      }
      if (d2 <= 512.0) {  // (rare: most targets are far from most samples)
        int e = 0;
        (void)frexp(d2, &e);  // d2 in [2^(e-1), 2^e): inside the radius with r^2 = 2^e
        const int b = min(max(e + 6, 0), kSampleR - 1);
        atomicAdd(&s_cnt[s][b], 1u);
      }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < kSampleQ * kSampleR; e += 256) {
    const uint32_t c = (&s_cnt[0][0])[e];
    if (c) atomicAdd(&counts[e], c);
  }
}

__device__ __forceinline__ int cell_of(double v, double org, double h, int dim) {
  int c = (int)floor((v - org) / h);
  return min(max(c, 0), dim - 1);
}

__global__ void __launch_bounds__(256) k_cell_keys(const double *feat, uint32_t n, MatchParams M, uint32_t *keys, uint32_t *vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int cx = cell_of(feat[(size_t)i * 6 + 0], M.org[0], M.h, M.dim[0]);
  const int cy = cell_of(feat[(size_t)i * 6 + 1], M.org[1], M.h, M.dim[1]);
  const int cz = cell_of(feat[(size_t)i * 6 + 2], M.org[2], M.h, M.dim[2]);
  keys[i] = (uint32_t)cx | ((uint32_t)cy << 10) | ((uint32_t)cz << 20);
  vals[i] = i;
}

// queries of a match against ANOTHER set: their cell in the target grid (clamped), so that they can be processed in cell
// order like the same-set queries - in time order their neighbourhoods are unrelated and k_knn_gate spends ten times as long
// in divergent gathers (14 ms instead of 1.4 ms per million queries)
__global__ void __launch_bounds__(256) k_query_keys(const wc_surfel *surf, const wc_pose *pose, uint32_t n, MatchParams M, uint32_t *keys,
                                                   uint32_t *vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double f[6];
  V3 cw, nw;
  feature6(surf[i], pose[i], M.cs, M.as, f, cw, nw);
  const int cx = cell_of(f[0], M.org[0], M.h, M.dim[0]), cy = cell_of(f[1], M.org[1], M.h, M.dim[1]), cz = cell_of(f[2], M.org[2], M.h, M.dim[2]);
  keys[i] = (uint32_t)cx | ((uint32_t)cy << 10) | ((uint32_t)cz << 20);
  vals[i] = i;
}

// dense cell table: cell_first[c] = number of sorted targets in cells < c (a lower bound for EVERY cell, empty ones
// included), so that a run of cells along x - contiguous in the sorted order - is ONE range [first[c0], first[c1 + 1]):
// two table reads per row of a shell instead of two per cell.  The table is pre-filled with n (cells behind the last
// target); the thread of the first target of a cell fills the gap since the previous non-empty cell.
__global__ void __launch_bounds__(256) k_cell_table(const uint32_t *skeys, uint32_t n, MatchParams M, uint32_t *cell_first) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = skeys[i];
  if (i > 0 && skeys[i - 1] == k) return;
  auto cell_of = [&](uint32_t key) { return (size_t)(key & 1023u) + (size_t)M.dim[0] * ((size_t)((key >> 10) & 1023u) + (size_t)M.dim[1] * (size_t)(key >> 20)); };
  const size_t c = cell_of(k);
  size_t c0 = (i == 0) ? 0 : cell_of(skeys[i - 1]) + 1;
  for (; c0 <= c; ++c0) cell_first[c0] = i;
}

// shalf: the two halves of every sorted feature once more in single precision, 16 bytes each (centre part [0, n), normal part
// [n, 2 n)): what k_knn_gate's first look at a candidate reads (see there)
__global__ void __launch_bounds__(256) k_sorted_feat(const double *feat, const uint32_t *sorted_idx, uint32_t n, double *sfeat, float4 *shalf) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t o = sorted_idx[i];
  double f[6];
  for (int d = 0; d < 6; ++d) f[d] = feat[(size_t)o * 6 + d];
  for (int d = 0; d < 6; ++d) sfeat[(size_t)i * 6 + d] = f[d];
  shalf[i] = make_float4((float)f[0], (float)f[1], (float)f[2], 0.f);
  shalf[(size_t)n + i] = make_float4((float)f[3], (float)f[4], (float)f[5], 0.f);
}

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *a, uint32_t n, uint32_t key) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a[mid] < key)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

#ifdef WC_PROF_KNN
__device__ unsigned long long g_knn_prof[4];  // candidates scanned, rows visited, shells, queries
#endif

template <int K>
struct TopK {
  double d[K];
  uint32_t id[K];
  int cnt;
  __device__ __forceinline__ double worst() const { return cnt < K ? 1e300 : d[K - 1]; }
  // sorted insertion with static indexing (keeps the arrays in registers); order = (distance, index)
  __device__ __forceinline__ void push(double dist, uint32_t idx) {
    if (cnt == K && !(dist < d[K - 1] || (dist == d[K - 1] && idx < id[K - 1]))) return;
    double cd = dist;
    uint32_t ci = idx;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const bool before = cd < d[i] || (cd == d[i] && ci < id[i]);  // empty slots hold (1e300, ~0): always "after"
      if (before) {
        const double td = d[i];
        const uint32_t ti = id[i];
        d[i] = cd, id[i] = ci;
        cd = td, ci = ti;
      }
    }
    if (cnt < K) ++cnt;
  }
};

constexpr int kKeep = 16;      // notes per lane (k_knn_gate)
constexpr int kKnnThreads = 64;  // k_knn_gate's workgroup: one wavefront (64 / 128 / 256 threads: 2.00 / 2.03 / 2.13 ms for the odometry step's searches)
constexpr int kRowChunk = 4;  // rows of a shell whose candidate ranges are looked up together (8: spills, no gain)

// exact k-NN + gates.  gated[j][q] (plane j of nq entries: the resolve rounds read plane 0 coalesced and rarely more) =
// j-th neighbour of q passing the first three gates (kNone-terminated).
template <int K, bool NF, bool F32 = false>
__global__ void __launch_bounds__(kKnnThreads, 4) k_knn_gate(const wc_surfel *q_surf, const wc_pose *q_pose, uint32_t nq, const double *sfeat,
                                                 const uint32_t *skeys, const uint32_t *sorig, const double *tworld, uint32_t nt,
                                                 MatchParams M, uint32_t *gated, uint32_t *knn_idx, double *knn_d2,
                                                 const uint32_t *__restrict__ qorder, uint32_t q_begin, uint32_t q_end, uint32_t *gated_shard,
                                                 double *kth_stat, uint32_t budget, uint32_t *defer, const float4 *__restrict__ shalf) {
  // F32 / shalf: the half of the sorted features the first look at a candidate tests (NF: the normal part), in SINGLE precision,
  // 16 bytes per target.  The first look only has to be conservative - whoever passes it is summed exactly, in fp64, by drain()
  // - so it compares the fp32 half-sum with the k-th distance plus a bound on its own rounding (thr32 below): one 16-byte load
  // per candidate instead of 24 bytes in two, a third of the cache lines per row of candidates.  Same lists, bit for bit.
  // Used for windows whose k-th neighbour is cells away (the call before this one on the context measured that: match_sparse):
  // the odometry step's two searches 2.65 -> 2.52 ms, a C4 window's 472 -> 500 M surfels/s.  On the facade's room stream - k-th
  // neighbour inside the query's own cells, long dense rows - the same first look is SLOWER (k_knn_gate 2.49 -> 2.67 ms normal
  // half first, 2.94 -> 4.10 centre first, alternating on one box by rocprofv3), so the fp64 look stays there.
  // budget / defer: a query that has looked at more than `budget` candidates when a shell ends without its k-th distance being
  // inside the searched cube gives up here and is put on the list defer[1..] (defer[0] = their number): k_knn_wave finishes it
  // with a whole wavefront.  One lane walking the 10 - 60 k candidates of a query whose 10th neighbour is metres away in a
  // room that holds 280 surfels per cubic metre WAS the kernel: 5 ms per 50 k queries, the other 63 lanes of its wavefront idle.
  // qorder: the queries in the order of their grid cell (same-set matching: the sorted target permutation).  Neighbouring
  // threads then scan the same cell ranges: their feature loads hit the same cache lines and their trip counts agree.
  __shared__ uint2 s_rng[2 * kRowChunk][kKnnThreads];  // per thread: the candidate ranges of a chunk of rows (only its own column)
  __shared__ uint32_t s_keep[kKeep][kKnnThreads];      // per thread: candidates noted for the next drain
  // query-sharded call (several GPUs): this rank takes the positions [q_begin, q_end) of the cell order and writes its gated
  // lists position-major into gated_shard (K words per position) - contiguous, so that ONE all-gather assembles all ranks'
  const uint32_t qi = q_begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (qi >= q_end) return;
  const uint32_t q = qorder ? qorder[qi] : qi;
  double f[6];
  V3 cq, nq_w;
  feature6(q_surf[q], q_pose[q], M.cs, M.as, f, cq, nq_w);
  const double tq = q_surf[q].t;
  TopK<K> top;
  top.cnt = 0;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    top.d[i] = 1e300;
    top.id[i] = 0xFFFFFFFFu;
  }
  // first look in fp32: a candidate that belongs to the list has |d_i| <= sqrt(w) in every component (w = the k-th distance), so
  // its components are at most |q_i| + sqrt(w) in magnitude, the fp32 images of the two operands are off by <= 2^-24 of their
  // magnitudes, each difference by <= delta = 2^-23 (2 qmax + sqrt(w)) including its own rounding, each square by
  // 2 sqrt(w) delta + delta^2, and the three-term fp32 sum by a few 2^-24 of itself: thr32 bounds all of that from above.
  // (the query's fp32 half and qmax are formed where they are used: three conversions per group of four candidates are cheaper
  // than five more live registers in a kernel that sits at its 128)
  // (sqrt(w) is bounded by (w + 1) / 2 instead of being taken: the bound is refreshed behind every drain, and a dense room drains
  // every few candidates - with an fp64 square root there the first look cost more than it saved, 2.4 -> 2.8 ms per room search)
  auto thr32_of = [&](double w) -> float {
    if (!(w < 1e30)) return __builtin_inff();
    const double qmax = fmax(fmax(fabs(f[NF ? 3 : 0]), fabs(f[NF ? 4 : 1])), fabs(f[NF ? 5 : 2]));
    const double sw = 0.5 * (w + 1.0), delta = 1.1920928955078125e-7 * (2.0 * qmax + sw) * 1.01;
    return __double2float_ru(w * (1.0 + 2e-6) + 3.0 * (2.0 * sw * delta + delta * delta) + 1e-30);
  };
  float thr32 = __builtin_inff();
  // query cell (unclamped, so that the distance bound stays valid for queries outside the target bbox)
  const double gx = (f[0] - M.org[0]) / M.h, gy = (f[1] - M.org[1]) / M.h, gz = (f[2] - M.org[2]) / M.h;
  const int cx = (int)floor(gx), cy = (int)floor(gy), cz = (int)floor(gz);
  // distance from the query to the nearest face of its own cell, in scaled units
  const double in_cell = fmin(fmin(fmin(gx - cx, cx + 1 - gx), fmin(gy - cy, cy + 1 - gy)), fmin(gz - cz, cz + 1 - gz)) * M.h;
  const int rmax = max(max(max(cx, M.dim[0] - 1 - cx), max(cy, M.dim[1] - 1 - cy)), max(cz, M.dim[2] - 1 - cz));

#ifdef WC_PROF_KNN
  unsigned long long pc_ = 0, pr_ = 0, ps_ = 0;
#endif
  // Candidates whose first half does not exceed the current k-th distance are only NOTED (their index, in LDS: up to kKeep per
  // lane); when some lane's notes are nearly full all lanes of the wavefront finish theirs together: all six components in
  // one round trip, the exact sum in flann::L2_Simple's order (plain running sum of squared differences, component by
  // component), insertion.  Finishing a candidate on the spot costs the WAVEFRONT the index load's round trip and a pass through
  // the insertion code whenever ANY of its 64 lanes has one - measured with clocks around the call: ~1.8 k clocks each, ~800
  // times per wavefront, most of the kernel.  The k-th distance used by the first look may be stale by a few notes: more notes,
  // same result.
  uint32_t bcnt = 0, scanned = 0;
  bool deferred = false;
  auto drain = [&]() {
    for (uint32_t j = 0; __ballot(j < bcnt); ++j) {
      if (j < bcnt) {
        const uint32_t ci = s_keep[j][threadIdx.x];
        const double *p = sfeat + (size_t)ci * 6;
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < 6; ++d) {
          const double df = f[d] - p[d];
          s += df * df;
        }
        if (!(s > top.worst())) top.push(s, sorig[ci]);
      }
    }
    bcnt = 0;
    if (F32) thr32 = thr32_of(top.worst());
  };
  for (int r = 0; r <= rmax; ++r) {
#ifdef WC_PROF_KNN
    ++ps_;
#endif
    // shell r of the cube of cells around the query: rows (dy, dz); full x-span on the faces |dy| = r or |dz| = r,
    // only the two end cells elsewhere
    for (int dz = -r; dz <= r && !deferred; ++dz)
      for (int dy0 = -r; dy0 <= r; dy0 += kRowChunk) {
        // (the budget is looked at between chunks of rows, not only between shells: the second shell of a dense room holds
        // ~11 k candidates, and a lane that had to finish it first made its whole wavefront wait)
        if (defer && scanned > budget && r > 0) {
          deferred = true;
          break;
        }
        // ---- the ranges of kRowChunk rows (two parts each off the faces), looked up TOGETHER: against a sparse target set
        // a query visits ~200 rows of ~4 candidates, and a dependent table lookup per row is a round trip to L2 per row
        uint32_t bb[2 * kRowChunk], ee[2 * kRowChunk];
        // Exact pruning against the current k-th distance: a target in a row is at least (dyd, dzd) cells away in y and
        // z, a target in cell x at least |x - cx| - 1 cells in x.  Rows and x-cells beyond the k-th distance are not
        // looked up (1e-9 relative slack and one spare cell: the bound is rounded differently from the distances it is
        // compared with; the k-th distance only shrinks while the chunk is scanned, so the test stays conservative).
        const double wq = top.worst() / (M.h * M.h);
        const int z = cz + dz;
        const double dzd = dz > 0 ? (double)(cz + dz) - gz : (dz < 0 ? gz - (double)(cz + dz + 1) : 0.0);
#pragma unroll
        for (int u = 0; u < kRowChunk; ++u) {
          const int dy = dy0 + u, y = cy + dy;
          bb[2 * u] = ee[2 * u] = bb[2 * u + 1] = ee[2 * u + 1] = 0u;
          if (dy > r || y < 0 || y >= M.dim[1] || z < 0 || z >= M.dim[2]) continue;
          const bool face = max(abs(dy), abs(dz)) == r;
          const int nparts = (face || r == 0) ? 1 : 2;
          const double dyd = dy > 0 ? (double)(cy + dy) - gy : (dy < 0 ? gy - (double)(cy + dy + 1) : 0.0);
          const double rem = wq - (dyd * dyd + dzd * dzd);
          if (rem < -1e-9 * wq) continue;
          const int xs = rem < 1e12 ? (int)sqrt(fmax(rem, 0.0)) + 2 : (1 << 20);
          if (!face && r > xs) continue;
          const size_t row = (size_t)M.dim[0] * ((size_t)y + (size_t)M.dim[1] * (size_t)z);
#pragma unroll
          for (int part = 0; part < 2; ++part) {
            if (part >= nparts) continue;
            int x0 = face ? max(cx - r, cx - xs) : (part == 0 ? cx - r : cx + r);
            int x1 = face ? min(cx + r, cx + xs) : x0;
            x0 = max(x0, 0);
            x1 = min(x1, M.dim[0] - 1);
            if (x0 > x1) continue;
            if (M.cell_start) {  // the x-run of cells is one contiguous range of sorted targets
              bb[2 * u + part] = M.cell_start[row + x0];
              ee[2 * u + part] = M.cell_start[row + x1 + 1];
            } else {  // binary-search fallback
              const uint32_t base = ((uint32_t)y << 10) | ((uint32_t)z << 20);
              bb[2 * u + part] = lower_bound_u32(skeys, nt, base | (uint32_t)x0);
              ee[2 * u + part] = lower_bound_u32(skeys, nt, (base | (uint32_t)x1) + 1u);
            }
          }
        }
#pragma unroll
        for (int sl = 0; sl < 2 * kRowChunk; ++sl) s_rng[sl][threadIdx.x] = make_uint2(bb[sl], ee[sl]);
        // ---- ONE scan site for all ranges (the kernel stays small enough to keep the top-k in registers)
        for (int sl = 0; sl < 2 * kRowChunk; ++sl) {
          const uint2 be = s_rng[sl][threadIdx.x];
          const uint32_t b = be.x, e = be.y;
#ifdef WC_PROF_KNN
          pc_ += e - b, pr_ += (e > b);
#endif
          scanned += e - b;
          // Candidates in groups of four: ONE half of the six components of a group is requested together (one candidate per
          // trip of a load - test loop costs a full round trip to L2 each).  Which half is decided per CALL from the k-th
          // distances of the previous call on this context (NF): the centre part while the k-th distance is small against the
          // cells (targets whose neighbours share their normal: most of a cell's candidates lie outside the sphere), the
          // normal part when it is not (normals that differ: 5 degrees are one unit, the 6-D k-th distance is several cells
          // and the centre part of nearly every candidate lies below it).  A half that exceeds the k-th distance on its own
          // bounds the full sum from below in floating point too (adding non-negative terms is monotone).
          // (tried on top: the candidates of ALL ranges of a chunk as one stream, a trip filling up from the next non-empty range -
          // 188 bytes of scratch and two more loops per trip: 2.45 -> 3.9 ms for the odometry step's searches.  Range by range.)
          // (a range's last, partial group is ONE trip too - the lanes' clamped loads repeat the range's last candidate -, not
          // one trip per candidate: against a sparse target set the rows hold ~5 candidates, and most trips were such leftovers)
          if (F32) {
            for (uint32_t i = b; i < e; i += 4) {
              float4 v4[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) v4[u] = shalf[min(i + (uint32_t)u, e - 1u)];
              float h4[4];
              const float q0 = (float)f[NF ? 3 : 0], q1 = (float)f[NF ? 4 : 1], q2 = (float)f[NF ? 5 : 2];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const float d0 = q0 - v4[u].x, d1 = q1 - v4[u].y, d2 = q2 - v4[u].z;
                h4[u] = d0 * d0 + d1 * d1 + d2 * d2;
              }
#pragma unroll
              for (int u = 0; u < 4; ++u)
                if (i + (uint32_t)u < e && !(h4[u] > thr32)) s_keep[bcnt++][threadIdx.x] = i + u;
              if (__ballot(bcnt >= (uint32_t)(kKeep - 4))) drain();
            }
            continue;
          }
          for (uint32_t i = b; i < e; i += 4) {
            double h4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const double *p = sfeat + (size_t)min(i + (uint32_t)u, e - 1u) * 6 + (NF ? 3 : 0);
              const double d0 = f[NF ? 3 : 0] - p[0], d1 = f[NF ? 4 : 1] - p[1], d2 = f[NF ? 5 : 2] - p[2];
              h4[u] = (0.0 + d0 * d0 + d1 * d1) + d2 * d2;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (i + (uint32_t)u < e && !(h4[u] > top.worst())) s_keep[bcnt++][threadIdx.x] = i + u;
            if (__ballot(bcnt >= (uint32_t)(kKeep - 4))) drain();
          }
        }
      }
    drain();
    if (deferred) break;
    // everything not yet scanned is at least `bound` away from the query (in 3-D, hence in 6-D)
    const double bound = r * M.h + in_cell;
    if (top.cnt == K && top.worst() < bound * bound) break;
    if (defer && scanned > budget && r < rmax) {
      deferred = true;
      break;
    }
  }
  {  // the deferred queries of this wavefront -> the list (one atomic per wavefront)
    const unsigned long long dm = __ballot(deferred);
    if (dm) {
      const int lane = threadIdx.x & 63;
      uint32_t base = 0;
      if (lane == __ffsll((long long)dm) - 1) base = atomicAdd(&defer[0], (uint32_t)__popcll(dm));
      base = (uint32_t)__shfl((int)base, __ffsll((long long)dm) - 1);
      if (deferred) defer[1 + base + (uint32_t)__popcll(dm & ((1ull << lane) - 1ull))] = qi;
    }
  }
#ifdef WC_PROF_KNN
  atomicAdd(&g_knn_prof[0], pc_), atomicAdd(&g_knn_prof[1], pr_), atomicAdd(&g_knn_prof[2], ps_), atomicAdd(&g_knn_prof[3], 1ull);
#endif
  if (kth_stat && (blockIdx.x & 15u) == 0u && q_begin + (blockIdx.x + 1u) * blockDim.x <= q_end) {  // (full workgroups only)  // a sample of the k-th distances (in cells^2): the next call's order of the halves
    const bool have = top.cnt == K && !deferred;
    double v = have ? fmin(top.worst() / (M.h * M.h), 1e6) : 0.0, c1 = have ? 1.0 : 0.0;
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m), c1 += __shfl_xor(c1, m);
    if ((threadIdx.x & 63) == 0) {
      double *slot = kth_stat + ((blockIdx.x >> 4) & 15u) * 16u;  // 16 slots, 128 bytes apart
      atomicAdd(slot, v);
      atomicAdd(slot + 1, c1);
    }
  }
  if (deferred) return;  // (k_knn_wave writes this query's lists)
  // Q10: FLANN leaves the tail of the result untouched (zero-initialised) when fewer than k targets exist
  uint32_t out = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const uint32_t c = (j < top.cnt) ? top.id[j] : 0u;
    if (knn_idx) {
      knn_idx[(size_t)q * K + j] = c;
      knn_d2[(size_t)q * K + j] = (j < top.cnt) ? top.d[j] : 0.0;
    }
    const double *w = tworld + (size_t)c * 7;
    if (fabs(w[6] - tq) < M.time_min) continue;                                      // cc:26
    const V3 nc = mk3(w[3], w[4], w[5]);
    if (acos(dot(nq_w, nc)) > M.ang_max) continue;                                    // cc:29, surfel.h:105-107
    if (fabs(dot(nq_w, cq - mk3(w[0], w[1], w[2]))) > M.dist_max) continue;           // cc:32
    if (gated_shard)
      gated_shard[(size_t)(qi - q_begin) * K + (out++)] = c;
    else
      gated[(size_t)(out++) * nq + q] = c;
  }
  for (; out < (uint32_t)K; ++out) {
    if (gated_shard)
      gated_shard[(size_t)(qi - q_begin) * K + out] = kNone;
    else
      gated[(size_t)out * nq + q] = kNone;
  }
}

// The queries k_knn_gate gave up on (more than `budget` candidates looked at and still not done): ONE WAVEFRONT per query.  The
// rows of a shell are looked up 64 at a time (lane = row: range + exact pruning bound), their candidates streamed 64 per load (lane =
// candidate: 48 contiguous bytes each, the distance summed in flann::L2_Simple's order as everywhere), the k best kept in lanes
// 0 .. K-1 in (distance, index) order - a strict total order, so the list that comes out is the one the per-lane search would
// have found.  Persistent wavefronts over the list (its length is only known on the device).
template <int K>
__global__ void __launch_bounds__(256) k_knn_wave(const wc_surfel *q_surf, const wc_pose *q_pose, uint32_t nq, const double *sfeat, const uint32_t *skeys,
                                                 const uint32_t *sorig, const double *tworld, uint32_t nt, MatchParams M, uint32_t *gated,
                                                 uint32_t *knn_idx, double *knn_d2, const uint32_t *__restrict__ qorder, uint32_t q_begin,
                                                 uint32_t *gated_shard, const uint32_t *defer) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = (gridDim.x * blockDim.x) >> 6;
  const uint32_t nd = defer[0];
  for (uint32_t w = wave; w < nd; w += nwave) {
    const uint32_t qi = defer[1 + w];
    const uint32_t q = qorder ? qorder[qi] : qi;
    double f[6];
    V3 cq, nq_w;
    feature6(q_surf[q], q_pose[q], M.cs, M.as, f, cq, nq_w);
    const double tq = q_surf[q].t;
    const double gx = (f[0] - M.org[0]) / M.h, gy = (f[1] - M.org[1]) / M.h, gz = (f[2] - M.org[2]) / M.h;
    const int cx = (int)floor(gx), cy = (int)floor(gy), cz = (int)floor(gz);
    const double in_cell = fmin(fmin(fmin(gx - cx, cx + 1 - gx), fmin(gy - cy, cy + 1 - gy)), fmin(gz - cz, cz + 1 - gz)) * M.h;
    const int rmax = max(max(max(cx, M.dim[0] - 1 - cx), max(cy, M.dim[1] - 1 - cy)), max(cz, M.dim[2] - 1 - cz));
    // the list: lane j < K holds the j-th best (ld, li); empty = (1e300, ~0)
    double ld = 1e300;
    uint32_t li = 0xFFFFFFFFu;
    int cnt = 0;  // (uniform)
    auto worst = [&]() -> double {
      const double dk = __shfl(ld, K - 1);
      return cnt < K ? 1e300 : dk;
    };
    for (int r = 0; r <= rmax; ++r) {
      const int side = 2 * r + 1, nrows = side * side;
      for (int r0 = 0; r0 < nrows; r0 += 64) {
        // lane = row (dy, dz) of the shell: its candidate ranges (two parts off the faces), pruned against the k-th distance
        const double wq = worst() / (M.h * M.h);
        uint32_t b0 = 0, e0 = 0, b1 = 0, e1 = 0;
        const int ri = r0 + lane;
        if (ri < nrows) {
          const int dz = ri / side - r, dy = ri % side - r;
          const int y = cy + dy, z = cz + dz;
          if (y >= 0 && y < M.dim[1] && z >= 0 && z < M.dim[2]) {
            const bool face = max(abs(dy), abs(dz)) == r;
            const int nparts = (face || r == 0) ? 1 : 2;
            const double dyd = dy > 0 ? (double)(cy + dy) - gy : (dy < 0 ? gy - (double)(cy + dy + 1) : 0.0);
            const double dzd = dz > 0 ? (double)(cz + dz) - gz : (dz < 0 ? gz - (double)(cz + dz + 1) : 0.0);
            const double rem = wq - (dyd * dyd + dzd * dzd);
            const int xs = rem < 1e12 ? (int)sqrt(fmax(rem, 0.0)) + 2 : (1 << 20);
            if (!(rem < -1e-9 * wq) && (face || r <= xs)) {
              const size_t row = (size_t)M.dim[0] * ((size_t)y + (size_t)M.dim[1] * (size_t)z);
              for (int part = 0; part < nparts; ++part) {
                int x0 = face ? max(cx - r, cx - xs) : (part == 0 ? cx - r : cx + r);
                int x1 = face ? min(cx + r, cx + xs) : x0;
                x0 = max(x0, 0), x1 = min(x1, M.dim[0] - 1);
                if (x0 > x1) continue;
                uint32_t b, e;
                if (M.cell_start) {
                  b = M.cell_start[row + x0], e = M.cell_start[row + x1 + 1];
                } else {
                  const uint32_t base = ((uint32_t)y << 10) | ((uint32_t)z << 20);
                  b = lower_bound_u32(skeys, nt, base | (uint32_t)x0), e = lower_bound_u32(skeys, nt, (base | (uint32_t)x1) + 1u);
                }
                if (part == 0)
                  b0 = b, e0 = e;
                else
                  b1 = b, e1 = e;
              }
            }
          }
        }
        // the rows with candidates, one after the other; 64 candidates per trip
        unsigned long long live = __ballot(e0 > b0 || e1 > b1);
        while (live) {
          const int src = __ffsll((long long)live) - 1;
          live &= live - 1ull;
          for (int part = 0; part < 2; ++part) {
            const uint32_t b = (uint32_t)__shfl((int)(part ? b1 : b0), src), e = (uint32_t)__shfl((int)(part ? e1 : e0), src);
            for (uint32_t i0 = b; i0 < e; i0 += 64u) {
              const uint32_t i = i0 + (uint32_t)lane;
              double sd = 1e300;
              uint32_t oid = 0xFFFFFFFFu;
              const double wk0 = worst();  // (read with all lanes active: a shuffle from an inactive lane returns 0)
              if (i < e) {
                const double *p = sfeat + (size_t)i * 6;
                double sacc = 0.0;
#pragma unroll
                for (int d = 0; d < 6; ++d) {
                  const double df = f[d] - p[d];
                  sacc += df * df;
                }
                sd = sacc;
                if (!(sd > wk0)) oid = sorig[i];
              }
              // insert the lanes' candidates that can still enter, one at a time (the k-th distance shrinks as they go in)
              unsigned long long pend = __ballot(oid != 0xFFFFFFFFu);
              while (pend) {
                const int c = __ffsll((long long)pend) - 1;
                pend &= pend - 1ull;
                const double cd = __shfl(sd, c);
                const uint32_t ci = (uint32_t)__shfl((int)oid, c);
                const double wk = __shfl(ld, K - 1);
                const uint32_t wi = (uint32_t)__shfl((int)li, K - 1);
                if (cnt == K && !(cd < wk || (cd == wk && ci < wi))) continue;
                // position = number of list entries before (cd, ci); entries from there on move up one lane
                const bool before = lane < K && (ld < cd || (ld == cd && li < ci));
                const int pos = __popcll(__ballot(before));
                const double upd = __shfl_up(ld, 1);
                const uint32_t upi = (uint32_t)__shfl_up((int)li, 1);
                if (lane < K && lane > pos) ld = upd, li = upi;
                if (lane == pos) ld = cd, li = ci;
                if (cnt < K) ++cnt;
              }
            }
          }
        }
      }
      const double bound = r * M.h + in_cell;
      if (cnt == K && worst() < bound * bound) break;
    }
    // outputs: lane j < K = the j-th neighbour (Q10: index 0 beyond the number of targets), gates as in k_knn_gate
    const uint32_t c = (lane < K && lane < cnt) ? li : 0u;
    bool pass = false;
    if (lane < K) {
      if (knn_idx) {
        knn_idx[(size_t)q * K + lane] = c;
        knn_d2[(size_t)q * K + lane] = lane < cnt ? ld : 0.0;
      }
      const double *wv = tworld + (size_t)c * 7;
      pass = !(fabs(wv[6] - tq) < M.time_min);
      const V3 nc = mk3(wv[3], wv[4], wv[5]);
      pass = pass && !(acos(dot(nq_w, nc)) > M.ang_max);
      pass = pass && !(fabs(dot(nq_w, cq - mk3(wv[0], wv[1], wv[2]))) > M.dist_max);
    }
    const unsigned long long pm = __ballot(pass);
    const int npass = __popcll(pm);
    if (lane < K) {
      const int o = __popcll(pm & ((1ull << lane) - 1ull));
      if (pass) {
        if (gated_shard)
          gated_shard[(size_t)(qi - q_begin) * K + o] = c;
        else
          gated[(size_t)o * nq + q] = c;
      }
      if (lane >= npass) {
        if (gated_shard)
          gated_shard[(size_t)(qi - q_begin) * K + lane] = kNone;
        else
          gated[(size_t)lane * nq + q] = kNone;
      }
    }
  }
}

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
__global__ void __launch_bounds__(256) k_resolve(const uint32_t *gated, uint32_t nq, int k, int same_set, const uint32_t *choice_in,
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
      if (same_set && c < q && choice_in[c] == q) continue;  // {c, q} is already in surfel_pairs (cc:35-38)
      pick = c;
      break;
    }
    choice_out[q] = pick;
    if (pick != choice_in[q]) s_changed = 1u;
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_changed) *changed = 1u;
}

__global__ void __launch_bounds__(256) k_flags(const uint32_t *choice, uint32_t nq, uint32_t *flags) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nq) flags[q] = choice[q] != kNone ? 1u : 0u;
}

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

// scratch lives in ctx->b_misc[1..7] slots to avoid another state struct.  want_shard: the call is a collective of the ctx's
// communicator (wc_match_sharded) - every rank makes it with the same replicated arguments
static int match_impl(wc_ctx *ctx, const wc_surfel *d_q_surf, const wc_pose *d_q_pose, uint64_t nq_, const wc_surfel *d_t_surf,
                      const wc_pose *d_t_pose, uint64_t nt_, int same_set, wc_pair *d_pairs, uint64_t cap, uint64_t *h_n_pairs,
                      uint32_t *d_knn_idx, double *d_knn_d2, bool want_shard) {
  wc_dev_guard dg_(ctx);
  const auto t_entry = std::chrono::steady_clock::now();
  if (!ctx || !h_n_pairs) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  *h_n_pairs = 0;
  if (nt_ == 0 || nq_ == 0) return WC_OK;  // knn_surfel_matcher.cc:18-20
  if (nq_ >= (1ull << 31) || nt_ >= (1ull << 31)) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  const uint32_t nq = (uint32_t)nq_, nt = (uint32_t)nt_;
  const wc_params &P = ctx->P;
  hipStream_t st = ctx->stream;
  wc_buf &b_feat = ctx->b_misc[1], &b_world = ctx->b_misc[2], &b_sfeat = ctx->b_misc[3], &b_gated = ctx->b_misc[4],
         &b_choice = ctx->b_misc[5], &b_aux = ctx->b_misc[6], &b_scan = ctx->b_misc[7];
  WC_TRY(wc_ensure(ctx, b_feat, (size_t)nt * 6 * 8));
  WC_TRY(wc_ensure(ctx, b_world, (size_t)nt * 7 * 8));
  WC_TRY(wc_ensure(ctx, b_sfeat, (size_t)nt * 6 * 8));
  WC_TRY(wc_ensure(ctx, ctx->b_match_half, (size_t)nt * 2 * 16 + 16));
  WC_TRY(wc_ensure(ctx, b_gated, (size_t)nq * P.knn_k * 4));
  WC_TRY(wc_ensure(ctx, b_choice, (size_t)nq * 4 * 4));  // choice[2], flags, offsets
  WC_TRY(wc_ensure(ctx, ctx->b_keys[0], (size_t)std::max(nt, nq) * 4));
  WC_TRY(wc_ensure(ctx, ctx->b_keys[1], (size_t)nt * 4));
  WC_TRY(wc_ensure(ctx, ctx->b_vals[0], (size_t)std::max(nt, nq) * 4));
  WC_TRY(wc_ensure(ctx, ctx->b_vals[1], (size_t)nt * 4));
  WC_TRY(wc_ensure(ctx, b_aux, 256));
  WC_TRY(wc_ensure(ctx, ctx->b_status, 64 * 4));
  uint32_t *status = (uint32_t *)ctx->b_status.p;
  unsigned long long *bbox = (unsigned long long *)b_aux.p;
  uint32_t *changed = (uint32_t *)((char *)b_aux.p + 64);

  // 1. features + bounding box of the scaled centres
  unsigned long long init[6] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull};
  WC_HIP(ctx, hipMemcpyAsync(bbox, init, sizeof(init), hipMemcpyHostToDevice, st));
  WC_HIP(ctx, hipMemsetAsync(status, 0, 64 * 4, st));
  k_features<<<(nt + 255) / 256, 256, 0, st>>>(d_t_surf, d_t_pose, nt, P.center_scale, P.angular_scale, (double *)b_feat.p,
                                              (double *)b_world.p, bbox);
  // optional (WC_KNN_CELL=<factor> or WC_MATCH_DEBUG): the k-th 6-D distances of a sample of this call's queries (k_kth_sample),
  // read back with the bounding box
  static const char *cell_env = getenv("WC_KNN_CELL");
  static const bool match_dbg = getenv("WC_MATCH_DEBUG") != nullptr;
  const bool sample = (cell_env && cell_env[0] != 'v') || match_dbg;
  const uint32_t samp_stride = (uint32_t)((nt + 262143u) / 262144u);  // at most 256 k targets are looked at
  unsigned long long hb[6];
  uint32_t h_cnt[kSampleQ * kSampleR];
  if (sample) {
    WC_TRY(wc_ensure(ctx, ctx->b_match_samp, (size_t)kSampleQ * 6 * 8 + (size_t)kSampleQ * kSampleR * 4));
    double *samp_f = (double *)ctx->b_match_samp.p;
    uint32_t *samp_c = (uint32_t *)(samp_f + kSampleQ * 6);
    k_sample_feats<<<1, kSampleQ, 0, st>>>(d_q_surf, d_q_pose, nq, P.center_scale, P.angular_scale, samp_f, samp_c);
    k_kth_sample<<<(unsigned)(((nt + samp_stride - 1) / samp_stride + 255) / 256), 256, 0, st>>>((const double *)b_feat.p, nt, samp_stride, samp_f, samp_c);
    WC_HIP(ctx, hipMemcpyAsync(h_cnt, samp_c, sizeof(h_cnt), hipMemcpyDeviceToHost, st));
  }
  WC_HIP(ctx, hipMemcpyAsync(hb, bbox, sizeof(hb), hipMemcpyDeviceToHost, st));
  WC_HIP(ctx, hipStreamSynchronize(st));
  MatchParams M;
  M.cs = P.center_scale, M.as = P.angular_scale;
  M.time_min = P.time_diff_min, M.ang_max = P.angular_scale, M.dist_max = P.surfel_dist_max;
  M.k = P.knn_k;
  double lo[3], hi[3], ext = 0;
  for (int d = 0; d < 3; ++d) {
    lo[d] = dec_host(hb[d]);
    hi[d] = dec_host(hb[3 + d]);
    if (!std::isfinite(lo[d]) || !std::isfinite(hi[d])) return wc_fail(ctx, WC_ERR_ARG, "non-finite surfel centre");
    ext = std::max(ext, hi[d] - lo[d]);
  }
  // cell size: about 4 targets per occupied-volume cell, at least extent / 1000, at most one scaled unit (= 1 m) when a set is
  // matched against itself and two when it is matched against another set.  (The fixed window holds a fifth of the sliding window's
  // surfels - one or two re-observations per query -, so the k-th neighbour of a query is 5 - 8 units away and the density rule asks
  // for cells of 1.3 - 1.9 units; capped at one, a query walked 178 rows of ~5 candidates.  Odometry step, both searches, 20
  // repetitions each on one box: cap 1 / 1 -> 2.39 ms, other set 1.5 / 2 / 3 -> 2.08 / 2.06 / 2.07 ms; C4 window 543 -> 587 M
  // surfels/s.  For the same-set search the cap stays: 0.8 / 1.0 / 1.2 -> 2.01 / 2.07 / 2.05, but 1.202 - the density rule's own
  // figure there, which cuts the 0.4 m patch lattice of the synthetic windows unevenly - 2.51.)  WC_KNN_HCAP_SAME / _OTHER: A/B.
  double vol = 1.0;
  for (int d = 0; d < 3; ++d) vol *= std::max(hi[d] - lo[d], 0.05);
  const char *hcap_env = getenv(same_set ? "WC_KNN_HCAP_SAME" : "WC_KNN_HCAP_OTHER");
  const double h_cap = hcap_env ? atof(hcap_env) : (same_set ? 1.0 : 2.0), h_raw = std::cbrt(4.0 * vol / (double)nt);
  const double h_vol = std::min(h_cap, h_raw);
  M.h = h_vol;
  if (getenv("WC_MATCH_DEBUG")) fprintf(stderr, "[match] nq %u nt %u same %d: density rule %.3f, capped %.3f\n", nq, nt, same_set, h_raw, h_vol);
  // Round 3 tried to take the cell size from THIS call's data: kSampleQ queries measure their k-th 6-D distance
  // (k_kth_sample) and the grid gets cells of a multiple of the median.  Measured (profiles/dev/time_match.py, time_facade.py;
  // DESIGN 3.3): the bench windows (random normals, k-th distance 2 - 5.7 units) are fastest with the density rule's cells of a
  // quarter of that distance; on the facade's room stream the MEDIAN is 0.18 - 0.25 units but the tail reaches 2.8 - 5.7 (surfels
  // whose normal has no like within metres), and a query walks (k-th distance / h)^3 cells: cells of 2 x the median made the
  // search 5 x slower (25 - 40 ms), of 0.5 x the median 2000 x.  The tail, not the median, sets the cost of a search, so the
  // density rule stays the default and the sampled rule an experiment behind WC_KNN_CELL=<factor>.  (Measured again with the heavy
  // queries handed to k_knn_wave, and with table look-ups counted against the budget: finer cells still lose on the room stream -
  // 2 - 12 ms against 1.7 - 2.3 - and counting look-ups sends half of a sparse fixed window's queries to the wavefront kernel.)
  if (sample) {
    // median over the samples of the smallest radius that holds k targets (strided counts scaled up); samples that never
    // reach k inside 22 units (fewer than k targets, or a target set far from the queries) vote for the largest radius
    std::vector<double> rk;
    for (int s = 0; s < kSampleQ; ++s) {
      uint64_t c = 0;
      int b = kSampleR;
      for (int r = 0; r < kSampleR; ++r) {
        c += (uint64_t)h_cnt[s * kSampleR + r] * samp_stride;
        if (c >= (uint64_t)std::min<uint32_t>((uint32_t)P.knn_k, nt)) {
          b = r;
          break;
        }
      }
      rk.push_back(std::ldexp(1.0, b - 6) > 0 ? std::sqrt(std::ldexp(1.0, b - 6)) : 0.125);
    }
    std::nth_element(rk.begin(), rk.begin() + rk.size() / 2, rk.end());
    const double r_med = rk[rk.size() / 2];
    const double factor = (cell_env && cell_env[0] != 'v') ? atof(cell_env) : 0.0;
    if (factor > 0.0) M.h = std::min(h_vol, factor * r_med);
    ctx->match_last_rk = r_med;
    if (match_dbg) {
      std::sort(rk.begin(), rk.end());
      fprintf(stderr, "[match] nq %u nt %u same %d: sampled k-th distance p10 %.3f p50 %.3f p90 %.3f max %.3f; h_vol %.3f -> h %.3f\n", nq, nt, same_set,
              rk[rk.size() / 10], rk[rk.size() / 2], rk[rk.size() * 9 / 10], rk.back(), h_vol, M.h);
    }
  }
  M.h = std::max(M.h, std::max(ext / 1000.0, 1e-3));
  for (int d = 0; d < 3; ++d) {
    M.org[d] = lo[d];
    M.dim[d] = std::min(1024, (int)std::floor((hi[d] - lo[d]) / M.h) + 1);
  }
  // 2. sort targets by cell (x fastest).  (rocPRIM's default - a merge sort below 2^20 items; the Onesweep radix path that
  // window.hip / extract.hip force is SLOWER here: 2.54 against 2.41 ms for the odometry step's two searches, three alternations of
  // 20 repetitions on one box - its look-back passes run beside the other search's k_knn_gate)
  // (Also tried: a counting sort through the dense cell table - atomic counts, a scan over the cells, an atomic scatter: three launches.
  // 2.00 against 2.03 ms, and the order inside a cell is whatever the atomics make it: the ranks of a query-sharded search, which
  // shard the POSITIONS of the cell order, then disagree about who searches what.  The stable key sort stays.)
  uint32_t *k0 = (uint32_t *)ctx->b_keys[0].p, *k1 = (uint32_t *)ctx->b_keys[1].p;
  uint32_t *v0 = (uint32_t *)ctx->b_vals[0].p, *v1 = (uint32_t *)ctx->b_vals[1].p;
  k_cell_keys<<<(nt + 255) / 256, 256, 0, st>>>((const double *)b_feat.p, nt, M, k0, v0);
  {
    size_t tmp = 0;
    WC_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp, k0, k1, v0, v1, (size_t)nt, 0u, 30u, st));
    WC_TRY(wc_ensure(ctx, ctx->b_sorttmp, tmp));
    tmp = ctx->b_sorttmp.cap;
    WC_HIP(ctx, rocprim::radix_sort_pairs(ctx->b_sorttmp.p, tmp, k0, k1, v0, v1, (size_t)nt, 0u, 30u, st));
  }
  k_sorted_feat<<<(nt + 255) / 256, 256, 0, st>>>((const double *)b_feat.p, v1, nt, (double *)b_sfeat.p, (float4 *)ctx->b_match_half.p);
  const size_t ncell = (size_t)M.dim[0] * M.dim[1] * M.dim[2];
  M.cell_start = nullptr;
  if (ncell <= (1u << 24)) {  // dense [start, end) table (<= 128 MB); larger grids fall back to binary searches
    WC_TRY(wc_ensure(ctx, ctx->b_misc[0], (ncell + 1) * 4));
    WC_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->b_misc[0].p, (int)nt, ncell + 1, st));
    M.cell_start = (const uint32_t *)ctx->b_misc[0].p;
    k_cell_table<<<(nt + 255) / 256, 256, 0, st>>>(k1, nt, M, (uint32_t *)ctx->b_misc[0].p);
  }
  // 3. exact k-NN + gates, queries in the order of their grid cell
  const uint32_t *qorder = v1;  // same set: the sorted target permutation
  if (!same_set) {
    uint32_t *qk = (uint32_t *)b_choice.p + 2 * (size_t)nq, *qo = (uint32_t *)b_choice.p + 3 * (size_t)nq;  // (flags / offsets: free until step 5)
    k_query_keys<<<(nq + 255) / 256, 256, 0, st>>>(d_q_surf, d_q_pose, nq, M, k0, v0);
    size_t tmp = 0;
    WC_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp, k0, qk, v0, qo, (size_t)nq, 0u, 30u, st));
    WC_TRY(wc_ensure(ctx, ctx->b_sorttmp, tmp));
    tmp = ctx->b_sorttmp.cap;
    WC_HIP(ctx, rocprim::radix_sort_pairs(ctx->b_sorttmp.p, tmp, k0, qk, v0, qo, (size_t)nq, 0u, 30u, st));
    qorder = qo;
  }
  // several GPUs: the queries are independent (knn_surfel_matcher.cc:22-48), the targets are replicated;
  // every rank searches a contiguous share of the queries (in cell order) and ONE all-gather of the gated lists (4 k bytes per
  // query) gives every rank the whole table; the order-dependent de-duplication below then runs replicated
  // (the predicate only depends on arguments every rank shares: a rank-dependent one would leave the others in the all-gather)
  const bool sharded = want_shard && ctx->have_comm && ctx->comm.world > 1 && ctx->comm.allgatherv && nq >= 4096;
  uint32_t q_begin = 0, q_end = nq;
  if (sharded) {
    const uint32_t w = (uint32_t)ctx->comm.world, r = (uint32_t)ctx->comm.rank;
    q_begin = (uint32_t)(((uint64_t)nq * r) / w), q_end = (uint32_t)(((uint64_t)nq * (r + 1)) / w);
  }
  // the searches' output: position-major lists (this rank's share [q_begin, q_end) of the positions when sharded); all ranks'
  // lists - or simply this call's - then sit in b_route[2] and are transposed into the planes of b_gated
  WC_TRY(wc_ensure(ctx, ctx->b_route[2], (size_t)nq * P.knn_k * 4));
  WC_TRY(wc_ensure(ctx, ctx->b_route[3], (size_t)(q_end - q_begin + 1) * P.knn_k * 4));
  uint32_t *gated_shard = sharded ? (uint32_t *)ctx->b_route[3].p : (uint32_t *)ctx->b_route[2].p;
  const uint32_t nq_mine = q_end - q_begin;
  // order of the two halves of a candidate (see k_knn_gate).  First guess: from the k-th distances of the previous call of this
  // kind on this context.  That rule is wrong for windows whose 6-D distances are dominated by the normals' noise (the facade's
  // room stream: the k-th neighbour lies within 1.5 cells, yet the normal half first is 1.6 x faster), so the device time of
  // the search is measured and, once both orders have been tried, the faster one is used; the other is tried again every 64th
  // call.  The lists do not depend on the order.
  const int kind = same_set ? 1 : 0;
  {  // a different workload (twice / half the queries or targets of the previous call of this kind) starts without history: the
     // per-query times of a 1 M-surfel window say nothing about a 250 k one, and with stale figures for the order NOT in use the
     // choice took half a dozen calls to turn - bench.py's odometry step, measured behind its window section, 3.3 ms instead of 2.5
    uint32_t *pn = ctx->match_prev_n[kind];
    const auto far = [](uint32_t a, uint32_t b) { return a > 2u * b || 2u * a < b; };
    if (pn[0] && (far(nq, pn[0]) || far(nt, pn[1]))) {
      ctx->match_ns_per_q[kind][0] = ctx->match_ns_per_q[kind][1] = 0.0;
      ctx->match_calls[kind] = 0;
    }
    pn[0] = nq, pn[1] = nt;
  }
  bool nf = ctx->match_nf[kind];
  {
    // calls 0 / 2 of a workload: the rule's order, calls 1 / 3: the other one; then the order whose BEST time is lower (the times are
    // taken beside the other search of wc_match_pair: one sample per order, averaged, once left the slower order in use for good -
    // its only sample of the faster one had waited for wavefront slots), the other one again every 64th call
    const double t0 = ctx->match_ns_per_q[kind][0], t1 = ctx->match_ns_per_q[kind][1];
    const uint32_t call = ctx->match_calls[kind]++;
    if (call < 4u)
      nf = (call & 1u) ? !nf : nf;
    else if (t0 > 0.0 && t1 > 0.0) {
      nf = t1 < t0;
      if ((call & 63u) == 63u) nf = !nf;
    }
  }
  if (const char *o = getenv("WC_KNN_ORDER")) nf = o[0] == 'n';  // "normal" / "centre": tests pin each instantiation
  // the single-precision first look (k_knn_gate<.., F32>): for windows whose k-th neighbour was cells away in the previous call
  bool f32 = ctx->match_sparse[kind];
  if (const char *o = getenv("WC_KNN_F32")) f32 = o[0] == '1';  // (tests pin each instantiation)
  for (hipEvent_t &e : ctx->ev_knn)
    if (!e) WC_HIP(ctx, hipEventCreate(&e));
  static const bool tdbg = getenv("WC_MATCH_TIMING") != nullptr;
  const auto t_prep = std::chrono::steady_clock::now();
  WC_HIP(ctx, hipEventRecord(ctx->ev_knn[0], st));
  WC_TRY(wc_ensure(ctx, ctx->b_match_stat, 16 * 16 * 8));
  double *kth_stat = (double *)ctx->b_match_stat.p;
  WC_HIP(ctx, hipMemsetAsync(kth_stat, 0, 16 * 16 * 8, st));
  // queries that look at more than `budget` candidates without finishing are handed to k_knn_wave (WC_KNN_BUDGET: 0 = never)
  static const char *budget_env = getenv("WC_KNN_BUDGET");
  const uint32_t budget = budget_env ? (uint32_t)atoi(budget_env) : 4096u;
  uint32_t *defer = nullptr;
  if (budget && nq_mine) {
    WC_TRY(wc_ensure(ctx, ctx->b_match_defer, ((size_t)nq_mine + 2) * 4));
    defer = (uint32_t *)ctx->b_match_defer.p;
    WC_HIP(ctx, hipMemsetAsync(defer, 0, 4, st));
  }
#define WC_KNN_ARGS                                                                                                              \
  d_q_surf, d_q_pose, nq, (const double *)b_sfeat.p, k1, v1, (const double *)b_world.p, nt, M, (uint32_t *)b_gated.p, d_knn_idx, d_knn_d2, qorder, \
      q_begin, q_end, gated_shard, kth_stat, budget, defer, (const float4 *)ctx->b_match_half.p + (nf ? (size_t)nt : 0)
#define WC_KNN_LAUNCH(KK)                                                                                                        \
  if (nq_mine) {                                                                                                                 \
    if (KK == 10 && f32 && nf)                                                                                                   \
      k_knn_gate<KK, true, (KK == 10)><<<(nq_mine + kKnnThreads - 1) / kKnnThreads, kKnnThreads, 0, st>>>(WC_KNN_ARGS);                                       \
    else if (KK == 10 && f32)                                                                                                    \
      k_knn_gate<KK, false, (KK == 10)><<<(nq_mine + kKnnThreads - 1) / kKnnThreads, kKnnThreads, 0, st>>>(WC_KNN_ARGS);                                      \
    else if (nf)                                                                                                                 \
      k_knn_gate<KK, true><<<(nq_mine + kKnnThreads - 1) / kKnnThreads, kKnnThreads, 0, st>>>(WC_KNN_ARGS);                                                   \
    else                                                                                                                         \
      k_knn_gate<KK, false><<<(nq_mine + kKnnThreads - 1) / kKnnThreads, kKnnThreads, 0, st>>>(WC_KNN_ARGS);                                                  \
    if (defer)                                                                                                                   \
      k_knn_wave<KK><<<1024, 256, 0, st>>>(d_q_surf, d_q_pose, nq, (const double *)b_sfeat.p, k1, v1, (const double *)b_world.p, nt, M,       \
                                          (uint32_t *)b_gated.p, d_knn_idx, d_knn_d2, qorder, q_begin, gated_shard, defer);                 \
  }
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
#undef WC_KNN_ARGS
  const auto t_launched = std::chrono::steady_clock::now();
  WC_HIP(ctx, hipGetLastError());
  WC_HIP(ctx, hipEventRecord(ctx->ev_knn[1], st));
  if (sharded) {
    const int w = ctx->comm.world;
    std::vector<uint64_t> bytes((size_t)w);
    for (int r = 0; r < w; ++r)
      bytes[r] = (uint64_t)(((uint64_t)nq * (r + 1)) / w - ((uint64_t)nq * r) / w) * P.knn_k * 4;
    if (!ctx->comm.stream_ordered) WC_HIP(ctx, hipStreamSynchronize(st));
    if (ctx->comm.allgatherv(ctx->comm.user, gated_shard, (uint64_t)nq_mine * P.knn_k * 4, ctx->b_route[2].p, bytes.data()) != 0)
      return wc_fail(ctx, WC_ERR_HIP, "wc_match: all-gather of the gated neighbour lists failed");
  }
  {
    uint32_t *qpos = (uint32_t *)b_choice.p + nq;  // (choice[1]: free until the resolve rounds)
    k_inv_perm<<<(nq + 255) / 256, 256, 0, st>>>(qorder, nq, qpos);
    k_gated_planes<<<(nq + 255) / 256, 256, 0, st>>>((const uint32_t *)ctx->b_route[2].p, qpos, nq, P.knn_k, (uint32_t *)b_gated.p);
    WC_HIP(ctx, hipGetLastError());
  }
  // 4. resolve the order-dependent "pair already seen" rule by fixed-point iteration
  uint32_t *choice[2] = {(uint32_t *)b_choice.p, (uint32_t *)b_choice.p + nq};
  uint32_t *flags = (uint32_t *)b_choice.p + 2 * (size_t)nq, *offsets = (uint32_t *)b_choice.p + 3 * (size_t)nq;
  WC_HIP(ctx, hipMemsetAsync(choice[0], 0xFF, (size_t)nq * 4, st));
  int cur = 0;
  // rounds are issued eight at a time between host checks (a round past the fixed point changes nothing, so the extra ones
  // are harmless); round r of a batch reports into changed[r] and only the last word is read back
  // The compaction (step 5) is enqueued right behind every batch, before the host knows whether the batch reached the fixed point: the
  // rule - it did - then costs ONE host round trip for rounds + compaction instead of two (a batch that did not is followed by another
  // one, and the compaction is redone on its result).
  bool converged = false;
  double h_stat[16 * 16];
  {
    size_t tmp = 0;
    WC_HIP(ctx, rocprim::exclusive_scan(nullptr, tmp, flags, offsets, 0u, (size_t)nq, rocprim::plus<uint32_t>(), st));
    WC_TRY(wc_ensure(ctx, b_scan, tmp + 16));
  }
  for (int batch = 0; batch < 250000 && !converged; ++batch) {
    const int rounds = same_set ? 8 : 1;
    WC_HIP(ctx, hipMemsetAsync(changed, 0, 32, st));
    for (int r = 0; r < rounds; ++r) {
      k_resolve<<<(nq + 255) / 256, 256, 0, st>>>((const uint32_t *)b_gated.p, nq, P.knn_k, same_set, choice[cur], choice[cur ^ 1], changed + r);
      cur ^= 1;
    }
    uint32_t hc8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    WC_HIP(ctx, hipMemcpyAsync(hc8, changed, 32, hipMemcpyDeviceToHost, st));
    // 5. compact in query order
    if (batch > 0) WC_HIP(ctx, hipMemsetAsync(status, 0, 8, st));  // (count + order flag of the previous, unconverged, attempt)
    k_flags<<<(nq + 255) / 256, 256, 0, st>>>(choice[cur], nq, flags);
    {
      size_t tmp = b_scan.cap;
      WC_HIP(ctx, rocprim::exclusive_scan(b_scan.p, tmp, flags, offsets, 0u, (size_t)nq, rocprim::plus<uint32_t>(), st));
    }
    k_emit_pairs<<<(nq + 255) / 256, 256, 0, st>>>(choice[cur], offsets, nq, d_q_surf, (const double *)b_world.p, same_set, d_pairs, cap, status);
    WC_HIP(ctx, hipGetLastError());
    if (batch == 0) WC_HIP(ctx, hipMemcpyAsync(h_stat, kth_stat, sizeof(h_stat), hipMemcpyDeviceToHost, st));
    WC_HIP(ctx, hipMemcpyAsync(ctx->h_status, status, 8, hipMemcpyDeviceToHost, st));
    WC_HIP(ctx, hipStreamSynchronize(st));
    const uint32_t hc = hc8[rounds - 1];
    if (getenv("WC_MATCH_DEBUG"))
      fprintf(stderr, "[match] resolve batch %d: rounds that changed something %u%u%u%u%u%u%u%u\n", batch, hc8[0], hc8[1], hc8[2], hc8[3], hc8[4], hc8[5],
              hc8[6], hc8[7]);
    converged = !hc || !same_set;
  }
  if (!converged) return wc_fail(ctx, WC_ERR_NUMERIC, "wc_match: the pair de-duplication did not reach its fixed point");
  {
    double sum = 0.0, cnt = 0.0;
    for (int s = 0; s < 16; ++s) sum += h_stat[16 * s], cnt += h_stat[16 * s + 1];
    if (cnt > 0.0) ctx->match_nf[same_set ? 1 : 0] = sum / cnt > 2.25;  // mean k-th distance beyond 1.5 cells: the centre half prunes little
    if (cnt > 0.0) ctx->match_sparse[same_set ? 1 : 0] = sum / cnt > 2.25;
    float ms = 0.f;
    if (nq_mine >= 4096 && hipEventElapsedTime(&ms, ctx->ev_knn[0], ctx->ev_knn[1]) == hipSuccess && ms > 0.f) {
      double &t = ctx->match_ns_per_q[kind][nf ? 1 : 0];
      const double now = (double)ms * 1e6 / (double)nq_mine;
      t = t > 0.0 ? fmin(now, 1.05 * t) : now;  // the best of the recent samples (an old best fades by 5 % per call)
    }
  }
#ifdef WC_PROF_KNN
  {
    unsigned long long h[4], z[4] = {0, 0, 0, 0};
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_knn_prof), sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_knn_prof), z, sizeof(z));
    fprintf(stderr, "knn: %llu queries, per query %.1f candidates, %.1f rows, %.2f shells; cell h = %.4f, dims %d x %d x %d\n", h[3], (double)h[0] / h[3],
            (double)h[1] / h[3], (double)h[2] / h[3], M.h, M.dim[0], M.dim[1], M.dim[2]);
  }
#endif
  if (tdbg) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ctx->ev_knn[0], ctx->ev_knn[1]);
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    const auto t_end = std::chrono::steady_clock::now();
    fprintf(stderr, "[match] kind %d nq %u nt %u: host enqueue of the preparation %.0f us, rendezvous + launch %.0f us, launch -> done %.0f us (k_knn_gate + k_knn_wave by events %.0f us), nf %d f32 %d\n",
            kind, nq, nt, us(t_entry, t_prep), us(t_prep, t_launched), us(t_launched, t_end), ms * 1e3, (int)nf, (int)f32);
  }
  *h_n_pairs = ctx->h_status[0];
  if (ctx->h_status[1] & 2u) return wc_fail(ctx, WC_ERR_ORDER, "fixed-window surfel newer than its sliding-window match");
  if (ctx->h_status[0] > cap) return wc_fail(ctx, WC_ERR_CAPACITY, "pair capacity %llu < %u", (unsigned long long)cap, ctx->h_status[0]);
  return WC_OK;
}

extern "C" int wc_match(wc_ctx *ctx, const wc_surfel *d_q_surf, const wc_pose *d_q_pose, uint64_t nq, const wc_surfel *d_t_surf,
                        const wc_pose *d_t_pose, uint64_t nt, int same_set, wc_pair *d_pairs, uint64_t cap, uint64_t *h_n_pairs,
                        uint32_t *d_knn_idx, double *d_knn_d2) {
  return match_impl(ctx, d_q_surf, d_q_pose, nq, d_t_surf, d_t_pose, nt, same_set, d_pairs, cap, h_n_pairs, d_knn_idx, d_knn_d2, false);
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

// The two searches of an outer iteration side by side (see include/wildcat_hip.h).  wc_match is synchronous and talks to the
// host between its launches (the fixed-point rounds of the pair rule), so the second search gets its own context AND its own
// host thread; both only read the surfels.
extern "C" int wc_match_pair(wc_ctx *ctx, const wc_surfel *d_sld_surf, const wc_pose *d_sld_pose, uint64_t n_sld,
                             const wc_surfel *d_fix_surf, const wc_pose *d_fix_pose, uint64_t n_fix, wc_pair *d_pairs_sld,
                             uint64_t cap_sld, uint64_t *h_n_pairs_sld, wc_pair *d_pairs_fix, uint64_t cap_fix,
                             uint64_t *h_n_pairs_fix) {
  if (!ctx || !h_n_pairs_sld || !h_n_pairs_fix) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  static const bool serial = getenv("WC_MATCH_PAIR_SERIAL") != nullptr;
  if (n_sld == 0 || n_fix == 0 || serial) {
    WC_TRY(wc_match(ctx, d_sld_surf, d_sld_pose, n_sld, d_sld_surf, d_sld_pose, n_sld, 1, d_pairs_sld, cap_sld, h_n_pairs_sld, nullptr, nullptr));
    return wc_match(ctx, d_sld_surf, d_sld_pose, n_sld, d_fix_surf, d_fix_pose, n_fix, 0, d_pairs_fix, cap_fix, h_n_pairs_fix, nullptr, nullptr);
  }
  wc_dev_guard dg_(ctx);
  if (!ctx->aux) {
    const int rc = wc_ctx_create(&ctx->P, ctx->device, &ctx->aux);
    if (rc != WC_OK) return wc_fail(ctx, rc, "wc_match_pair: no helper context");
  }
  wc_ctx *aux = ctx->aux;
  WC_TRY(wc_ctx_set_params(aux, &ctx->P));
  // the helper's stream starts behind everything already enqueued on the ctx stream (the producers of the surfels and poses)
  if (!ctx->ev_aux) WC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_aux, hipEventDisableTiming));
  WC_HIP(ctx, hipEventRecord(ctx->ev_aux, ctx->stream));
  WC_HIP(ctx, hipStreamWaitEvent(aux->stream, ctx->ev_aux, 0));
  int rc_fix = WC_OK, rc_sld = WC_OK;
  // (Round 3, tried twice: a rendezvous of the two searches behind their preparations - both k_knn_gate ready together, the ctx
  // stream's enqueued first, with and without a lower priority for the helper's stream.  2.50 - 2.76 ms for the odometry step's
  // searches against 2.45 - 2.50 without, alternating on one box: the search that is ready first had better start.  What had made
  // single runs take 3.1 - 3.4 ms was the ORDER of the candidate halves chosen from one noisy timing sample - see match_impl.)
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
  static const bool swap = getenv("WC_MATCH_PAIR_SWAP") != nullptr;
  wc_ctx *c_fix = swap ? ctx : aux, *c_sld = swap ? aux : ctx;
  auto search_fix = [&] { return wc_match(c_fix, d_sld_surf, d_sld_pose, n_sld, d_fix_surf, d_fix_pose, n_fix, 0, d_pairs_fix, cap_fix, h_n_pairs_fix, nullptr, nullptr); };
  auto search_sld = [&] { return wc_match(c_sld, d_sld_surf, d_sld_pose, n_sld, d_sld_surf, d_sld_pose, n_sld, 1, d_pairs_sld, cap_sld, h_n_pairs_sld, nullptr, nullptr); };
  bool threaded = true;
  wc_pair_worker *worker = (wc_pair_worker *)ctx->pair_worker;
  if (!worker) {
    try {
      worker = new wc_pair_worker();
      ctx->pair_worker = worker;
      ctx->pair_worker_free = [](void *p) { delete (wc_pair_worker *)p; };
    } catch (...) {
      worker = nullptr;
      threaded = false;
    }
  }
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

