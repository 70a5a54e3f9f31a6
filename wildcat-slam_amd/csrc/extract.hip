// extract.hip — surfel extraction on gfx950: root-voxel binning, 3-level octree planarity tests, temporal
// clustering and per-cluster 3x3 PCA.  Replaces BuildSurfels (src/odometry/surfel_extraction.cc:316-337) and
// everything beneath it (BuildVoxelMap :186-220, InitOctoTree :128-140, CutOctoTree :142-184, InitPlane :82-126,
// ExtractSurfelInfo :304-314, ClusterSurfels :12-65).  Decision rules: SURVEY.md Appendix A.
//
// Pipeline of one sweep (all on one stream, no host synchronisation until wc_extract_surfels_finish):
//   0. k_init        every per-call fill in one launch (status words, bucket / bin counters, mailbox address); on the fast
//                    path it is issued by the PREVIOUS finish(), so that it runs while the host turns around
//   1. k_pt_runs     the only pass over the AoS input: root-voxel key relative to the voxel of point 0
//                    (floor(p / (double)0.8f), true fp64 division; 10 bits per axis, 21 in the wide fallback), RUNS of
//                    consecutive points with one key, one composite per run into the bin of its bucket (4096 buckets)
//   2. k_pt_bucket   per bucket: sorted runs (voxel, start) + point offsets + the compacted work list of live roots
//   3. k_roots<1>    one wavefront per root voxel, lanes = (copy, level, moment): the root's points are streamed IN TIME ORDER
//                    through per-node accumulators, so every sum is formed in exactly the order the reference forms it
//                    (bit-identical moments => bit-identical gate decisions); open-cluster sums become candidate
//                    slots when the gap rule fires (ClusterSurfels, cc:22-29)
//   4. k_roots_emit  three roots per wavefront: node tests + candidate clusters as ONE batch of 3x3 Jacobi PCAs, gates,
//                    view-point flip, surfel into its slot, time key into its time bin; roots whose layer-1 nodes
//                    split are queued
//   5. k_roots<2>    layer 2 of the queued roots (launched only if the previous sweep needed it)
//   6. k_slot_emit   time order (surfel_extraction.cc:334) + copy slot -> caller's buffer, count to the host mailbox
// Fallbacks, chosen by finish() from the flags: radix-sort path (k_keygen + rocPRIM + k_heads + k_roots_banks) for
// sweeps without run structure or with overfull bins, 21-bit keys for wide sweeps, radix sort of the slot keys.
// The path is HBM/latency bound (20 B read per point, 144 B written per surfel); no MFMA.
#include <hip/hip_runtime.h>
#include <type_traits>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "ctx.h"
#include "dmath.h"

namespace {

constexpr int kMom = 11;    // n, St, Sx, Sy, Sz, Sxx, Sxy, Sxz, Syy, Syz, Szz
constexpr int kProdStride = 66;  // row stride (doubles) of k_roots' moment-major staging table: even (16-byte rows), not a multiple of 16
constexpr unsigned kRootsGrid = 256 * 16;  // wavefronts of the layer-0/1 pass (one root each from the dense work list; all resident)
constexpr unsigned kEmitGrid = 256 * 8;   // wavefronts of the node-test + emission pass (three roots at a time each)
constexpr unsigned kRoots2Grid = 256 * 4;  // wavefronts of the (rare) layer-2 pass
constexpr int kBuckets = 4096;    // buckets of the composite sorts
constexpr int kTile = 2048;       // points per workgroup of k_pt_runs
constexpr uint32_t kFlagKeyRange = 1u, kFlagSlotOverflow = 2u, kFlagTimeRange = 4u, kFlagBucketOverflow = 8u, kFlagSlotBinOverflow = 16u, kFlagLdsOverflow = 32u;

struct ExParams {
  double vs;           // (double)voxel_size
  float vs_f;          // voxel_size as float
  int max_layer;
  int min_points;
  double thr;          // (double)planer_threshold
  double min_like;
  double view[3];
  double gap;
  int cluster_min;
  uint64_t t_lo_bits;  // ordered bits of the time hint lower bound
  uint64_t t_span_bits;  // ordered bits of the upper bound - t_lo_bits
  int dbg;             // development option debug_skip: knock-out bits (profiling experiments only)
  int merge_min;       // a (node, time slot) list of more records than this makes the next sweep run k_fx_merge (default 3; WC_FX_MERGE_MIN: experiments)
};

__device__ __forceinline__ uint64_t ordered_bits(double t) {
  uint64_t u = (uint64_t)__double_as_longlong(t);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
inline uint64_t ordered_bits_host(double t) {
  uint64_t u;
  memcpy(&u, &t, 8);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__device__ __forceinline__ void load_xyz(const wc_points &pts, uint64_t i, double &x, double &y, double &z) {
  const float *f = (const float *)((const char *)pts.xyz + i * pts.xyz_stride);
  x = (double)f[0];
  y = (double)f[1];
  z = (double)f[2];
}
__device__ __forceinline__ double load_t(const wc_points &pts, uint64_t i) {
  return *(const double *)((const char *)pts.time + i * pts.time_stride);
}
// VoxelLoc (surfel_extraction.h:59-64): floor(pos / resolution) cast to int32, true fp64 division
__device__ __forceinline__ int vox(double p, double vs) { return (int)floor(p / vs); }

template <typename K>
struct KeyTraits;
template <>
struct KeyTraits<uint32_t> {
  static constexpr int bits = 10;
};
template <>
struct KeyTraits<uint64_t> {
  static constexpr int bits = 21;
};

// Flags and the final counts reach the host without a copy kernel: status words [8, 9] hold the device-visible address of
// the context's pinned host mailbox (written by k_init); a flag is raised in HBM (for the kernels that test it) AND as a
// plain store of 1 into mailbox word 8 + log2(flag) (idempotent, no PCIe atomic); k_slot_emit leaves the surfel count and
// the layer-2 queue length in mailbox words 0 and 4.  The 32-byte hipMemcpyAsync this replaces was a 4 us blit kernel.
__device__ __forceinline__ uint32_t *host_mailbox(const uint32_t *status) {
  return (uint32_t *)(((unsigned long long)status[9] << 32) | (unsigned long long)status[8]);
}
__device__ __forceinline__ void raise_flag(uint32_t *status, uint32_t flag) {
  atomicOr(&status[1], flag);
  uint32_t *hm = host_mailbox(status);
  if (hm) hm[8 + (__ffs((int)flag) - 1)] = 1u;
}

template <typename K>
__global__ void __launch_bounds__(256) k_keygen(wc_points pts, double vs, K *keys, uint32_t *vals, uint32_t *status) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pts.n) return;
  constexpr int B = KeyTraits<K>::bits;
  constexpr int half = 1 << (B - 1);
  double x0, y0, z0, x, y, z;
  load_xyz(pts, 0, x0, y0, z0);
  load_xyz(pts, i, x, y, z);
  int rx = vox(x, vs) - vox(x0, vs) + half;
  int ry = vox(y, vs) - vox(y0, vs) + half;
  int rz = vox(z, vs) - vox(z0, vs) + half;
  if ((unsigned)rx >= (unsigned)(2 * half) || (unsigned)ry >= (unsigned)(2 * half) || (unsigned)rz >= (unsigned)(2 * half)) {
    raise_flag(status, kFlagKeyRange);
    rx = min(max(rx, 0), 2 * half - 1);
    ry = min(max(ry, 0), 2 * half - 1);
    rz = min(max(rz, 0), 2 * half - 1);
  }
  keys[i] = (K)rx | ((K)ry << B) | ((K)rz << (2 * B));
  vals[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(256) k_voxel_keys(wc_points pts, double vs, int32_t *out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pts.n) return;
  double x, y, z;
  load_xyz(pts, i, x, y, z);
  out[3 * i + 0] = vox(x, vs);
  out[3 * i + 1] = vox(y, vs);
  out[3 * i + 2] = vox(z, vs);
}

struct Pca {
  double c[3];
  double cov[9];
  double ev[3];
  double nrm[3];  // eigenvector of the smallest eigenvalue (unflipped)
  double tmean;
  double like;
};

// moments -> mean, un-centred population covariance, eigen-decomposition
// (surfel_extraction.cc:36-51 and :89-101)
__device__ __forceinline__ void pca_from_moments(const double *m, Pca &r) {
  const double n = m[0];
  r.tmean = m[1] / n;
  r.c[0] = m[2] / n, r.c[1] = m[3] / n, r.c[2] = m[4] / n;
  const double sxx = m[5], sxy = m[6], sxz = m[7], syy = m[8], syz = m[9], szz = m[10];
  wc::M3 C;
  C.m[0][0] = sxx / n - r.c[0] * r.c[0];
  C.m[0][1] = sxy / n - r.c[0] * r.c[1];
  C.m[0][2] = sxz / n - r.c[0] * r.c[2];
  C.m[1][0] = sxy / n - r.c[1] * r.c[0];
  C.m[1][1] = syy / n - r.c[1] * r.c[1];
  C.m[1][2] = syz / n - r.c[1] * r.c[2];
  C.m[2][0] = sxz / n - r.c[2] * r.c[0];
  C.m[2][1] = syz / n - r.c[2] * r.c[1];
  C.m[2][2] = szz / n - r.c[2] * r.c[2];
  wc::M3 V;
  wc::eig3_sym(C, r.ev, V);
  for (int i = 0; i < 3; ++i) {
    r.nrm[i] = V.m[i][0];
    for (int j = 0; j < 3; ++j) r.cov[3 * i + j] = C.m[i][j];
  }
  r.like = 2 * (r.ev[1] - r.ev[0]) / ((r.ev[0] + r.ev[1]) + r.ev[2]);
}

// ---- voxel key <-> (bucket digit, rest) and the run composites of the run-binned point sort (see k_pt_runs) ----
__device__ __forceinline__ uint32_t key_digit(uint32_t key) {
  const uint32_t x = key & 1023u, y = (key >> 10) & 1023u, z = key >> 20;
  return (x & 15u) | ((y & 15u) << 4) | ((z & 15u) << 8);
}
__device__ __forceinline__ uint32_t key_rest(uint32_t key) {
  const uint32_t x = key & 1023u, y = (key >> 10) & 1023u, z = key >> 20;
  return (x >> 4) | ((y >> 4) << 6) | ((z >> 4) << 12);
}
__device__ __forceinline__ uint32_t key_join(uint32_t d, uint32_t r) {
  const uint32_t x = (d & 15u) | ((r & 63u) << 4), y = ((d >> 4) & 15u) | (((r >> 6) & 63u) << 4),
                 z = ((d >> 8) & 15u) | (((r >> 12) & 63u) << 4);
  return x | (y << 10) | (z << 20);
}

// run composite: key rest (18 bits) << 45 | start index (32 bits) << 13 | run length - 1 (13 bits): sorting the composites
// sorts by (voxel, start), and the length rides along (no second array, no dependent gather in k_pt_bucket)
__device__ __forceinline__ uint64_t run_comp(uint32_t rest, uint32_t start, uint32_t len) {
  return ((uint64_t)rest << 45) | ((uint64_t)start << 13) | (uint64_t)(len - 1);
}
__device__ __forceinline__ uint32_t comp_rest(uint64_t c) { return (uint32_t)(c >> 45); }
__device__ __forceinline__ uint32_t comp_start(uint64_t c) { return (uint32_t)(c >> 13); }
__device__ __forceinline__ uint32_t comp_len(uint64_t c) { return ((uint32_t)c & 8191u) + 1u; }
static_assert(kTile <= 8192, "run length field");

// a live root voxel (more than min_points points).  pos = rank of its first point in the voxel-sorted order (names the
// head slot and the candidate slot range).  RUNS mode (run-binned point sort): the root's points are nr sorted runs
// starting at runs[gidx], total points; otherwise they are keys/vals[pos ...] while the key stays the same.
struct HeadRec {
  uint32_t pos, gidx, nr, total;
};
struct SplitJob;
struct RootsArgs {
  wc_points pts;
  ExParams P;
  uint64_t n;
  const uint32_t *vals;    // sorted point indices
  const HeadRec *heads;    // head slot table: the live root whose first sorted position is pos sits in slot pos / (min_points + 1)
  const uint64_t *runs;    // RUNS mode: the sorted run composites of every bucket, [kBuckets][run_cap]
  const uint32_t *run_off; // RUNS mode: bucket-local point offset of every sorted run
  uint32_t run_cap;
  const uint32_t *root_cnt;    // RUNS mode: live roots found by workgroup w of k_pt_bucket ...
  const uint32_t *root_first;  // ... compacted into head table slots [root_first[w], root_first[w] + root_cnt[w])
  uint32_t nslots;
  double *cand;            // [total_slots][11] candidate cluster moments
  uint32_t *cand_meta;     // [total_slots] local node | phase << 7 | ordinal << 8
  wc_surfel *slots;        // [total_slots]
  wc_surfel_id *slot_ids;  // [total_slots]
  uint64_t *slot_keys;     // [total_slots] time sort keys (memset to ~0 = invalid)
  uint64_t total_slots;
  uint32_t *status;        // [0] emitted count, [1] flags
  uint32_t *slot_counts;   // bucket histogram of the surfel time keys (fast slot sort), or null
  uint32_t slot_shift;
  uint64_t *slot_bins;          // [kBuckets][slot_bin_cap] (time key << 32 | slot) of the surfels of each time bucket
  uint32_t slot_bin_cap;
  struct SplitJob *split_jobs;  // roots queued for the layer-2 pass; count in status[4]
  uint32_t *prof;               // WC_PROF_ROOTS builds: 8 section timers per head slot
  double *node_tot;             // [head slot][9][11] node totals of the layer-0/1 pass
  uint32_t *root_ncand;         // [head slot] candidates written by the layer-0/1 pass
};

// a candidate cluster that passed every gate becomes a surfel (ClusterSurfels' second loop, cc:54-64): normal towards the
// view point, record + id into the candidate's slot, and the time key into the fast slot order (or the key array)
__device__ __forceinline__ void emit_surfel(const RootsArgs &A, const Pca &rr, uint64_t slot, int layer, int nu, uint32_t ord, int kx, int ky,
                                          int kz, float q0) {
  const ExParams &P = A.P;
  double nx = rr.nrm[0], ny = rr.nrm[1], nz = rr.nrm[2];
  const double d = nx * (rr.c[0] - P.view[0]) + ny * (rr.c[1] - P.view[1]) + nz * (rr.c[2] - P.view[2]);
  if (d < 0) nx = -nx, ny = -ny, nz = -nz;  // cc:59-61
  const float ql = layer == 0 ? q0 : (layer == 1 ? q0 / 2 : (q0 / 2) / 2);
  wc_surfel sf;
  sf.t = rr.tmean;
  sf.center[0] = rr.c[0], sf.center[1] = rr.c[1], sf.center[2] = rr.c[2];
  for (int i = 0; i < 9; ++i) sf.cov[i] = rr.cov[i];
  sf.normal[0] = nx, sf.normal[1] = ny, sf.normal[2] = nz;
  sf.resolution = (double)(ql * 4);  // quarter_length_ * 4, float arithmetic (cc:307)
  sf.sigma = sqrt(rr.ev[0]);
  A.slots[slot] = sf;
  uint32_t node = (uint32_t)layer;
  if (layer == 1) node |= (uint32_t)(nu - 1) << 2;
  if (layer == 2) node |= ((uint32_t)(nu >> 3) << 2) | ((uint32_t)(nu & 7) << 5);
  A.slot_ids[slot] = wc_surfel_id{kx, ky, kz, node | (ord << 8)};
  const uint64_t ob = ordered_bits(rr.tmean);
  uint64_t key;
  if (ob < P.t_lo_bits) {
    raise_flag(A.status, kFlagTimeRange);
    key = 0;
  } else {
    key = ob - P.t_lo_bits;
    if (key > P.t_span_bits) raise_flag(A.status, kFlagTimeRange);  // above the hint: the time order would be wrong
  }
  if (A.slot_counts) {  // fast slot order: drop the surfel into its time bucket right here (k_slot_emit sorts each bucket)
    if (key >> 32) raise_flag(A.status, kFlagTimeRange);
    const uint32_t bkt = min((uint32_t)(key >> A.slot_shift), (uint32_t)(kBuckets - 1));
    const uint32_t r = atomicAdd(&A.slot_counts[bkt], 1u);
    if (r < A.slot_bin_cap) A.slot_bins[(size_t)bkt * A.slot_bin_cap + r] = (key << 32) | (uint64_t)(uint32_t)slot;
  } else {
    A.slot_keys[slot] = key;
  }
}

// Heads of the root-voxel segments that can emit anything (n > min_points, InitOctoTree cc:129).  Two live heads are at
// least min_points + 1 positions apart, so slot = pos / (min_points + 1) is collision free: no atomics, no compaction
// (a single append counter serialised at ~12 ns per live root: 46 us for 3.9 k roots).
template <typename K>
__global__ void __launch_bounds__(256) k_heads(const K *__restrict__ keys, uint64_t n, int min_points, HeadRec *head_slots) {
  const uint64_t pos = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= n) return;
  const K k = keys[pos];
  const bool live = (pos == 0 || keys[pos - 1] != k) && (pos + (uint64_t)min_points < n) && keys[pos + min_points] == k;
  if (live) head_slots[pos / (uint64_t)(min_points + 1)] = HeadRec{(uint32_t)pos, 0u, 0u, 0u};
}

#ifdef WC_PROF_ROOTS
// per-root section timers (cycle counter), kept in registers and written once per root to a private row: no shared
// counters (a hot atomic would sit in the same in-order memory queue as the loads being timed)
#define WC_TICK(i)                                                  \
  do {                                                              \
    const unsigned long long now_ = __builtin_readcyclecounter();   \
    prof_[i] += (uint32_t)(now_ - tick_);                           \
    tick_ = now_;                                                   \
  } while (0)
#else
#define WC_TICK(i)
#endif

struct SplitJob {  // a root whose layer-1 nodes need the layer-2 pass (k_roots<K, 2>)
  uint32_t slot, ncand;  // head table slot of the root
  unsigned long long split1;
};

// RUNS mode work list: k_pt_bucket leaves the live roots of its workgroup w compacted in the head table (root_first /
// root_cnt, kBuckets / 4 workgroups).  Every consumer wavefront scans the 1024 counts once (4 KB out of L2) and can then
// turn a global root number into a head table slot: exactly one root (or three, k_roots_emit) per wavefront, instead
// of whatever number of heads happens to fall into a fixed range of table slots (0, 1 or 2: the 2s set the kernel time).
struct RootLocator {
  static constexpr int PER = (kBuckets / 4) / 64;
  uint32_t c[PER], f[PER];
  uint32_t ex, total;
  __device__ __forceinline__ void init(const RootsArgs &A, int lane) {
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      c[k] = A.root_cnt[lane * PER + k];
      f[k] = A.root_first[lane * PER + k];
      sum += c[k];
    }
    uint32_t inc = sum;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = __shfl_up(inc, off);
      if (lane >= off) inc += v;
    }
    ex = inc - sum;
    total = __shfl(inc, 63);
  }
  // head table slot of root number w (wavefront-uniform, w < total)
  __device__ __forceinline__ uint32_t slot_of(const RootsArgs &A, uint32_t w) const {
    const unsigned long long mask = __ballot(ex <= w);  // ex is non-decreasing over the lanes
    const int owner = 63 - __clzll((long long)mask);
    uint32_t rel = w - (uint32_t)__builtin_amdgcn_readlane((int)ex, owner);
    uint32_t first = 0;
    bool found = false;
#pragma unroll
    for (int k = 0; k < PER; ++k) {  // no early exit: everything stays in registers (no dependent load for root_first)
      const uint32_t ck = (uint32_t)__builtin_amdgcn_readlane((int)c[k], owner);
      const uint32_t fk = (uint32_t)__builtin_amdgcn_readlane((int)f[k], owner);
      if (!found) {
        if (rel < ck)
          first = fk, found = true;
        else
          rel -= ck;
      }
    }
    return first + rel;
  }
};

// One wavefront per root voxel.  PHASE 1 streams the root + its eight layer-1 nodes (9 table rows) and leaves node totals
// and candidate clusters to k_roots_emit; roots with layer-1 nodes that were tested and are not planes (CutOctoTree
// recursion, cc:175-182) are queued for PHASE 2, a second launch of the same streaming code over the 64 layer-2 nodes with
// the tests and the emission fused in (rare on regular scenes, so the common case keeps a 7 KB LDS footprint).
// Work assignment: RUNS mode takes root number blockIdx.x of the dense work list (RootLocator); the radix-sort path scans a
// fixed range of the sparse head table.  (A device-side dequeue word serialised at ~90 dequeues/us and cost more than it
// balanced.)
template <typename K, int PHASE, bool RUNS>
__global__ void __launch_bounds__(64) k_roots(RootsArgs A, const K *__restrict__ keys) {
  constexpr int phase = PHASE;  // octree pass: 1 = root + layer 1 (streaming only, k_roots_emit finishes it), 2 = layer 2 (fused)
  constexpr int ntab = (phase == 1) ? 9 : 64;
  __shared__ double s_open[ntab * kMom];
  __shared__ double s_total[ntab * kMom];
  __shared__ double s_last[ntab];
  __shared__ int s_cnt[ntab];
  __shared__ uint32_t s_ord[ntab];
  // the 11 moment terms {1, t, x, y, z, xx, xy, xz, yy, yz, zz} of every staged point, MOMENT-major (row m = term m of the 64
  // points, kProdStride doubles apart): the lane of term m reads two consecutive points with one 16-byte LDS read (an LDS
  // read costs ~3.3 clk per wavefront whatever its width, profiles/micro/lds.hip).  Measured: half the read instructions
  // do not shorten the sequential pass (87 clk per point with 15 wavefronts per CU streaming) - it is bound by issuing
  // two dependent fp64 additions per point and wavefront, which only two roots per wavefront would halve.
  __shared__ __attribute__((aligned(16))) double s_prod[kMom * kProdStride];
  __shared__ uint32_t s_code[64];

  const int lane = threadIdx.x;
  const ExParams &P = A.P;
  constexpr int B = KeyTraits<K>::bits;
  constexpr int half = 1 << (B - 1);
  const uint32_t njobs = (PHASE == 2) ? A.status[4] : 0u;
  if (PHASE == 2 && blockIdx.x >= njobs) return;  // usually nothing is queued: leave before anything else is loaded
  // (no status check: after a bin overflow the run structures are incomplete but consistent - the roots of the overflowed
  // bucket are simply missing - and the host reruns the general path; a check would put one more dependent load in
  // front of every wavefront)

  // lane roles while streaming: lane = (copy, level, moment).  Copy 0 accumulates the OPEN cluster of its node, copy 1 the
  // node TOTAL: the two sums see the same terms but restart at different times, and as two additions per lane and point
  // they were what the sequential pass issued (it is bound by exactly that); side by side they are one addition.
  // Everything else (current node, open count, last time stamp) is mirrored by both copies.
  const int nlev = (phase == 1) ? (P.max_layer >= 1 ? 2 : 1) : 1;  // levels: 0,1 in phase 1; phase 2 streams one level
  const int lgrp = nlev * kMom;
  const bool is_tot = lane >= lgrp && lane < 2 * lgrp;
  const int lane2 = is_tot ? lane - lgrp : lane;
  const int Lq = lane2 / kMom;
  const int m = lane2 - Lq * kMom;
  const bool act = lane < 2 * lgrp;

  double x0, y0, z0;
  load_xyz(A.pts, 0, x0, y0, z0);
  const int k0x = vox(x0, P.vs), k0y = vox(y0, P.vs), k0z = vox(z0, P.vs);

  // ---- one root: stream its points (PHASE 2: also test and emit) ----
  auto do_root = [&](const HeadRec hrec, uint32_t tslot, uint32_t ncand, unsigned long long split1) {
    uint32_t emitted = 0;
    const uint64_t head = hrec.pos;
#ifdef WC_PROF_ROOTS
    unsigned long long tick_ = __builtin_readcyclecounter();
    uint32_t prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // RUNS: lane r < 64 keeps run r of this root (start index, root-local offset) in registers
    uint32_t r_start = 0, r_off = 0;
    K rootkey;
    if (RUNS) {
      uint64_t comp = 0;
      if ((uint32_t)lane < min(hrec.nr, 64u)) {
        comp = A.runs[(size_t)hrec.gidx + lane];
        r_off = A.run_off[(size_t)hrec.gidx + lane];
      }
      r_start = comp_start(comp);
      r_off -= (uint32_t)__builtin_amdgcn_readfirstlane((int)r_off);
      rootkey = (K)key_join(hrec.gidx / A.run_cap, (uint32_t)__builtin_amdgcn_readfirstlane((int)comp_rest(comp)));
    } else {
      rootkey = keys[head];
    }
    // index of the root's p-th point (time order)
    // (called by all lanes: the register path broadcasts run descriptors with readlane)
    auto point_index = [&](uint32_t pp, bool ok_lane) -> uint32_t {
      if (!RUNS) return ok_lane ? A.vals[head + pp] : 0u;
      if (hrec.nr <= 64u) {
        uint32_t st = (uint32_t)__builtin_amdgcn_readfirstlane((int)r_start), o = 0;
        for (uint32_t k = 1; k < hrec.nr; ++k) {
          const uint32_t ok = (uint32_t)__builtin_amdgcn_readlane((int)r_off, (int)k);
          const uint32_t sk = (uint32_t)__builtin_amdgcn_readlane((int)r_start, (int)k);
          if (pp >= ok) st = sk, o = ok;
        }
        return st + (pp - o);
      }
      if (!ok_lane) return 0u;
      const uint32_t *ro = A.run_off + hrec.gidx;  // many runs (unordered input): binary search in the offset table
      const uint32_t o0 = ro[0];
      uint32_t lo = 0, hi = hrec.nr;
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (ro[mid] - o0 <= pp)
          lo = mid;
        else
          hi = mid;
      }
      return comp_start(A.runs[(size_t)hrec.gidx + lo]) + (pp - (ro[lo] - o0));
    };

    // absolute root voxel index and centre ((0.5 + k) * voxel_size, cc:208-210)
    const int kx = (int)(rootkey & ((K(1) << B) - 1)) - half + k0x;
    const int ky = (int)((rootkey >> B) & ((K(1) << B) - 1)) - half + k0y;
    const int kz = (int)((rootkey >> (2 * B)) & ((K(1) << B) - 1)) - half + k0z;
    const double cx = (0.5 + kx) * P.vs_f, cy = (0.5 + ky) * P.vs_f, cz = (0.5 + kz) * P.vs_f;
    const float q0 = P.vs_f / 4;  // quarter_length_ of the root (cc:207)
    const uint64_t slot_base = (head * (uint64_t)(P.max_layer + 1)) / (uint64_t)P.cluster_min;

    const uint32_t cand_begin = ncand;
    {
    for (int i = lane; i < ntab * kMom; i += 64) {
      s_open[i] = 0.0;
      s_total[i] = 0.0;
    }
    for (int i = lane; i < ntab; i += 64) {
      s_last[i] = 0.0;
      s_cnt[i] = 0;
      s_ord[i] = 0;
    }

    // register-cached accumulators of the node this lane is currently feeding
    int cur = -1, n_open = 0;
    double a_acc = 0.0, last = 0.0;  // this lane's sum: the open cluster (copy 0) or the node total (copy 1)

    // ---- software-pipelined chunk loop: the next 64 points are in flight while the current ones stream ----
    uint32_t pp = lane;  // position inside the root
    bool valid = RUNS ? (pp < hrec.total) : (head + pp < A.n && keys[head + pp] == rootkey);
    double px = 0, py = 0, pz = 0, pt = 0;
    uint32_t idx = point_index(pp, valid);
    if (valid) {
      load_xyz(A.pts, idx, px, py, pz);
      pt = load_t(A.pts, idx);
    }
    WC_TICK(0);  // head -> first loads issued
    int carry_o1 = -2;      // layer-1 octant / timestamp of the last point of the previous chunk
    double carry_t = 0.0;
    while (true) {
      const int nvalid = __popcll(__ballot(valid));  // valid lanes are a prefix: keys are sorted
      if (nvalid == 0) break;
      __syncthreads();
      int my_o1 = -1;
      if (valid) {
        // octant = 4*[x>cx] + 2*[y>cy] + [z>cz] (strict >, cc:147-158); child centre = centre +- quarter (cc:163-165)
        const int bx = px > cx, by = py > cy, bz = pz > cz;
        const double c1x = cx + (double)((float)(2 * bx - 1) * q0);
        const double c1y = cy + (double)((float)(2 * by - 1) * q0);
        const double c1z = cz + (double)((float)(2 * bz - 1) * q0);
        const int o1 = 4 * bx + 2 * by + bz;
        const int o2 = 4 * (px > c1x) + 2 * (py > c1y) + (pz > c1z);
        // the moment terms are formed here, lane-parallel: the sequential pass below is bound by LDS reads (15 wavefronts
        // per CU stream at once), and one 8-byte read per (point, moment) is half of what two factors would cost
        double *pr = s_prod + lane;
        pr[0] = 1.0, pr[kProdStride] = pt, pr[2 * kProdStride] = px, pr[3 * kProdStride] = py, pr[4 * kProdStride] = pz;
        pr[5 * kProdStride] = px * px, pr[6 * kProdStride] = px * py, pr[7 * kProdStride] = px * pz, pr[8 * kProdStride] = py * py,
                         pr[9 * kProdStride] = py * pz, pr[10 * kProdStride] = pz * pz;
        s_code[lane] = (uint32_t)(o1 * 8 + o2);
        my_o1 = o1;
      }
      // EVENT points (lane-parallel, before the sequential pass): a point needs the full per-point logic only when its
      // layer-1 node differs from its predecessor's or the time gap to its predecessor exceeds cluster_gap; between
      // two events every level keeps feeding the same node and no cluster can close, so the sequential pass can run a
      // bare multiply-accumulate loop over whole segments.  (Phase 2 skips points, so there every point is an event.)
      unsigned long long ev = ~0ull;
      if (phase == 1) {
        int po1 = __shfl_up(my_o1, 1);
        double ptv = __shfl_up(pt, 1);
        if (lane == 0) {
          po1 = carry_o1;
          ptv = carry_t;
        }
        ev = __ballot(valid && (my_o1 != po1 || pt - ptv > P.gap));
        carry_o1 = __shfl(my_o1, nvalid - 1);
        carry_t = __shfl(pt, nvalid - 1);
      }
      __syncthreads();
      // prefetch the next chunk
      bool nvalid_next = false;
      if (nvalid == 64) {
        pp += 64;
        nvalid_next = RUNS ? (pp < hrec.total) : (head + pp < A.n && keys[head + pp] == rootkey);
        idx = point_index(pp, nvalid_next);
        if (nvalid_next) {
          load_xyz(A.pts, idx, px, py, pz);
          pt = load_t(A.pts, idx);
        }
      }

      WC_TICK(1);  // staging (incl. waiting for the gather)
      // ---- stream the staged points in time order ----
      int j = (WC_DBG(P, 1)) ? nvalid : 0;
      while (j < nvalid) {
        const unsigned long long rem = ev >> j;
        const int je = rem ? j + (__ffsll((long long)rem) - 1) : nvalid;  // next event (or end of chunk)
        if (je > j) {  // event-free segment: every active lane keeps accumulating into its cached node
          if (act) {
            double ao = a_acc;
            const double *sp = s_prod + m * kProdStride;
            int q = j;
            if (q & 1) ao += sp[q++];  // (16-byte reads start at even points)
            for (; q + 8 <= je; q += 8) {  // (all four reads first, then the eight dependent additions)
              const double2 v0 = *(const double2 *)(sp + q), v1 = *(const double2 *)(sp + q + 2), v2 = *(const double2 *)(sp + q + 4),
                            v3 = *(const double2 *)(sp + q + 6);
              ao += v0.x, ao += v0.y, ao += v1.x, ao += v1.y, ao += v2.x, ao += v2.y, ao += v3.x, ao += v3.y;
            }
            for (; q + 2 <= je; q += 2) {
              const double2 v = *(const double2 *)(sp + q);
              ao += v.x;
              ao += v.y;
            }
            if (q < je) ao += sp[q];
            a_acc = ao;
            n_open += je - j;
            last = s_prod[kProdStride + (je - 1)];
          }
          j = je;
          if (j >= nvalid) break;
        }
        // ---- event point: full logic ----
        const uint32_t code = s_code[j];
        const double t = s_prod[kProdStride + j];
        const double v_ev = s_prod[m * kProdStride + j];
        ++j;
        if (phase == 2 && !((split1 >> (code >> 3)) & 1ull)) continue;  // parent layer-1 node is not split
        const int nu = (phase == 2) ? (int)code : (Lq == 0 ? 0 : 1 + (int)(code >> 3));
        if (act && nu != cur) {  // switch node: write the cached accumulators back, fetch the new node's
          double *tab = is_tot ? s_total : s_open;
          if (cur >= 0) {
            tab[cur * kMom + m] = a_acc;
            if (m == 0 && !is_tot) {
              s_last[cur] = last;
              s_cnt[cur] = n_open;
            }
          }
          a_acc = tab[nu * kMom + m];
          last = s_last[nu];
          n_open = s_cnt[nu];
          cur = nu;
        }
        // a new cluster starts when the gap to the previous point OF THIS NODE exceeds cluster_gap (cc:24)
        const bool close = act && n_open > 0 && (t - last > P.gap);
        const unsigned long long cm = __ballot(close);
        if (cm) {
          for (int l = 0; l < nlev; ++l) {
            if (!((cm >> (l * kMom)) & 1ull)) continue;
            const int cnt_l = __shfl(n_open, l * kMom);
            const bool mine = act && (Lq == l);
            if (cnt_l >= P.cluster_min) {  // clusters with fewer points are dropped (cc:33)
              const uint64_t slot = slot_base + ncand;
              if (slot < A.total_slots) {
                if (mine && !is_tot) {
                  A.cand[slot * kMom + m] = a_acc;
                  if (m == 0) A.cand_meta[slot] = (uint32_t)nu | ((uint32_t)(phase - 1) << 7) | (s_ord[nu] << 8);
                }
              } else if (lane == 0) {
                raise_flag(A.status, kFlagSlotOverflow);
              }
              ++ncand;
            }
            if (mine) {
              if (!is_tot) a_acc = 0.0;
              n_open = 0;
              if (m == 0 && !is_tot) s_ord[nu] += 1;
            }
          }
        }
        if (act) {
          a_acc += v_ev;
          n_open += 1;
          last = t;
        }
      }
      WC_TICK(2);  // sequential pass
      if (nvalid < 64) break;
      valid = nvalid_next;
    }
    // write the cached node back
    if (act && cur >= 0) {
      (is_tot ? s_total : s_open)[cur * kMom + m] = a_acc;
      if (m == 0 && !is_tot) {
        s_last[cur] = last;
        s_cnt[cur] = n_open;
      }
    }
    __syncthreads();

    // still-open clusters become candidates too (end of ClusterSurfels' first loop)
    for (int nu = 0; nu < ntab; ++nu) {
      if (s_cnt[nu] >= P.cluster_min) {
        const uint64_t slot = slot_base + ncand;
        if (slot < A.total_slots) {
          if (lane < kMom) A.cand[slot * kMom + lane] = s_open[nu * kMom + lane];
          if (lane == 0) A.cand_meta[slot] = (uint32_t)nu | ((uint32_t)(phase - 1) << 7) | (s_ord[nu] << 8);
        } else if (lane == 0) {
          raise_flag(A.status, kFlagSlotOverflow);
        }
        ++ncand;
      }
    }
    }
    if (PHASE == 1) {  // hand the node totals and the candidate count to k_roots_emit (tests + emission)
      for (int i = lane; i < ntab * kMom; i += 64) A.node_tot[(size_t)tslot * (9 * kMom) + i] = s_total[i];
      if (lane == 0) A.root_ncand[tslot] = ncand;
      WC_TICK(3);
#ifdef WC_PROF_ROOTS
      if (lane == 0 && A.prof)
        for (int i = 0; i < 8; ++i) A.prof[(head / 21) * 8 + i] += prof_[i];
#endif
      __syncthreads();
      return;
    }
    __threadfence_block();
    __syncthreads();
    WC_TICK(3);  // write-back + open-cluster flush + fence
    if (WC_DBG(P, 2)) return;

    // ---- ONE pass of 3x3 PCAs for the node tests (InitOctoTree / CutOctoTree gates, cc:129-138, :170-183) and for
    //      the candidate clusters (ClusterSurfels' second loop, cc:32-64): lanes [0, ntab) take the nodes, the lanes
    //      above them the first candidates, so a root costs one eigen-solve latency instead of two ----
    const uint32_t ncap = (uint32_t)min((uint64_t)ncand, A.total_slots > slot_base ? A.total_slots - slot_base : 0);
    unsigned long long plane_mask = 0;
    for (uint32_t batch = 0;; ++batch) {
      // lane job: node test (first batch only) or candidate
      const bool node_lane = (batch == 0) && lane < ntab;
      const int cand_lane0 = (batch == 0) ? ntab : 0;
      const uint32_t cbase = cand_begin + (batch == 0 ? 0u : (uint32_t)(64 - ntab) + (batch - 1) * 64u);
      const uint32_t c = cbase + (uint32_t)(lane - cand_lane0);
      const bool cand_lane = lane >= cand_lane0 && c < ncap;
      if (batch > 0 && cbase >= ncap) break;
      double mom[kMom];
      bool have = false;
      int nu = 0;
      uint32_t ord = 0;
      uint64_t slot = 0;
      if (node_lane) {
        bool exists = true;
        if (phase == 2) exists = (split1 >> (lane >> 3)) & 1ull;
        nu = lane;
        if (exists && s_total[lane * kMom] > (double)P.min_points) {
          have = true;
          for (int i = 0; i < kMom; ++i) mom[i] = s_total[lane * kMom + i];
        }
      } else if (cand_lane) {
        slot = slot_base + c;
        const uint32_t meta = A.cand_meta[slot];
        nu = (int)(meta & 0x7F);
        ord = meta >> 8;
        have = true;
        for (int i = 0; i < kMom; ++i) mom[i] = A.cand[slot * kMom + i];
      }
      WC_TICK(4);  // moments loaded
      Pca rr;
      if (have && !(WC_DBG(P, 4))) pca_from_moments(mom, rr);
      WC_TICK(5);  // eigen-solves
      if (WC_DBG(P, 4)) { rr.ev[0] = 1e-5; rr.ev[1] = rr.ev[2] = 1e-2; rr.like = 0.9; rr.tmean = mom[1] / mom[0]; for (int i = 0; i < 3; ++i) { rr.c[i] = mom[2 + i] / mom[0]; rr.nrm[i] = 0.577; } for (int i = 0; i < 9; ++i) rr.cov[i] = 0; }
      if (batch == 0) {
        const bool plane = node_lane && have && (rr.ev[0] < P.thr) && (rr.like > P.min_like);  // cc:106-111
        plane_mask = __ballot(plane);
      }
      bool ok = false;
      if (cand_lane && have && ((plane_mask >> nu) & 1ull) && !(rr.ev[0] > P.thr || rr.like < P.min_like)) {  // cc:54
        const int layer = (phase == 2) ? 2 : (nu == 0 ? 0 : 1);
        emit_surfel(A, rr, slot, layer, nu, ord, kx, ky, kz, q0);
        ok = true;
      }
      emitted += (uint32_t)__popcll(__ballot(ok));
      WC_TICK(6);  // gates + surfel stores
    }
    // the surfel count: with the time bins it is the total of the bin counts (k_slot_emit); a per-root atomic on one
    // word serialises at ~12 ns per root once every wavefront reaches this point at the same time (47 us for 3.9 k roots)
    if (lane == 0 && emitted && !A.slot_counts) atomicAdd(&A.status[0], emitted);
#ifdef WC_PROF_ROOTS
    if (lane == 0 && PHASE != 2 && A.prof)
      for (int i = 0; i < 8; ++i) A.prof[(head / 21) * 8 + i] += prof_[i];
#endif
    __syncthreads();
  };

  // ---- work items of this wavefront ----
  if (PHASE == 2) {  // the queued split jobs, strided
    for (uint32_t it = blockIdx.x; it < njobs; it += gridDim.x) {
      const SplitJob job = A.split_jobs[it];
      do_root(A.heads[job.slot], job.slot, job.ncand, job.split1);  // split1 = layer-1 octants that get split (cc:175-182)
    }
  } else if (RUNS) {  // compacted root list: one root per wavefront
    RootLocator loc;
    loc.init(A, lane);
    for (uint32_t w = blockIdx.x; w < loc.total; w += gridDim.x) {
      const uint32_t tslot = loc.slot_of(A, w);
      do_root(A.heads[tslot], tslot, 0u, 0ull);
    }
  } else {  // sparse head table: every wavefront owns a contiguous range of slots
    const uint32_t per_wave = (A.nslots + gridDim.x - 1) / gridDim.x;
    const uint32_t it_end = min((blockIdx.x + 1) * per_wave, A.nslots);
    for (uint32_t it = blockIdx.x * per_wave; it < it_end; it += 64) {
      HeadRec my_head{0xFFFFFFFFu, 0u, 0u, 0u};
      if (it + lane < it_end) my_head = A.heads[it + lane];
      unsigned long long live_mask = __ballot(my_head.pos != 0xFFFFFFFFu);
      while (live_mask) {
        const int hb = __ffsll((long long)live_mask) - 1;
        live_mask &= live_mask - 1;
        do_root(HeadRec{(uint32_t)__shfl((int)my_head.pos, hb), 0u, 0u, 0u}, it + (uint32_t)hb, 0u, 0ull);
      }
    }
  }
}

// ---- streaming for sweeps WITHOUT run structure (the radix-sort path) --------------------------------------------------
// A spinning multi-beam lidar interleaves its beams: consecutive points of a voxel come from different octants, so in
// k_roots above almost every point is an "event" (the cached layer-1 accumulators go through LDS, ~350 clk per point)
// and the largest voxel of the sweep (thousands of points, one wavefront) sets the kernel time.  Here every accumulator
// lives in a register for the whole root: lanes 0..10 carry the 11 moment sums of the LEVEL node (PHASE 1: the root;
// PHASE 2: unused), lanes 11..54 those of the eight CHILD nodes of the pass (PHASE 1: the layer-1 octants; PHASE 2:
// the layer-2 children of ONE split octant per pass over the points): four lane groups of 11, child s in register bank
// A and child s + 4 in bank B.  A point costs one LDS read and two to four EXEC-masked fp64 adds whatever its octant is;
// which lanes add is decided on the scalar unit from ballot masks formed once per 64-point chunk.  Same sums in the same
// order as k_roots, hence the same bits.
__device__ __forceinline__ double readlane_d(double v, int src_lane) {  // src_lane must be wave-uniform: v_readlane_b32 x2
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// a += v, b += v on the lanes of `mask` only, as two EXEC-masked adds (the wavefront is fully active around it)
__device__ __forceinline__ void masked_add2(double &a, double &b, double v, unsigned long long mask_in) {
  // the mask is wave-uniform by construction (ballots, scalar shifts); readfirstlane pins it to SGPRs where the compiler's
  // uniformity analysis gives up (it folds away where the value already is scalar)
  const unsigned long long mask = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(mask_in >> 32)) << 32) |
                                  (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mask_in);
  unsigned long long saved;
  asm volatile(
      "s_mov_b64 %0, exec\n\t"
      "s_mov_b64 exec, %4\n\t"
      "v_add_f64 %1, %1, %3\n\t"
      "v_add_f64 %2, %2, %3\n\t"
      "s_mov_b64 exec, %0"
      : "=&s"(saved), "+v"(a), "+v"(b)
      : "v"(v), "s"(mask));
}

// both banks of a point in one block: (a0, a1) += v on the lanes of mask_a, (b0, b1) += v on the lanes of mask_b
__device__ __forceinline__ void masked_add4(double &a0, double &a1, double &b0, double &b1, double v, unsigned long long mask_a,
                                            unsigned long long mask_b) {
  unsigned long long saved;
  asm volatile(
      "s_mov_b64 %0, exec\n\t"
      "s_mov_b64 exec, %6\n\t"
      "v_add_f64 %1, %1, %5\n\t"
      "v_add_f64 %2, %2, %5\n\t"
      "s_mov_b64 exec, %7\n\t"
      "v_add_f64 %3, %3, %5\n\t"
      "v_add_f64 %4, %4, %5\n\t"
      "s_mov_b64 exec, %0"
      : "=&s"(saved), "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)
      : "v"(v), "s"(mask_a), "s"(mask_b));
}
__device__ __forceinline__ unsigned long long readlane_u64(uint32_t lo, uint32_t hi, int src_lane) {
  return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)hi, src_lane) << 32) |
         (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)lo, src_lane);
}

template <typename K, int PHASE, bool RUNS>
__global__ void __launch_bounds__(64) k_roots_banks(RootsArgs A, const K *__restrict__ keys) {
  constexpr int phase = PHASE;
  constexpr int ntab = (phase == 1) ? 9 : 64;
  __shared__ double s_total[(PHASE == 2) ? 64 * kMom : 1];  // PHASE 2: totals of the 64 layer-2 nodes for the fused tests
  __shared__ double s_prod[64 * kMom];                      // the 11 moment terms of every staged point
  const int lane = threadIdx.x;
  const ExParams &P = A.P;
  constexpr int B = KeyTraits<K>::bits;
  constexpr int half = 1 << (B - 1);
  const uint32_t njobs = (PHASE == 2) ? A.status[4] : 0u;
  if (PHASE == 2 && blockIdx.x >= njobs) return;

  const bool is_lvl = (PHASE == 1) && lane < kMom;
  const int l1 = lane - kMom;
  const int my_slot = (lane >= kMom && lane < 5 * kMom) ? l1 / kMom : 7;  // 7: no child group
  const int m = is_lvl ? lane : (my_slot < 4 ? l1 - my_slot * kMom : 0);
  const bool use_children = (PHASE == 2) || P.max_layer >= 1;
  const unsigned long long lvl_mask = (PHASE == 1) ? ((1ull << kMom) - 1ull) : 0ull;

  double x0, y0, z0;
  load_xyz(A.pts, 0, x0, y0, z0);
  const int k0x = vox(x0, P.vs), k0y = vox(y0, P.vs), k0z = vox(z0, P.vs);

  auto do_root = [&](const HeadRec hrec, uint32_t tslot, uint32_t ncand, unsigned long long split1) __attribute__((always_inline)) {
    uint32_t emitted = 0;
    const uint64_t head = hrec.pos;
    // RUNS: the root's points are hrec.nr sorted runs starting at runs[hrec.gidx] (a sweep without run structure: one run
    // per point, so point p of the root is simply run p; otherwise a binary search in the run offsets)
    const K rootkey = RUNS ? (K)key_join(hrec.gidx / A.run_cap, comp_rest(A.runs[hrec.gidx])) : keys[head];
    const bool unit_runs = RUNS && hrec.nr == hrec.total;
    const uint32_t off0 = (RUNS && !unit_runs) ? A.run_off[hrec.gidx] : 0u;
    auto run_point = [&](uint32_t pp) __attribute__((always_inline)) -> uint32_t {  // RUNS: index of the root's pp-th point, pp < total
      if (unit_runs) return comp_start(A.runs[(size_t)hrec.gidx + pp]);
      const uint32_t *ro = A.run_off + hrec.gidx;
      uint32_t lo = 0, hi = hrec.nr;
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (ro[mid] - off0 <= pp)
          lo = mid;
        else
          hi = mid;
      }
      return comp_start(A.runs[(size_t)hrec.gidx + lo]) + (pp - (ro[lo] - off0));
    };
    const int kx = (int)(rootkey & ((K(1) << B) - 1)) - half + k0x;
    const int ky = (int)((rootkey >> B) & ((K(1) << B) - 1)) - half + k0y;
    const int kz = (int)((rootkey >> (2 * B)) & ((K(1) << B) - 1)) - half + k0z;
    const double cx = (0.5 + kx) * P.vs_f, cy = (0.5 + ky) * P.vs_f, cz = (0.5 + kz) * P.vs_f;
    const float q0 = P.vs_f / 4;  // quarter_length_ of the root (cc:207)
    const uint64_t slot_base = (head * (uint64_t)(P.max_layer + 1)) / (uint64_t)P.cluster_min;
    const uint32_t cand_begin = ncand;

    // a temporal cluster of node nu ends (ClusterSurfels' first loop, cc:22-29): clusters with fewer than cluster_min points
    // are dropped (cc:33), the others become candidates; `cnt_lane` is the m = 0 lane of the node (its sum is the count).
    // (The sums are passed by value: a reference bound to bank A in one call and bank B in another turns into a pointer
    // select after inlining and drags the sums into scratch memory.)
    auto end_cluster = [&](bool lanes, int nu, double ao, uint32_t ord, int cnt_lane) __attribute__((always_inline)) {
      const int cnt = (int)readlane_d(ao, cnt_lane);
      if (cnt >= P.cluster_min) {
        const uint64_t slot = slot_base + ncand;
        if (slot < A.total_slots) {
          if (lanes) {
            A.cand[slot * kMom + m] = ao;
            if (m == 0) A.cand_meta[slot] = (uint32_t)nu | ((uint32_t)(phase - 1) << 7) | (ord << 8);
          }
        } else if (lane == 0) {
          raise_flag(A.status, kFlagSlotOverflow);
        }
        ++ncand;
      }
    };

    // ---- one pass over the root's points.  so < 0 (PHASE 1): every point feeds the level node and the child o1;
    //      so >= 0 (PHASE 2): only the points of layer-1 octant so, child = their layer-2 octant o2.
    //      (The sums are locals of the lambda, returned by value: captured by reference they would stay memory objects.) ----
    struct Totals {
      double a, b;
    };
    auto stream_pass = [&](const int so) __attribute__((always_inline)) -> Totals {
      double aoA = 0.0, atA = 0.0, aoB = 0.0, atB = 0.0;  // open-cluster / node-total sums, banks A and B
      uint32_t ordA = 0, ordB = 0;                        // cluster ordinals of the nodes this lane feeds
      double carry_t = 0.0;   // lane o < 8: time of the last point of child o so far
      bool carry_has = false;
      double prev_t = 0.0;    // time of the previous point of the level node
      bool have_prev = false;
      // two-stage software pipeline over the 64-point chunks: (key, index) of chunk c + 2 and the points of chunk c + 1
      // are in flight while chunk c is processed, so no step of the dependent chain key -> index -> point is waited for
      // inside the loop (the loads are unconditional from clamped positions; validity is decided when they are used)
      uint64_t pos = head + lane;  // (RUNS: only pos - head, the position inside the root, is used)
      const uint64_t last = A.n - 1;
      K kcur = 0, knext = 0;
      uint32_t icur, inext;
      bool valid;
      if (RUNS) {
        valid = (uint32_t)lane < hrec.total;
        icur = valid ? run_point((uint32_t)lane) : 0u;
        inext = ((uint32_t)lane + 64u < hrec.total) ? run_point((uint32_t)lane + 64u) : 0u;
      } else {
        kcur = keys[min(pos, last)];
        icur = A.vals[min(pos, last)];
        knext = keys[min(pos + 64, last)];
        inext = A.vals[min(pos + 64, last)];
        valid = pos < A.n && kcur == rootkey;
      }
      double px = 0, py = 0, pz = 0, pt = 0;
      if (valid) {
        load_xyz(A.pts, icur, px, py, pz);
        pt = load_t(A.pts, icur);
      }
      while (true) {
        const int nvalid = __popcll(__ballot(valid));  // valid lanes are a prefix: keys are sorted
        if (nvalid == 0) break;
        __builtin_amdgcn_wave_barrier();
        int node = 0;
        bool in = false;
        if (valid) {
          // octant = 4*[x>cx] + 2*[y>cy] + [z>cz] (strict >, cc:147-158); child centre = centre +- quarter (cc:163-165)
          const int bx = px > cx, by = py > cy, bz = pz > cz;
          const int o1 = 4 * bx + 2 * by + bz;
          if (so < 0) {
            node = o1;
            in = true;
          } else {
            const double c1x = cx + (double)((float)(2 * bx - 1) * q0);
            const double c1y = cy + (double)((float)(2 * by - 1) * q0);
            const double c1z = cz + (double)((float)(2 * bz - 1) * q0);
            node = 4 * (px > c1x) + 2 * (py > c1y) + (pz > c1z);
            in = (o1 == so);
          }
          double *pr = s_prod + lane * kMom;
          pr[0] = 1.0, pr[1] = pt, pr[2] = px, pr[3] = py, pr[4] = pz;
          pr[5] = px * px, pr[6] = px * py, pr[7] = px * pz, pr[8] = py * py, pr[9] = py * pz, pr[10] = pz * pz;
        }
        // cluster boundaries, lane-parallel: a new cluster starts at a point whose gap to the previous point OF THE SAME
        // NODE exceeds cluster_gap (cc:24).  Predecessor inside the chunk: highest lower lane with the same child.
        const unsigned long long IN = __ballot(in);
        unsigned long long mymask = 0, lanemask = 0;  // lanemask: for lane o < 8 the points of child o (carry update)
#pragma unroll
        for (int o = 0; o < 8; ++o) {
          const unsigned long long mo = __ballot(in && node == o);
          if (node == o) mymask = mo;
          if (lane == o) lanemask = mo;
        }
        const unsigned long long below = mymask & ((1ull << lane) - 1ull);
        const int pred = below ? 63 - __clzll((long long)below) : 0;
        const double t_in = __shfl(pt, pred);
        const double t_ca = __shfl(carry_t, node);
        const bool has_ca = __shfl((int)carry_has, node) != 0;
        const bool g1 = in && (below ? (pt - t_in > P.gap) : (has_ca && pt - t_ca > P.gap));
        const unsigned long long G1 = use_children ? __ballot(g1) : 0ull;
        unsigned long long G0 = 0;
        if (PHASE == 1) {
          double t_pv = __shfl_up(pt, 1);
          bool has_pv = true;
          if (lane == 0) t_pv = prev_t, has_pv = have_prev;
          G0 = __ballot(valid && has_pv && (pt - t_pv > P.gap));
          prev_t = __shfl(pt, nvalid - 1);
          have_prev = true;
        }
        {  // carries for the next chunk
          const int hi = lanemask ? 63 - __clzll((long long)lanemask) : 0;
          const double t_hi = __shfl(pt, hi);
          if (lane < 8 && lanemask) carry_t = t_hi, carry_has = true;
        }
        // lane q: the two EXEC masks of point q - the lanes that add it into bank A (level node | child group, children
        // 0..3) and into bank B (child group, children 4..7); the loop below fetches them with four v_readlane
        unsigned long long kid = (in && use_children) ? (((1ull << kMom) - 1ull) << (kMom + kMom * (node & 3))) : 0ull;
        const unsigned long long mB = (node & 4) ? kid : 0ull;
        const unsigned long long mA = ((PHASE == 1 && valid) ? lvl_mask : 0ull) | ((node & 4) ? 0ull : kid);
        const uint32_t mAlo = (uint32_t)mA, mAhi = (uint32_t)(mA >> 32), mBlo = (uint32_t)mB, mBhi = (uint32_t)(mB >> 32);
        const int code = node;
        __builtin_amdgcn_wave_barrier();
        // the next chunk's points (their key / index arrived during the previous chunk), and key / index of the one after
        bool nvalid_next = false;
        if (nvalid == 64) {
          pos += 64;
          nvalid_next = RUNS ? ((uint32_t)(pos - head) < hrec.total) : (pos < A.n && knext == rootkey);
          if (nvalid_next) {
            load_xyz(A.pts, inext, px, py, pz);
            pt = load_t(A.pts, inext);
          }
          if (RUNS) {
            const uint32_t pn = (uint32_t)(pos - head) + 64u;
            inext = (pn < hrec.total) ? run_point(pn) : 0u;
          } else {
            knext = keys[min(pos + 64, last)];
            inext = A.vals[min(pos + 64, last)];
          }
        }

        // ---- the sequential pass: time order.  Points between two cluster ends run through a branch-free body: four
        //      v_readlane for the masks, one LDS read, four EXEC-masked adds ----
        const unsigned long long GE = G0 | G1;
        unsigned long long todo = (PHASE == 1) ? ((nvalid == 64) ? ~0ull : ((1ull << nvalid) - 1ull)) : IN;
        if (WC_DBG(P, 1)) todo = 0;  // development option debug_skip = 1: profiling experiments only
        while (todo) {
          const unsigned long long evs = GE & todo;
          unsigned long long seg = evs ? (todo & ((evs & (0ull - evs)) - 1ull)) : todo;  // the points in front of the next cluster end
          todo &= ~seg;
          if (PHASE == 1 && seg) {  // every point is in the pass: the segment is a contiguous range, a counted loop does it
            int j = __ffsll((long long)seg) - 1;
            const int je = j + __popcll(seg);
            seg = 0;
            for (; j + 4 <= je; j += 4) {
              const double v0 = s_prod[j * kMom + m], v1 = s_prod[(j + 1) * kMom + m], v2 = s_prod[(j + 2) * kMom + m],
                           v3 = s_prod[(j + 3) * kMom + m];
              masked_add4(aoA, atA, aoB, atB, v0, readlane_u64(mAlo, mAhi, j), readlane_u64(mBlo, mBhi, j));
              masked_add4(aoA, atA, aoB, atB, v1, readlane_u64(mAlo, mAhi, j + 1), readlane_u64(mBlo, mBhi, j + 1));
              masked_add4(aoA, atA, aoB, atB, v2, readlane_u64(mAlo, mAhi, j + 2), readlane_u64(mBlo, mBhi, j + 2));
              masked_add4(aoA, atA, aoB, atB, v3, readlane_u64(mAlo, mAhi, j + 3), readlane_u64(mBlo, mBhi, j + 3));
            }
            for (; j < je; ++j)
              masked_add4(aoA, atA, aoB, atB, s_prod[j * kMom + m], readlane_u64(mAlo, mAhi, j), readlane_u64(mBlo, mBhi, j));
          }
          while (seg) {
            const int j0 = __ffsll((long long)seg) - 1;
            seg &= seg - 1;
            const bool h1 = seg != 0;
            const int j1 = h1 ? __ffsll((long long)seg) - 1 : j0;
            seg &= seg - 1;
            const bool h2 = seg != 0;
            const int j2 = h2 ? __ffsll((long long)seg) - 1 : j0;
            seg &= seg - 1;
            const bool h3 = seg != 0;
            const int j3 = h3 ? __ffsll((long long)seg) - 1 : j0;
            seg &= seg - 1;
            const double v0 = s_prod[j0 * kMom + m], v1 = s_prod[j1 * kMom + m], v2 = s_prod[j2 * kMom + m], v3 = s_prod[j3 * kMom + m];
            masked_add4(aoA, atA, aoB, atB, v0, readlane_u64(mAlo, mAhi, j0), readlane_u64(mBlo, mBhi, j0));
            // (masks of a missing point are forced to zero: nothing is added)
            masked_add4(aoA, atA, aoB, atB, v1, h1 ? readlane_u64(mAlo, mAhi, j1) : 0ull, h1 ? readlane_u64(mBlo, mBhi, j1) : 0ull);
            masked_add4(aoA, atA, aoB, atB, v2, h2 ? readlane_u64(mAlo, mAhi, j2) : 0ull, h2 ? readlane_u64(mBlo, mBhi, j2) : 0ull);
            masked_add4(aoA, atA, aoB, atB, v3, h3 ? readlane_u64(mAlo, mAhi, j3) : 0ull, h3 ? readlane_u64(mBlo, mBhi, j3) : 0ull);
          }
          if (evs) {  // a cluster ends in front of point j: close it, then add the point
            const int j = __ffsll((long long)evs) - 1;
            todo &= ~(1ull << j);
            const int c = __builtin_amdgcn_readlane(code, j);
            const int cs = c & 3;
            const bool bankB = (c >> 2) != 0;
            if (PHASE == 1 && ((G0 >> j) & 1ull)) {
              end_cluster(is_lvl, 0, aoA, ordA, 0);
              if (is_lvl) aoA = 0.0, ordA += 1;
            }
            if ((G1 >> j) & 1ull) {
              const int nu = (PHASE == 1) ? 1 + c : so * 8 + c;
              const bool mine = (my_slot == cs);
              if (bankB) {
                end_cluster(mine, nu, aoB, ordB, kMom + kMom * cs);
                if (mine) aoB = 0.0, ordB += 1;
              } else {
                end_cluster(mine, nu, aoA, ordA, kMom + kMom * cs);
                if (mine) aoA = 0.0, ordA += 1;
              }
            }
            masked_add4(aoA, atA, aoB, atB, s_prod[j * kMom + m], readlane_u64(mAlo, mAhi, j), readlane_u64(mBlo, mBhi, j));
          }
        }
        if (nvalid < 64) break;
        valid = nvalid_next;
      }
      // still-open clusters become candidates too (end of ClusterSurfels' first loop), node order
      if (PHASE == 1) end_cluster(is_lvl, 0, aoA, ordA, 0);
      if (use_children) {
        for (int o = 0; o < 4; ++o) end_cluster(my_slot == o, (PHASE == 1) ? 1 + o : so * 8 + o, aoA, ordA, kMom + kMom * o);
        for (int o = 0; o < 4; ++o) end_cluster(my_slot == o, (PHASE == 1) ? 5 + o : so * 8 + 4 + o, aoB, ordB, kMom + kMom * o);
      }
      return Totals{atA, atB};
    };

    if (PHASE == 1) {
      const Totals t1 = stream_pass(-1);
      double *tot = A.node_tot + (size_t)tslot * (9 * kMom);
      if (is_lvl) tot[m] = t1.a;
      if (my_slot < 4) {
        tot[(1 + my_slot) * kMom + m] = t1.a;
        tot[(5 + my_slot) * kMom + m] = t1.b;
      }
      if (lane == 0) A.root_ncand[tslot] = ncand;
      __builtin_amdgcn_wave_barrier();
      return;
    }
    for (int so = 0; so < 8; ++so) {
      if (!((split1 >> so) & 1ull)) continue;  // only the split layer-1 octants have layer-2 nodes
      const Totals tt = stream_pass(so);
      if (my_slot < 4) {
        s_total[(so * 8 + my_slot) * kMom + m] = tt.a;
        s_total[(so * 8 + my_slot + 4) * kMom + m] = tt.b;
      }
    }
    __threadfence_block();
    __syncthreads();
    // ---- node tests of the 64 layer-2 nodes + their candidate clusters, one batch of 3x3 PCAs (as in k_roots<K, 2>) ----
    const uint32_t ncap = (uint32_t)min((uint64_t)ncand, A.total_slots > slot_base ? A.total_slots - slot_base : 0);
    unsigned long long plane_mask = 0;
    for (uint32_t batch = 0;; ++batch) {
      const bool node_lane = (batch == 0) && lane < ntab;
      const int cand_lane0 = (batch == 0) ? ntab : 0;
      const uint32_t cbase = cand_begin + (batch == 0 ? 0u : (uint32_t)(64 - ntab) + (batch - 1) * 64u);
      const uint32_t c = cbase + (uint32_t)(lane - cand_lane0);
      const bool cand_lane = lane >= cand_lane0 && c < ncap;
      if (batch > 0 && cbase >= ncap) break;
      double mom[kMom];
      bool have = false;
      int nu = 0;
      uint32_t ord = 0;
      uint64_t slot = 0;
      if (node_lane) {
        const bool exists = (split1 >> (lane >> 3)) & 1ull;
        nu = lane;
        if (exists && s_total[lane * kMom] > (double)P.min_points) {
          have = true;
          for (int i = 0; i < kMom; ++i) mom[i] = s_total[lane * kMom + i];
        }
      } else if (cand_lane) {
        slot = slot_base + c;
        const uint32_t meta = A.cand_meta[slot];
        nu = (int)(meta & 0x7F);
        ord = meta >> 8;
        have = true;
        for (int i = 0; i < kMom; ++i) mom[i] = A.cand[slot * kMom + i];
      }
      Pca rr;
      if (have) pca_from_moments(mom, rr);
      if (batch == 0) {
        const bool plane = node_lane && have && (rr.ev[0] < P.thr) && (rr.like > P.min_like);  // cc:106-111
        plane_mask = __ballot(plane);
      }
      bool ok = false;
      if (cand_lane && have && ((plane_mask >> nu) & 1ull) && !(rr.ev[0] > P.thr || rr.like < P.min_like)) {  // cc:54
        emit_surfel(A, rr, slot, 2, nu, ord, kx, ky, kz, q0);
        ok = true;
      }
      emitted += (uint32_t)__popcll(__ballot(ok));
    }
    if (lane == 0 && emitted && !A.slot_counts) atomicAdd(&A.status[0], emitted);
    __syncthreads();
  };

  // ---- work items of this wavefront ----
  if (PHASE == 2) {  // the queued split jobs, strided
    for (uint32_t it = blockIdx.x; it < njobs; it += gridDim.x) {
      const SplitJob job = A.split_jobs[it];
      do_root(A.heads[job.slot], job.slot, job.ncand, job.split1);
    }
  } else if (RUNS) {  // compacted root list: one root per wavefront
    RootLocator loc;
    loc.init(A, lane);
    for (uint32_t w = blockIdx.x; w < loc.total; w += gridDim.x) {
      const uint32_t tslot = loc.slot_of(A, w);
      do_root(A.heads[tslot], tslot, 0u, 0ull);
    }
  } else {  // sparse head table: every wavefront owns a contiguous range of slots
    const uint32_t per_wave = (A.nslots + gridDim.x - 1) / gridDim.x;
    const uint32_t it_end = min((blockIdx.x + 1) * per_wave, A.nslots);
    for (uint32_t it = blockIdx.x * per_wave; it < it_end; it += 64) {
      const uint32_t my_pos = (it + lane < it_end) ? A.heads[it + lane].pos : 0xFFFFFFFFu;
      unsigned long long live_mask = __ballot(my_pos != 0xFFFFFFFFu);
      while (live_mask) {
        const int hb = __ffsll((long long)live_mask) - 1;
        live_mask &= live_mask - 1;
        do_root(HeadRec{(uint32_t)__shfl((int)my_pos, hb), 0u, 0u, 0u}, it + (uint32_t)hb, 0u, 0ull);
      }
    }
  }
}

// Node tests + emission of the layer-0/1 pass, THREE roots per wavefront.  One root offers 9 node tests and a dozen or so
// candidate clusters, i.e. ~20 of 64 lanes for an eigen-solve that costs ~13 k cycles of fp64 VALU issue: at one root per
// wavefront the launch is VALU-issue bound on mostly idle lanes.  Lane = 21 * group + l; in the first batch l < 9 tests
// node l (InitOctoTree / CutOctoTree gates, cc:129-138, :170-183) and l >= 9 takes candidate l - 9 (ClusterSurfels'
// second loop, cc:32-64); later batches (rare) take 21 more candidates per group.
template <typename K, bool RUNS>
__global__ void __launch_bounds__(64) k_roots_emit(RootsArgs A, const K *__restrict__ keys) {
  constexpr int G = 21, NG = 3;
  const int lane = threadIdx.x;
  const ExParams &P = A.P;
  constexpr int B = KeyTraits<K>::bits;
  constexpr int half = 1 << (B - 1);
  const int g = lane / G, l = lane - g * G;
  double x0, y0, z0;
  load_xyz(A.pts, 0, x0, y0, z0);
  const int k0x = vox(x0, P.vs), k0y = vox(y0, P.vs), k0z = vox(z0, P.vs);
  const float q0 = P.vs_f / 4;  // quarter_length_ of the root (cc:207)

  // one pass over up to NG roots; tslot = head table slot of this lane's group's root (~0: none)
  auto do_roots = [&](uint32_t tslot) {
      bool active = g < NG && tslot != 0xFFFFFFFFu;
      HeadRec hrec{0u, 0u, 0u, 0u};
      if (active) hrec = A.heads[tslot];
      const uint64_t head = hrec.pos;
      const uint32_t gidx_or = hrec.gidx;
      const uint32_t ncand = active ? A.root_ncand[tslot] : 0u;  // side tables are indexed by table slot: no wait for hrec
      K rootkey;
      if (RUNS) {
        const uint32_t gidx = gidx_or;
        rootkey = (K)key_join(gidx / A.run_cap, comp_rest(A.runs[gidx]));
      } else {
        rootkey = keys[head];
      }
      const int kx = (int)(rootkey & ((K(1) << B) - 1)) - half + k0x;
      const int ky = (int)((rootkey >> B) & ((K(1) << B) - 1)) - half + k0y;
      const int kz = (int)((rootkey >> (2 * B)) & ((K(1) << B) - 1)) - half + k0z;
      const uint64_t slot_base = (head * (uint64_t)(P.max_layer + 1)) / (uint64_t)P.cluster_min;
      const uint32_t ncap = (uint32_t)min((uint64_t)ncand, A.total_slots > slot_base ? A.total_slots - slot_base : 0);
      uint32_t plane_mask = 0, split1 = 0, emitted = 0;
      for (uint32_t batch = 0;; ++batch) {
        const bool node_lane = active && batch == 0 && l < 9;
        const uint32_t c = (batch == 0) ? (uint32_t)(l - 9) : (uint32_t)(G - 9) + (batch - 1) * G + (uint32_t)l;
        const bool cand_lane = active && (batch > 0 || l >= 9) && c < ncap;
        if (!__ballot(node_lane || cand_lane)) break;
        double mom[kMom];
        bool have = false;
        int nu = 0;
        uint32_t ord = 0;
        uint64_t slot = 0;
        if (node_lane) {
          const double *tot = A.node_tot + (size_t)tslot * (9 * kMom) + l * kMom;
          nu = l;
          if (tot[0] > (double)P.min_points) {
            have = true;
            for (int i = 0; i < kMom; ++i) mom[i] = tot[i];
          }
        } else if (cand_lane) {
          slot = slot_base + c;
          const uint32_t meta = A.cand_meta[slot];
          nu = (int)(meta & 0x7F);
          ord = meta >> 8;
          have = true;
          for (int i = 0; i < kMom; ++i) mom[i] = A.cand[slot * kMom + i];
        }
        Pca rr;
        if (have) pca_from_moments(mom, rr);
        if (batch == 0) {
          const bool plane = node_lane && have && (rr.ev[0] < P.thr) && (rr.like > P.min_like);  // cc:106-111
          const int sh = (g < NG) ? g * G : 0;
          plane_mask = (uint32_t)(__ballot(plane) >> sh) & 0x1FFu;
          const bool root_tested = (__ballot(node_lane && have) >> sh) & 1ull;
          if (!root_tested) active = false;  // n <= min_points: nothing below this root exists (cc:129)
          const uint32_t nonplane = (uint32_t)(__ballot(node_lane && have && !plane) >> sh) & 0x1FFu;
          split1 = (P.max_layer >= 2 && active) ? (nonplane >> 1) : 0u;  // tested layer-1 nodes that are not planes (cc:175-182)
        }
        bool ok = false;
        if (cand_lane && active && have && ((plane_mask >> nu) & 1u) && !(rr.ev[0] > P.thr || rr.like < P.min_like)) {  // cc:54
          emit_surfel(A, rr, slot, nu == 0 ? 0 : 1, nu, ord, kx, ky, kz, q0);
          ok = true;
        }
        emitted += (uint32_t)__popcll(__ballot(ok));
      }
      if (active && l == 0 && split1 != 0) {  // layer-2 pass needed: queue the root for k_roots<K, 2>
        const uint32_t q = atomicAdd(&A.status[4], 1u);
        A.split_jobs[q] = SplitJob{tslot, ncand, (unsigned long long)split1};
      }
      // without the slot histogram (general path) the surfel count is accumulated here
      if (lane == 0 && emitted && !A.slot_counts) atomicAdd(&A.status[0], emitted);
  };

  if (RUNS) {  // compacted root list: roots 3 w .. 3 w + 2
    RootLocator loc;
    loc.init(A, lane);
    for (uint32_t w0 = blockIdx.x * NG; w0 < loc.total; w0 += gridDim.x * NG) {
      uint32_t tslot = 0xFFFFFFFFu;
      for (int k = 0; k < NG; ++k)
        if (w0 + k < loc.total) {
          const uint32_t sl = loc.slot_of(A, w0 + k);
          if (g == k) tslot = sl;
        }
      do_roots(tslot);
    }
  } else {  // sparse head table: every wavefront owns a contiguous range of slots
    const uint32_t per_wave = (A.nslots + gridDim.x - 1) / gridDim.x;
    const uint32_t it_end = min((blockIdx.x + 1) * per_wave, A.nslots);
    for (uint32_t it = blockIdx.x * per_wave; it < it_end; it += 64) {
      const uint32_t my_pos = (it + lane < it_end) ? A.heads[it + lane].pos : 0xFFFFFFFFu;
      unsigned long long live_mask = __ballot(my_pos != 0xFFFFFFFFu);
      while (live_mask) {
        uint32_t tslot = 0xFFFFFFFFu;
        for (int k = 0; k < NG; ++k) {
          if (!live_mask) break;
          const int bit = __ffsll((long long)live_mask) - 1;
          live_mask &= live_mask - 1;
          if (g == k) tslot = it + (uint32_t)bit;
        }
        do_roots(tslot);
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_iota(uint32_t *v, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (uint32_t)i;
}

// slot -> output, 16 bytes per thread (a wc_surfel is 9 x 16 B, a wc_surfel_id 1 x 16 B)
__global__ void __launch_bounds__(256) k_gather(const uint32_t *__restrict__ sorted_slot, const wc_surfel *__restrict__ slots,
                                               const wc_surfel_id *__restrict__ slot_ids, const uint32_t *status,
                                               wc_surfel *out, wc_surfel_id *out_ids, uint64_t cap) {
  if (status[1] & (kFlagBucketOverflow | kFlagKeyRange)) return;
  const uint64_t n = min((uint64_t)status[0], cap);
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t rec = i / 10, part = i - rec * 10;
  if (rec >= n) return;
  const uint32_t s = sorted_slot[rec];
  if (part < 9) {
    const double2 *src = (const double2 *)(slots + s);
    double2 *dst = (double2 *)(out + rec);
    dst[part] = src[part];
  } else if (out_ids) {
    const uint4 *src = (const uint4 *)(slot_ids + s);
    uint4 *dst = (uint4 *)(out_ids + rec);
    *dst = *src;
  }
}

// Surfels with EQUAL time stamps: the reference's std::sort on the stamp (surfel_extraction.cc:334) over an unordered hash
// map leaves their order unspecified; here it is the canonical one of SURVEY Q7 - root voxel index, then node id - on
// every path, so that the output is byte for byte the oracle's.
__device__ __forceinline__ bool surfel_id_less(const wc_surfel_id &a, const wc_surfel_id &b) {
  if (a.kx != b.kx) return a.kx < b.kx;
  if (a.ky != b.ky) return a.ky < b.ky;
  if (a.kz != b.kz) return a.kz < b.kz;
  return a.node < b.node;
}

// radix-sort path: the sort is stable in the slot index; one thread per group of equal keys puts the group (two or three
// slots, rarely) into canonical order
__global__ void __launch_bounds__(256) k_fix_ties(const uint64_t *__restrict__ keys, uint32_t *idx, uint64_t n,
                                                 const wc_surfel_id *__restrict__ slot_ids) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t key = keys[i];
  if (key == ~0ull) return;  // no surfel in the slot
  if ((i > 0 && keys[i - 1] == key) || i + 1 >= n || keys[i + 1] != key) return;
  uint64_t e = i + 1;
  while (e < n && keys[e] == key) ++e;
  for (uint64_t a = i + 1; a < e; ++a) {  // insertion sort
    const uint32_t v = idx[a];
    const wc_surfel_id vid = slot_ids[v];
    uint64_t b = a;
    while (b > i && surfel_id_less(vid, slot_ids[idx[b - 1]])) {
      idx[b] = idx[b - 1];
      --b;
    }
    idx[b] = v;
  }
}

// Fast slot order + gather in one launch.  The emission dropped every surfel into one of 4096 time buckets (a count and a
// fixed-capacity bin per bucket), so what is left is: the exclusive prefix of the counts (every workgroup sums the counts
// in front of its four buckets: 16 KB out of L2, cheaper than a separate scan launch), the order inside a bucket (rank by
// counting, one wavefront per bucket, a handful of surfels each) and the 160-byte copy slot -> output, 16 B per lane.
constexpr int kSlotBinMax = 512;
struct SlotEmitArgs {
  const uint32_t *counts;
  const uint64_t *bins;
  uint32_t bin_cap;
  const wc_surfel *slots;
  const wc_surfel_id *slot_ids;
  uint32_t *status;
  wc_surfel *out;
  wc_surfel_id *out_ids;
  uint64_t cap;
  uint32_t *next_ctrl;
  uint32_t next_words;
};
// (bid / nblk: the workgroup's index and the grid of ITS sweep - the kernel proper, or one sweep's share of a batched launch)
__device__ __forceinline__ void slot_emit_body(const SlotEmitArgs &E, const uint32_t bid, const uint32_t nblk) {
  const uint32_t *__restrict__ counts = E.counts;
  const uint64_t *__restrict__ bins = E.bins;
  const uint32_t bin_cap = E.bin_cap;
  const wc_surfel *__restrict__ slots = E.slots;
  const wc_surfel_id *__restrict__ slot_ids = E.slot_ids;
  uint32_t *status = E.status;
  wc_surfel *out = E.out;
  wc_surfel_id *out_ids = E.out_ids;
  const uint64_t cap = E.cap;
  uint32_t *next_ctrl = E.next_ctrl;
  const uint32_t next_words = E.next_words;
  __shared__ uint64_t s_item[4][kSlotBinMax];
  if (next_ctrl) {  // fast path: the control block of the NEXT sweep (the other of two) is cleared here, off the host's path
    const uint32_t i = bid * 256u + threadIdx.x;
    if (i < next_words) next_ctrl[i] = (i == 8u || i == 9u) ? status[i] : 0u;  // (words 8, 9: the mailbox address)
  }
  __shared__ uint32_t s_sorted[4][kSlotBinMax];
  __shared__ uint32_t s_red[4];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const uint32_t b0 = bid * 4u;
  // everything this workgroup reads before it knows its counts goes out in ONE round of loads: the counts in front of it
  // (exactly bid uint4's, at most four per thread), its own four counts and - speculatively - the first 64 entries
  // of its wavefront's bin.  (A loop of dependent 4-byte loads here was the kernel's critical path: 16 round trips for
  // the last workgroups.)
  static_assert(kBuckets / 4 <= 4 * 256, "four uint4 loads per thread cover the counts");
  const uint4 *c4 = (const uint4 *)counts;
  uint4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t i = (uint32_t)t + 256u * k;
    v[k] = i < bid ? c4[i] : make_uint4(0u, 0u, 0u, 0u);
  }
  const uint4 own = c4[bid];
  const uint64_t *bin = bins + (size_t)(b0 + w) * bin_cap;
  const uint64_t first = (uint32_t)lane < bin_cap ? bin[lane] : 0ull;
  uint32_t part = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) part += v[k].x + v[k].y + v[k].z + v[k].w;
  for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
  if (lane == 0) s_red[w] = part;
  __syncthreads();
  uint32_t base = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  const uint32_t c0 = own.x, c1 = own.y, c2 = own.z, c3 = own.w;
  if (bid == nblk - 1 && t == 0) {
    status[0] = base + c0 + c1 + c2 + c3;  // surfels emitted
    uint32_t *hm = host_mailbox(status);
    if (hm) {
      hm[0] = base + c0 + c1 + c2 + c3;
      uint32_t queued = status[4];  // roots queued for the layer-2 pass (every emitting kernel has finished)
      const uint32_t *fxc = counts + kBuckets;  // fast path: layer-2 node sub-counters (extract_fast.inc)
      for (int j = 0; j < 16; ++j) queued += fxc[(16 + j) * 32];
      hm[4] = queued;
      hm[5] = status[5];  // runs of the sweep (run-binned sort only): tells the host whether the input has run structure
    }
  }
  base += (w > 0 ? c0 : 0u) + (w > 1 ? c1 : 0u) + (w > 2 ? c2 : 0u);
  const uint32_t c = w == 0 ? c0 : (w == 1 ? c1 : (w == 2 ? c2 : c3));
  if (c == 0) return;
  if (c > bin_cap) {
    if (lane == 0) raise_flag(status, kFlagSlotBinOverflow);
    return;
  }
  // equal keys: the time stamps themselves (a key of the fast path is a 32-bit fraction of the sweep), then - equal
  // stamps - the canonical order, not the slot order
  auto before = [&](uint64_t other, uint64_t mine) -> bool {
    if ((other >> 32) == (mine >> 32) && other != mine) {
      const double to = slots[(uint32_t)other].t, tm = slots[(uint32_t)mine].t;
      return to < tm || (to == tm && surfel_id_less(slot_ids[(uint32_t)other], slot_ids[(uint32_t)mine]));
    }
    return other < mine;
  };
  if (c <= 64u) {  // the usual bucket: the items stay in registers, a lane reads the others with v_readlane
    const uint32_t lo = (uint32_t)first, hi = (uint32_t)(first >> 32);
    uint32_t rank = 0;
    for (uint32_t j = 0; j < c; ++j) {  // composites are unique (slot index)
      const uint64_t other = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, (int)j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)lo, (int)j);
      if ((uint32_t)lane < c) rank += before(other, first) ? 1u : 0u;
    }
    if ((uint32_t)lane < c) s_sorted[w][rank] = lo;
  } else {
    s_item[w][lane] = first;
    for (uint32_t i = lane + 64; i < c; i += 64) s_item[w][i] = bin[i];
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < c; i += 64) {
      const uint64_t mine = s_item[w][i];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < c; ++j) rank += before(s_item[w][j], mine) ? 1u : 0u;
      s_sorted[w][rank] = (uint32_t)mine;
    }
  }
  __builtin_amdgcn_wave_barrier();
  const int sub = lane / 10, piece = lane - sub * 10;  // six records per pass, ten 16-byte pieces per record
  if (sub >= 6) return;
  auto load = [&](uint32_t r, bool on) -> double2 {
    double2 d = make_double2(0.0, 0.0);
    if (!on) return d;
    const uint32_t sl = s_sorted[w][r];
    if (piece < 9)
      d = ((const double2 *)(slots + sl))[piece];
    else if (out_ids) {
      const uint4 q = *(const uint4 *)(slot_ids + sl);
      d = *(const double2 *)&q;
    }
    return d;
  };
  auto store = [&](uint32_t r, bool on, const double2 &d) {
    if (!on) return;
    const uint64_t o = (uint64_t)base + r;
    if (piece < 9)
      ((double2 *)(out + o))[piece] = d;
    else if (out_ids)
      *(double2 *)(out_ids + o) = d;
  };
  if (c <= 12u) {  // the usual bucket of a sweep: two passes' loads in flight before the first store
    const uint32_t r = sub;
    const bool on0 = r < c && (uint64_t)base + r < cap, on1 = r + 6 < c && (uint64_t)base + r + 6 < cap;
    const double2 d0 = load(r, on0), d1 = load(r + 6, on1);
    store(r, on0, d0);
    store(r + 6, on1, d1);
    return;
  }
  for (uint32_t r = sub; r < c; r += 24) {  // large clouds (~76 surfels per bucket at 10 M points): four passes in flight
    bool on[4];
    double2 d[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      on[q] = r + 6u * q < c && (uint64_t)base + r + 6u * q < cap;
      d[q] = load(r + 6u * q, on[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) store(r + 6u * q, on[q], d[q]);
  }
}

__global__ void __launch_bounds__(256) k_slot_emit(const uint32_t *__restrict__ counts, const uint64_t *__restrict__ bins, uint32_t bin_cap,
                                                  const wc_surfel *__restrict__ slots, const wc_surfel_id *__restrict__ slot_ids,
                                                  uint32_t *status, wc_surfel *out, wc_surfel_id *out_ids, uint64_t cap, uint32_t *next_ctrl,
                                                  uint32_t next_words) {
  const SlotEmitArgs E{counts, bins, bin_cap, slots, slot_ids, status, out, out_ids, cap, next_ctrl, next_words};
  slot_emit_body(E, blockIdx.x, gridDim.x);
}
// K sweeps' time ordering in one launch: kBuckets / 4 workgroups per sweep
__global__ void __launch_bounds__(256) k_slot_emit_b(const SlotEmitArgs *__restrict__ Es, int K) {
  const uint32_t per = kBuckets / 4;
  const uint32_t k = blockIdx.x / per;
  if ((int)k >= K) return;
  slot_emit_body(Es[k], blockIdx.x - k * per, per);
}

// ---- run-binned bucket sort of the points ---------------------------------------------------------------------------
// The extraction only needs "points of one voxel contiguous, in time order".  A sweep is time ordered and a scan line
// stays inside one 0.8 m voxel for many consecutive points, so the unit that is sorted is the RUN (maximal stretch of
// consecutive points with one voxel key, cut at tile boundaries), not the point:
//   k_pt_runs    the only pass over the AoS input: voxel keys of a tile, run heads, one (key rest | start index)
//                composite per run dropped into the bin of its bucket (4096 buckets of fixed capacity, the slot claimed
//                with one global atomic per non-empty (tile, bucket)), the run length stored next to the start index
//   k_pt_bucket  one wavefront per bucket: exclusive prefix of the bucket point counts (summed in place, no scan launch),
//                rank-by-counting of the bucket's runs, prefix of the run lengths, expansion into the per-point
//                (key, index) arrays k_roots consumes, and the head slot table of the live root voxels
// The bucket digit is built from the LOW bits of the voxel index (x&15, y&15, z&15): neighbouring voxels land in
// different buckets, so planar scenes do not overload one bucket; the order of the voxels among each other is
// irrelevant (only grouping matters).  2 launches instead of rocPRIM's ~20 (merge path) for 1 M pairs.  A bucket with
// more runs than the bin capacity raises kFlagBucketOverflow and the caller falls back to the rocPRIM radix sort.

// run heads of a tile held in s_key[0..cnt): bit i of the bitmap is set when point i starts a run; returns the run
// length of head i (distance to the next head or to the end of the tile)
__device__ __forceinline__ uint32_t run_length(const unsigned long long *s_bits, uint32_t i, uint32_t cnt) {
  uint32_t w = (i + 1) >> 6;
  unsigned long long m = (w < kTile / 64) ? (s_bits[w] & (~0ull << ((i + 1) & 63))) : 0ull;
  while (m == 0ull && ++w < kTile / 64) m = s_bits[w];
  const uint32_t nxt = m ? (w * 64 + (uint32_t)__ffsll((long long)m) - 1) : cnt;
  return min(nxt, cnt) - i;
}

// every per-call fill (status words, slot keys, bucket counters, head table) in ONE launch: each hipMemsetAsync is its
// own ~2-7 us kernel, and an extraction call needs five of them
struct InitArgs {
  uint32_t *p[8];
  uint32_t nw[8];  // 32-bit words
  uint32_t val[8];
};
__global__ void __launch_bounds__(256) k_init(InitArgs I) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
#pragma unroll
  for (int r = 0; r < 8; ++r)
    if (i < I.nw[r]) I.p[r][i] = I.val[r];
}

// counts = run counts [kBuckets] | point counts [kBuckets].  1024 threads per 4096-point tile: one tile per CU is all a
// 1 M-point sweep offers, so the latency hiding has to come from wavefronts of the same workgroup.
constexpr int kRunThreads = 512;
__global__ void __launch_bounds__(kRunThreads) k_pt_runs(wc_points pts, double vs, uint64_t n, uint32_t *counts, uint64_t *bins,
                                                        uint32_t bin_cap, uint32_t *status) {
  __shared__ uint32_t s_key[kTile];
  __shared__ unsigned long long s_bits[kTile / 64];
  __shared__ uint32_t s_runs[kBuckets];
  __shared__ uint32_t s_pts[kBuckets];
  const int t = threadIdx.x;
  const uint64_t t0 = (uint64_t)blockIdx.x * kTile;
  const uint32_t cnt = (uint32_t)min((uint64_t)kTile, n - t0);
  double x0, y0, z0;
  load_xyz(pts, 0, x0, y0, z0);
  uint32_t key[kTile / kRunThreads];
#pragma unroll
  for (int j = 0; j < kTile / kRunThreads; ++j) {  // all loads of the tile in flight before anything waits on them
    const uint32_t i = (uint32_t)j * kRunThreads + t;
    key[j] = 0;
    if (i < cnt) {
      double x, y, z;
      load_xyz(pts, t0 + i, x, y, z);
      int rx = vox(x, vs) - vox(x0, vs) + 512, ry = vox(y, vs) - vox(y0, vs) + 512, rz = vox(z, vs) - vox(z0, vs) + 512;
      if ((unsigned)rx >= 1024u || (unsigned)ry >= 1024u || (unsigned)rz >= 1024u) {
        raise_flag(status, kFlagKeyRange);
        rx = min(max(rx, 0), 1023), ry = min(max(ry, 0), 1023), rz = min(max(rz, 0), 1023);
      }
      key[j] = (uint32_t)rx | ((uint32_t)ry << 10) | ((uint32_t)rz << 20);
    }
  }
  for (int b = t; b < kBuckets; b += kRunThreads) {
    s_runs[b] = 0;
    s_pts[b] = 0;
  }
  if (t < kTile / 64) s_bits[t] = 0ull;
#pragma unroll
  for (int j = 0; j < kTile / kRunThreads; ++j) s_key[j * kRunThreads + t] = key[j];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kTile / kRunThreads; ++j) {
    const uint32_t i = (uint32_t)j * kRunThreads + t;
    const bool headp = i < cnt && (i == 0 || key[j] != s_key[i - 1]);
    const unsigned long long m = __ballot(headp);  // i is 64-aligned per wavefront: one bitmap word each
    if ((t & 63) == 0) s_bits[i >> 6] = m;
  }
  __syncthreads();
  uint32_t rank[kTile / kRunThreads], len[kTile / kRunThreads];
#pragma unroll
  for (int j = 0; j < kTile / kRunThreads; ++j) {
    const uint32_t i = (uint32_t)j * kRunThreads + t;
    rank[j] = 0xFFFFFFFFu;
    if (i < cnt && ((s_bits[i >> 6] >> (i & 63)) & 1ull)) {
      const uint32_t d = key_digit(key[j]);
      len[j] = run_length(s_bits, i, cnt);
      rank[j] = atomicAdd(&s_runs[d], 1u);
      atomicAdd(&s_pts[d], len[j]);
    }
  }
  __syncthreads();
  for (int b = t; b < kBuckets; b += kRunThreads) {
    const uint32_t c = s_runs[b];
    if (c) {
      s_runs[b] = atomicAdd(&counts[b], c);  // this tile's range inside the bin of bucket b
      atomicAdd(&counts[kBuckets + b], s_pts[b]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kTile / kRunThreads; ++j) {
    if (rank[j] == 0xFFFFFFFFu) continue;
    const uint32_t i = (uint32_t)j * kRunThreads + t;
    const uint32_t d = key_digit(key[j]);
    const uint32_t pos = s_runs[d] + rank[j];
    if (pos < bin_cap) bins[(size_t)d * bin_cap + pos] = run_comp(key_rest(key[j]), (uint32_t)(t0 + i), len[j]);
  }
}

// one wavefront per bucket (four per workgroup).  Dynamic LDS per wavefront: lds_cap x {composite, composite, point
// offset}.  Nothing is expanded to per-point arrays: the roots pass walks the runs itself.  Buckets with up to 64 runs
// (time-ordered sweeps: a handful) are ranked by counting; larger ones (sweeps without run structure: one run per point)
// by a bitonic sort inside the wavefront, in LDS, no barriers.
// The live roots (voxel segments with more than min_points points, InitOctoTree cc:129) of the workgroup go into the head
// table COMPACTED from slot first = (points in front of the workgroup) / (min_points + 1): live root i of the workgroup
// starts at least (min_points + 1) i points behind the workgroup's first point, so first + i never reaches the range of the
// next workgroup - a dense work list without a global counter (root_first / root_cnt per workgroup).
__global__ void __launch_bounds__(256) k_pt_bucket(uint64_t *bins, uint32_t bin_cap, uint32_t lds_cap, const uint32_t *__restrict__ counts,
                                                  uint32_t *run_off, HeadRec *head_slots, uint32_t *root_cnt, uint32_t *root_first,
                                                  int min_points, uint32_t *status) {
  extern __shared__ __align__(16) unsigned char s_dyn[];
  __shared__ uint32_t s_red[4], s_live[4];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  uint64_t *s_a = (uint64_t *)s_dyn + (size_t)w * lds_cap;
  uint64_t *s_b = (uint64_t *)s_dyn + (size_t)(4 + w) * lds_cap;
  uint32_t *s_off = (uint32_t *)((uint64_t *)s_dyn + (size_t)8 * lds_cap) + (size_t)w * lds_cap;
  const uint32_t b0 = blockIdx.x * 4u, b = b0 + w;
  const uint32_t *pcounts = counts + kBuckets;
  // everything that comes from HBM / L2 is requested up front: the bucket's own counts, its first 64 runs, and this
  // thread's share of the point counts in front of the workgroup's buckets
  uint32_t nb = counts[b];
  const uint32_t total = pcounts[b];
  uint64_t *bin = bins + (size_t)b * bin_cap;
  const uint64_t first = ((uint32_t)lane < min(nb, bin_cap)) ? bin[lane] : 0ull;
  uint32_t part = 0, runs_part = 0;
  for (uint32_t i = t; i < b0; i += 256) part += pcounts[i];
  if (blockIdx.x == gridDim.x - 1)
    for (uint32_t i = t; i < kBuckets; i += 256) runs_part += counts[i];  // run total of the sweep (statistics for the host)
  uint32_t before = 0;  // points of the workgroup's buckets in front of this wavefront's
  for (int j = 0; j < w; ++j) before += pcounts[b0 + j];
  for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
  if (lane == 0) s_red[w] = part;
  if (blockIdx.x == gridDim.x - 1) {
    for (int off = 32; off >= 1; off >>= 1) runs_part += __shfl_xor(runs_part, off);
    if (lane == 0) atomicAdd(&status[5], runs_part);
  }
  if (nb > bin_cap || nb > lds_cap) {
    if (lane == 0) raise_flag(status, nb > bin_cap ? kFlagBucketOverflow : kFlagLdsOverflow);
    nb = 0;  // the caller reruns with a larger LDS capacity or on the general path
  }
  if (nb <= 64) {
    if ((uint32_t)lane < nb) s_a[lane] = first;
    __builtin_amdgcn_wave_barrier();
    if ((uint32_t)lane < nb) {  // rank by counting: composites are unique (start index)
      uint32_t rank = 0;
      for (uint32_t j = 0; j < nb; ++j) rank += (s_a[j] < first) ? 1u : 0u;
      s_b[rank] = first;
    }
  } else {
    uint32_t N = 128;
    while (N < nb) N <<= 1;
    s_b[lane] = first;
    for (uint32_t i = lane + 64; i < N; i += 64) s_b[i] = (i < nb) ? bin[i] : ~0ull;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t k = 2; k <= N; k <<= 1)
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t p = lane; p < N / 2; p += 64) {
          const uint32_t i = 2 * p - (p & (j - 1)), l = i + j;
          const uint64_t x = s_b[i], y = s_b[l];
          if ((x > y) == ((i & k) == 0)) {
            s_b[i] = y;
            s_b[l] = x;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
  }
  __builtin_amdgcn_wave_barrier();
  // exclusive prefix of the run lengths in sorted order -> offsets inside the bucket; sorted runs back in place
  uint32_t carry = 0;
  for (uint32_t i0 = 0; i0 < nb; i0 += 64) {
    const uint32_t i = i0 + lane;
    const uint64_t c = (i < nb) ? s_b[i] : 0ull;
    const uint32_t l = (i < nb) ? comp_len(c) : 0u;
    uint32_t inc = l;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = __shfl_up(inc, off);
      if (lane >= off) inc += v;
    }
    if (i < nb) {
      s_off[i] = carry + inc - l;
      bin[i] = c;
      run_off[(size_t)b * bin_cap + i] = carry + inc - l;
    }
    carry += __shfl(inc, 63);
  }
  __builtin_amdgcn_wave_barrier();
  // live roots: voxel segments (runs with one key rest) with more than min_points points.  Chunks of 64 runs from the END,
  // so that the start of the next segment is known when a segment head is looked at; the live ones are listed in LDS
  // (the unsorted-composite array is free now) as head | end << 16.
  uint32_t *s_list = (uint32_t *)s_a;
  uint32_t nlive = 0, next_head = nb;
  for (int i0 = (int)((nb + 63) / 64) * 64 - 64; i0 >= 0; i0 -= 64) {
    const uint32_t r = (uint32_t)i0 + lane;
    const bool in = r < nb;
    const uint32_t rest = in ? comp_rest(s_b[r]) : 0u;
    const bool head = in && (r == 0 || comp_rest(s_b[r - 1]) != rest);
    const unsigned long long hm = __ballot(head);
    const unsigned long long above = (lane == 63) ? 0ull : (hm >> (lane + 1));
    const uint32_t r2 = above ? r + (uint32_t)__ffsll((long long)above) : next_head;  // first run of the next segment
    const uint32_t seg = (r2 < nb ? s_off[min(r2, nb - 1)] : total) - (in ? s_off[r] : 0u);
    const bool live = head && seg > (uint32_t)min_points;
    const unsigned long long lm = __ballot(live);
    if (live) s_list[nlive + (uint32_t)__popcll(lm & ((1ull << lane) - 1ull))] = r | (r2 << 16);
    nlive += (uint32_t)__popcll(lm);
    if (hm) next_head = (uint32_t)i0 + (uint32_t)__ffsll((long long)hm) - 1u;
  }
  if (lane == 0) s_live[w] = nlive;
  __syncthreads();
  const uint32_t pbase0 = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  const uint32_t pbase = pbase0 + before;
  const uint32_t first_slot = pbase0 / (uint32_t)(min_points + 1);
  uint32_t lbase = 0;
  for (int j = 0; j < w; ++j) lbase += s_live[j];
  if (t == 0) {
    root_cnt[blockIdx.x] = s_live[0] + s_live[1] + s_live[2] + s_live[3];
    root_first[blockIdx.x] = first_slot;
  }
  for (uint32_t k = lane; k < nlive; k += 64) {
    const uint32_t r = s_list[k] & 0xFFFFu, r2 = s_list[k] >> 16;
    const uint32_t seg = (r2 < nb ? s_off[r2] : total) - s_off[r];
    head_slots[first_slot + lbase + k] = HeadRec{pbase + s_off[r], b * bin_cap + r, r2 - r, seg};
  }
}

constexpr uint32_t kPtBinMax = 1024;  // runs per bucket the in-LDS path takes (4 wavefronts x 20 B x 1024 = 80 KB of LDS)

// bin capacity for n points: 32 x the mean a cloud without any run structure would produce (voxel occupancy is heavy
// tailed: a room sweep of 64 k points has buckets of 280 runs at a mean of 16), 64 at least
inline uint32_t pt_bin_cap(uint64_t n) {
  uint32_t cap = 64;
  while (cap < kPtBinMax && (uint64_t)cap * kBuckets < 32 * n) cap *= 2;
  return cap;
}

// The extraction's control block (its own buffer: the other entry points of the library never touch it, so it can be
// cleared ahead of time): status words | run counts, point counts | root_cnt, root_first | time-bin counts.
constexpr uint32_t kCtrlStatus = 0, kCtrlCounts = 64, kCtrlRoots = kCtrlCounts + 2 * kBuckets, kCtrlBins = kCtrlRoots + 2 * (kBuckets / 4),
                   kCtrlFx = kCtrlBins + kBuckets, kCtrlWords = kCtrlFx + 4 * 16 * 32;  // + the fast path's counter banks (extract_fast.inc)

// clears the control block and stores the mailbox address (status words [8, 9]); used in front of a call, or - the usual
// case - by finish() for the NEXT call: run ahead of time it hides behind the host's turn-around between two sweeps
// instead of being the first, 3 us long, kernel of the sweep with a 4 us submission gap behind it.
int clear_ctrl(wc_ctx *ctx) {
  WC_TRY(wc_ensure(ctx, ctx->b_ex_ctrl, kCtrlWords * 4));
  uint32_t *ctrl = (uint32_t *)ctx->b_ex_ctrl.p;
  InitArgs I{};
  void *dp = nullptr;
  if (hipHostGetDevicePointer(&dp, ctx->h_status, 0) != hipSuccess) dp = nullptr;
  const unsigned long long a = (unsigned long long)dp;
  I.p[0] = ctrl, I.nw[0] = 8, I.val[0] = 0u;
  I.p[1] = ctrl + 10, I.nw[1] = kCtrlRoots - 10, I.val[1] = 0u;  // rest of the status words, run / point counts
  I.p[2] = ctrl + 8, I.nw[2] = 1, I.val[2] = (uint32_t)a;
  I.p[3] = ctrl + 9, I.nw[3] = 1, I.val[3] = (uint32_t)(a >> 32);
  I.p[4] = ctrl + kCtrlBins, I.nw[4] = kBuckets + 4 * 16 * 32, I.val[4] = 0u;  // time-bin counts + fast-path counters
  k_init<<<(kCtrlRoots + 255) / 256, 256, 0, ctx->stream>>>(I);
  WC_HIP(ctx, hipGetLastError());
  return WC_OK;
}

// counts (run counts | point counts) and status must be cleared by the caller.  Leaves the sorted runs in b_misc[1], their
// point offsets in b_misc[3] and the live roots in the head slot table.  The LDS capacity per bucket (ctx->ex.lds_cap:
// 256, 512 or 1024 runs) follows the data: a bucket above it raises kFlagLdsOverflow and the call is repeated one size up.
int point_sort_runs(wc_ctx *ctx, const wc_points &pts, double vs, HeadRec *head_slots, int min_points, uint32_t *status) {
  hipStream_t st = ctx->stream;
  const uint64_t n = pts.n;
  const uint32_t cap = pt_bin_cap(n);
  const uint32_t lds_cap = std::min(cap, std::max(64u, ctx->ex.lds_cap));
  WC_TRY(wc_ensure(ctx, ctx->b_misc[1], (uint64_t)kBuckets * cap * 8));  // run bins
  WC_TRY(wc_ensure(ctx, ctx->b_misc[3], (uint64_t)kBuckets * cap * 4));  // point offsets of the sorted runs
  uint32_t *counts = (uint32_t *)ctx->b_ex_ctrl.p + kCtrlCounts;  // run counts | point counts | root_cnt | root_first
  const unsigned tiles = (unsigned)((n + kTile - 1) / kTile);
  const size_t lds = (size_t)4 * lds_cap * 20;
  if (!ctx->ex.bucket_attr_set) {  // per ctx (= per device), not per process
    WC_HIP(ctx, hipFuncSetAttribute((const void *)k_pt_bucket, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * kPtBinMax * 20)));
    ctx->ex.bucket_attr_set = true;
  }
  k_pt_runs<<<tiles, kRunThreads, 0, st>>>(pts, vs, n, counts, (uint64_t *)ctx->b_misc[1].p, cap, status);
  k_pt_bucket<<<kBuckets / 4, 256, lds, st>>>((uint64_t *)ctx->b_misc[1].p, cap, lds_cap, counts, (uint32_t *)ctx->b_misc[3].p, head_slots,
                                             counts + 2 * kBuckets, counts + 2 * kBuckets + kBuckets / 4, min_points, status);
  WC_HIP(ctx, hipGetLastError());
  return WC_OK;
}

template <typename K>
int sort_pairs(wc_ctx *ctx, K *kin, K *kout, uint32_t *vin, uint32_t *vout, size_t n, unsigned end_bit) {
  // rocPRIM's default switches to a ~20-launch merge sort below 2^20 items (MergeSortLimit); the Onesweep radix path is
  // several times faster for (u32 key, u32 index) pairs at these sizes, so the limit is set to zero.
  using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;
  size_t tmp = 0;
  WC_HIP(ctx, rocprim::radix_sort_pairs<cfg>(nullptr, tmp, kin, kout, vin, vout, n, 0u, end_bit, ctx->stream));
  WC_TRY(wc_ensure(ctx, ctx->b_sorttmp, tmp));
  tmp = ctx->b_sorttmp.cap;
  WC_HIP(ctx, rocprim::radix_sort_pairs<cfg>(ctx->b_sorttmp.p, tmp, kin, kout, vin, vout, n, 0u, end_bit, ctx->stream));
  return WC_OK;
}

#include "extract_fast.inc"
#include "extract_split.inc"

// ---- host side of the fast path ------------------------------------------------------------------------------------------
// grow-only buffer that is ZERO when it is (re)allocated: the fast path's tables are zero at rest
int fx_ensure_zero(wc_ctx *ctx, wc_buf &b, size_t bytes) {
  if (bytes <= b.cap) return WC_OK;
  WC_TRY(wc_ensure(ctx, b, bytes));
  WC_HIP(ctx, hipMemsetAsync(b.p, 0, b.cap, ctx->stream));
  return WC_OK;
}

bool fx_applicable(const wc_ctx *ctx, uint64_t n, double t_lo, double t_hi) {
  const wc_params &P = ctx->P;
  if (P.exact_sums || ctx->dev.exact_sums) return false;
  if (n < 64 || !(P.voxel_size > 0.0f) || P.voxel_size >= 0.99f) return false;  // |p - centre| 2^32 must fit an int32
  if (!(P.cluster_gap > 1e-6) || !(t_hi > t_lo)) return false;
  if ((t_hi - t_lo) / (P.cluster_gap * 0.999) >= 1048000.0) return false;
  return true;
}

int fx_tail(wc_ctx *ctx, bool layer2);

int run_pipeline_fast(wc_ctx *ctx, const wc_points &pts, double t_lo, double t_hi, wc_surfel *d_out, wc_surfel_id *d_ids, uint64_t cap) {
  const wc_params &P = ctx->P;
  const uint64_t n = pts.n;
  hipStream_t st = ctx->stream;
  FxArgs A;
  std::memset(&A, 0, sizeof(A));
  ExParams &E = A.P;
  E.vs = (double)P.voxel_size, E.vs_f = P.voxel_size, E.max_layer = P.max_layer, E.min_points = P.min_points;
  E.thr = (double)P.planer_threshold, E.min_like = P.min_plane_likeness;
  for (int i = 0; i < 3; ++i) E.view[i] = P.view_point[i];
  E.gap = P.cluster_gap, E.cluster_min = P.cluster_min_points;
  E.t_lo_bits = ordered_bits_host(t_lo);
  E.t_span_bits = ordered_bits_host(t_hi) - E.t_lo_bits;
  E.dbg = ctx->dev.debug_skip;  // (development options: wc_ctx_set_dev_option)
  E.merge_min = ctx->dev.fx_merge_min;
  unsigned tbits = 1;
  while (tbits < 64 && (E.t_span_bits >> tbits)) ++tbits;
  A.pts = pts;
  A.t_lo = t_lo;
  int e = 0;
  (void)std::frexp(t_hi - t_lo, &e);  // span < 2^e
  A.tick = std::ldexp(1.0, 40 - e), A.inv_tick = std::ldexp(1.0, e - 40);
  A.inv_w = 1.0 / (P.cluster_gap * 0.999);
  A.inv_span = 1.0 / (t_hi - t_lo);
  A.inv_vs = 1.0 / (double)P.voxel_size;
  A.qs = 4294967296.0, A.inv_q = 1.0 / 4294967296.0, A.inv_qq = std::ldexp(1.0, -44);
  // capacities: a root block (640 B) per 8 points, a layer-2 node block per 16; the hash holds 4 x the root blocks; one
  // 128-byte record per LDS hash slot of every tile + a spill pool
  uint32_t mr = 1024;
  while ((uint64_t)mr * 8 < n) mr *= 2;
  A.mr_per = mr / kFxSub, A.mq_per = std::max(64u, mr / 2 / kFxSub);
  uint32_t tr = 2048;  // hash slots = root blocks: one per four points
  while ((uint64_t)tr * 4 < n) tr *= 2;
  A.tr_mask = tr - 1;
  A.static_map = n <= 2000000ull ? 1u : 0u;  // (up to ~2 M points the node wavefronts are one round)
  {  // the two layouts k_fx_acc has wide loads for (anything else: element-wise loads)
    const uintptr_t px = (uintptr_t)pts.xyz, pt = (uintptr_t)pts.time;
    if (pts.xyz_stride == 48u && pts.time_stride == 48u && pt == px + 24u && px % 16u == 0u)
      A.fmt = 1u;
    else if (pts.xyz_stride == 12u && pts.time_stride == 8u && px % 16u == 0u && pt % 16u == 0u)
      A.fmt = 2u;
  }
  const unsigned tiles = (unsigned)((n + kFxTile - 1) / kFxTile);
  A.rec_tiles = tiles;
  // spill pool (partials the tile's 256-cell LDS hash could not take): eight banks by tile index.  A tile spills at most one
  // record per point, so ceil(tiles / 8) x 1024 records per bank can never overflow - used for sweeps up to 2 M points (a sparse
  // sweep in firing order, e.g. 75 k points of a 40 m room, puts three quarters of its points there) and, sticky, on any context
  // whose pool has overflowed once; larger clouds start with n / 32 per bank
  const uint64_t spill_full = ((uint64_t)tiles + 7) / 8 * kFxTile;
  // (round 5: full size up to 8 M points - a 4 M-point room in firing order overflowed the n / 32 pool on every fresh context, whose first
  // two calls then ran on the exact path; 128 bytes per point of never-cleared memory are nothing on this card)
  A.spill_per = (uint32_t)((n <= 8000000ull || ctx->ex.fx_spill_full) ? spill_full : std::max<uint64_t>(1024, n / 32));
  A.rec_cap = tiles * (uint32_t)kFxRecTile + 8u * A.spill_per;
  const uint64_t total_slots = (n * (uint64_t)(P.max_layer + 1)) / (uint64_t)P.cluster_min_points + 1;
  uint32_t bin_cap = 64;
  while (bin_cap < kSlotBinMax && (uint64_t)bin_cap * kBuckets < 2 * total_slots) bin_cap *= 2;
  WC_TRY(fx_ensure_zero(ctx, ctx->b_fx[0], (size_t)tr * 4));
  WC_TRY(wc_ensure(ctx, ctx->b_fx[1], (size_t)mr * 4));  // dense root list: written before it is read
  WC_TRY(fx_ensure_zero(ctx, ctx->b_fx[2], (size_t)tr * kFxBlockW * 8));
  WC_TRY(fx_ensure_zero(ctx, ctx->b_fx[3], (size_t)A.mq_per * kFxSub * kFxBlockW * 8));
  WC_TRY(wc_ensure(ctx, ctx->b_fx[4], (size_t)2 * A.rec_cap * kFxRecW * 8));  // records: reachable through list heads only, never cleared
  const unsigned ngrid = std::min<unsigned>(256 * 16, std::max<unsigned>(64, (unsigned)(n / 256)));  // k_fx_nodes<1>; <2> uses fewer
  WC_TRY(wc_ensure(ctx, ctx->b_fx[6], (size_t)ngrid * kFxJobCap * kFxJobW * 8));  // cluster jobs: written before they are read
  // node stage as two kernels (extract_split.inc) where one round of wavefronts does not hold the sweep's parents: above 2 M points
  // (the development option fx_split = 0 / 1 pins the choice: tests run both forms on the same clouds)
  ctx->ex.fx_split = ctx->ex.batch_defer || (ctx->dev.fx_split >= 0 ? ctx->dev.fx_split != 0 : !A.static_map);  // (a batch: K sweeps' parents in one launch)
  if (ctx->ex.fx_split) {
    A.static_map = 0u;
    A.jobpool_per = (uint32_t)std::max<uint64_t>(4096, n / (uint64_t)std::max(1, P.cluster_min_points) / 2);  // 8 sub-pools: 4 x the worst case
    WC_TRY(wc_ensure(ctx, ctx->b_fx[7], ((size_t)mr / 8 + 2) * kFxNodeW * 64 * 8));
    WC_TRY(wc_ensure(ctx, ctx->b_fx[8], ((size_t)mr / 8 + 2) * kFxDescW * 4));
    WC_TRY(wc_ensure(ctx, ctx->b_fx[9], (size_t)8 * A.jobpool_per * kFxJobW * 8));
    A.nodes = (unsigned long long *)ctx->b_fx[7].p, A.wdesc = (uint32_t *)ctx->b_fx[8].p, A.jobpool = (unsigned long long *)ctx->b_fx[9].p;
  }
  WC_TRY(wc_ensure(ctx, ctx->b_slots, total_slots * sizeof(wc_surfel)));
  WC_TRY(wc_ensure(ctx, ctx->b_slot_ids, total_slots * sizeof(wc_surfel_id)));
  WC_TRY(wc_ensure(ctx, ctx->b_slot_keys[1], (uint64_t)kBuckets * bin_cap * 8));
  if (ctx->ex.fx_dirty) {  // a previous sweep ended abnormally: everything back to zero
    for (int i : {0, 2, 3}) WC_HIP(ctx, hipMemsetAsync(ctx->b_fx[i].p, 0, ctx->b_fx[i].cap, st));  // (not the root list, not the records)
    ctx->ex.fx_dirty = false;
  }
  // two control blocks, used alternately: k_slot_emit of a sweep clears the block of the next one
  WC_TRY(wc_ensure(ctx, ctx->b_fx[5], (size_t)2 * kCtrlWords * 4));
  if (!ctx->ex.fx_ctrl_ready) {
    WC_HIP(ctx, hipMemsetAsync(ctx->b_fx[5].p, 0, (size_t)2 * kCtrlWords * 4, st));
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, ctx->h_status, 0) != hipSuccess) dp = nullptr;
    const unsigned long long a = (unsigned long long)dp;
    const uint32_t addr[2] = {(uint32_t)a, (uint32_t)(a >> 32)};
    for (int b = 0; b < 2; ++b)
      WC_HIP(ctx, hipMemcpyAsync((uint32_t *)ctx->b_fx[5].p + (size_t)b * kCtrlWords + 8, addr, 8, hipMemcpyHostToDevice, st));
    WC_HIP(ctx, hipStreamSynchronize(st));  // (addr is a stack variable)
    ctx->ex.fx_ctrl_ready = true;
    ctx->ex.fx_parity = 0;
  }
  uint32_t *ctrl = (uint32_t *)ctx->b_fx[5].p + (size_t)ctx->ex.fx_parity * kCtrlWords;
  for (int q = 0; q < 64; ++q) ctx->h_status[q] = 0;
  A.rkey = (uint32_t *)ctx->b_fx[0].p, A.rlist = (uint32_t *)ctx->b_fx[1].p;
  A.blk = (unsigned long long *)ctx->b_fx[2].p, A.blk2 = (unsigned long long *)ctx->b_fx[3].p;
  A.rec = (unsigned long long *)ctx->b_fx[4].p;
  A.jobs = (unsigned long long *)ctx->b_fx[6].p;
  A.status = ctrl + kCtrlStatus;
  A.cnt = ctrl + kCtrlFx;
  A.slots = (wc_surfel *)ctx->b_slots.p, A.slot_ids = (wc_surfel_id *)ctx->b_slot_ids.p;
  A.slots_per = (uint32_t)std::min<uint64_t>(total_slots / kFxSub, 0x7FFFFFFFu);
  A.slot_counts = ctrl + kCtrlBins;
  A.slot_bins = (uint64_t *)ctx->b_slot_keys[1].p;
  A.slot_bin_cap = bin_cap;
  auto mark = [&](int i) {
    if (ctx->ex_prof && (ctx->ex_prof_mode != 2 || i == 1 || i == 5)) (void)hipEventRecord(ctx->ex_ev[i], st);
  };
  mark(0);
  mark(1);
  static const bool dbg = wc_log_env("WC_FX_DEBUG");
  auto dbg_sync = [&](const char *what) {
    if (!dbg) return;
    fprintf(stderr, "[fx] %s ...", what);
    const hipError_t e = hipStreamSynchronize(st);
    uint32_t w[64], c[4 * 16 * 32];
    (void)hipMemcpy(w, A.status, sizeof(w), hipMemcpyDeviceToHost);
    (void)hipMemcpy(c, A.cnt, sizeof(c), hipMemcpyDeviceToHost);
    uint32_t r = 0, q = 0, sl = 0, sp = 0;
    for (int j = 0; j < 16; ++j) r += c[j * 32], q += c[(16 + j) * 32], sl += c[(32 + j) * 32], sp += c[(48 + j) * 32];
    fprintf(stderr, " %s flags=%u roots=%u nodes2=%u slots=%u spill=%u\n", hipGetErrorString(e), w[1], r, q, sl, sp);
  };
  if (dbg) fprintf(stderr, "[fx] n=%llu tiles=%u mr_per=%u mq_per=%u tr=%u slots_per=%u bin_cap=%u\n", (unsigned long long)n, tiles, A.mr_per, A.mq_per, tr, A.slots_per, bin_cap);
  if (ctx->ex.batch_defer) {  // wc_extract_surfels_batch_enqueue launches this sweep's kernels together with the other sweeps'
    static_assert(sizeof(FxArgs) <= sizeof(ctx->ex.roots_args), "ctx.h: roots_args too small");
    std::memcpy(ctx->ex.roots_args, &A, sizeof(A));
    ctx->ex.total_slots = total_slots;
    ctx->ex.bin_cap = bin_cap;
    ctx->ex.fast_slots = true;
    ctx->ex.tail = &fx_tail;
    ctx->ex.d_out = d_out, ctx->ex.d_ids = d_ids, ctx->ex.cap = cap;
    ctx->ex.layer2_done = P.max_layer >= 2 && ctx->ex.last_splits > 0;
    // (the node kernels of a batch loop over their sweep's parents, eight per wavefront: a grid of n / 256 blocks per sweep - what
    // a single sweep's static hand-out wants - is 39 k workgroups for ten sweeps, 34 k of which find nothing to do)
    ctx->ex.fx_tiles = tiles, ctx->ex.fx_ngrid = std::max(64u, ngrid / 8u);
    ctx->ex.deferred = true;
    return WC_OK;
  }
  k_fx_acc<1><<<tiles, kFxThreads, 0, st>>>(A);
  dbg_sync("k_fx_acc<1>");
  mark(2);
  if (ctx->ex.fx_long_lists) k_fx_merge<1><<<std::min<unsigned>(ngrid * 2u, 8192u), 128, 0, st>>>(A);
  if (ctx->ex.fx_split) {
    k_fx_walk<1><<<ngrid, 64, 0, st>>>(A);
    dbg_sync("k_fx_walk<1>");
    k_fx_test<1><<<ngrid, 64, 0, st>>>(A);
  } else {
    k_fx_nodes<1><<<ngrid, 64, 0, st>>>(A);
  }
  dbg_sync("k_fx_nodes<1>");
  mark(3);
  static_assert(sizeof(FxArgs) <= sizeof(ctx->ex.roots_args), "ctx.h: roots_args too small");
  std::memcpy(ctx->ex.roots_args, &A, sizeof(A));
  ctx->ex.total_slots = total_slots;
  ctx->ex.bin_cap = bin_cap;
  ctx->ex.fast_slots = true;
  ctx->ex.tail = &fx_tail;
  ctx->ex.d_out = d_out, ctx->ex.d_ids = d_ids, ctx->ex.cap = cap;
  return fx_tail(ctx, P.max_layer >= 2 && ctx->ex.last_splits > 0);
}

constexpr int kMailTicket = 120;  // word of the pinned mailbox (ctx->h_status) that carries the extraction's completion ticket
__global__ void k_ex_ticket(uint32_t *host_word, uint32_t ticket) {
  __hip_atomic_store(host_word, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// layer-2 pair (optional), time order + gather
int fx_tail(wc_ctx *ctx, bool layer2) {
  hipStream_t st = ctx->stream;
  FxArgs A;
  std::memcpy(&A, ctx->ex.roots_args, sizeof(A));
  auto mark = [&](int i) {
    if (ctx->ex_prof && (ctx->ex_prof_mode != 2 || i == 1 || i == 5)) (void)hipEventRecord(ctx->ex_ev[i], st);
  };
  if (layer2) {
    const unsigned tiles = (unsigned)((A.pts.n + kFxTile - 1) / kFxTile);
    k_fx_acc<2><<<tiles, kFxThreads, 0, st>>>(A);
    const unsigned ngrid = std::min<unsigned>(256 * 16, std::max<unsigned>(64, (unsigned)(A.pts.n / 256)));  // (the job buffer's blocks)
    if (ctx->ex.fx_long_lists2) k_fx_merge<2><<<std::min(8192u, std::max(64u, ctx->ex.last_splits)), 128, 0, st>>>(A);
    const unsigned g2 = std::min(std::min(256u * 8u, ngrid), std::max(64u, ctx->ex.last_splits));
    if (ctx->ex.fx_split) {
      k_fx_walk<2><<<g2, 64, 0, st>>>(A);
      k_fx_test<2><<<g2, 64, 0, st>>>(A);
    } else {
      k_fx_nodes<2><<<g2, 64, 0, st>>>(A);
    }
  }
  ctx->ex.layer2_done = layer2;
  mark(4);
  uint32_t *next_ctrl = (uint32_t *)ctx->b_fx[5].p + (size_t)(ctx->ex.fx_parity ^ 1) * kCtrlWords;
  static_assert(kCtrlWords <= (kBuckets / 4) * 256, "k_slot_emit's grid covers the control block");
  k_slot_emit<<<kBuckets / 4, 256, 0, st>>>(A.slot_counts, A.slot_bins, ctx->ex.bin_cap, (const wc_surfel *)ctx->b_slots.p,
                                           (const wc_surfel_id *)ctx->b_slot_ids.p, A.status, ctx->ex.d_out, ctx->ex.d_ids, ctx->ex.cap, next_ctrl,
                                           kCtrlWords);
  mark(5);
  // the completion ticket: ONE thread behind the sweep's last kernel stores it to the pinned mailbox and wc_extract_surfels_finish reads
  // host memory instead of waiting for the stream (the LM loop's wait_mail).  Behind a kernel boundary nothing has to be fenced inside
  // k_slot_emit (what round 3's last-workgroup ticket paid for): every store of the sweep has left its L2 when this kernel starts.
  ctx->ex.ticket = 0;
  if (ctx->h_status_dev && !ctx->ex_prof && !ctx->dev.ex_sync) {
    if (++ctx->ex.ticket_seq == 0u) ++ctx->ex.ticket_seq;
    ctx->ex.ticket = ctx->ex.ticket_seq;
    k_ex_ticket<<<1, 1, 0, st>>>(ctx->h_status_dev + kMailTicket, ctx->ex.ticket);
  }
  static const bool fx_dbg = wc_log_env("WC_FX_DEBUG");
  if (fx_dbg) fprintf(stderr, "[fx] k_slot_emit (layer2=%d) ... %s\n", (int)layer2, hipGetErrorString(hipStreamSynchronize(st)));
  WC_HIP(ctx, hipGetLastError());
  return WC_OK;
}

// layer-2 pass of the queued roots (optional), time order + gather of the surfels, status read-back
template <typename K, bool RUNS>
int pipeline_tail(wc_ctx *ctx, bool layer2) {
  hipStream_t st = ctx->stream;
  RootsArgs A;
  memcpy(&A, ctx->ex.roots_args, sizeof(A));
  uint32_t *status = A.status;
  wc_surfel *d_out = ctx->ex.d_out;
  wc_surfel_id *d_ids = ctx->ex.d_ids;
  const uint64_t cap = ctx->ex.cap, total_slots = ctx->ex.total_slots;
  auto mark = [&](int i) {
    if (ctx->ex_prof && (ctx->ex_prof_mode != 2 || i == 1 || i == 5)) (void)hipEventRecord(ctx->ex_ev[i], st);
  };
  if (layer2) {
    const unsigned grid2 = std::min(kRoots2Grid, std::max(64u, ctx->ex.last_splits));  // sized by the previous call's queue
    if constexpr (RUNS) {
      if (ctx->ex.unordered)
        k_roots_banks<K, 2, true><<<grid2, 64, 0, st>>>(A, (const K *)ctx->b_keys[1].p);
      else
        k_roots<K, 2, true><<<grid2, 64, 0, st>>>(A, (const K *)ctx->b_keys[1].p);
    } else {
      k_roots_banks<K, 2, false><<<grid2, 64, 0, st>>>(A, (const K *)ctx->b_keys[1].p);
    }
  }
  ctx->ex.layer2_done = layer2;
  mark(4);
  if (ctx->ex.fast_slots) {
    k_slot_emit<<<kBuckets / 4, 256, 0, st>>>((const uint32_t *)ctx->b_ex_ctrl.p + kCtrlBins, A.slot_bins, ctx->ex.bin_cap, (const wc_surfel *)ctx->b_slots.p,
                                             (const wc_surfel_id *)ctx->b_slot_ids.p, status, d_out, d_ids, cap, nullptr, 0u);
  } else {
    k_iota<<<(unsigned)((total_slots + 255) / 256), 256, 0, st>>>((uint32_t *)ctx->b_slot_idx[0].p, total_slots);
    WC_TRY(sort_pairs<uint64_t>(ctx, (uint64_t *)ctx->b_slot_keys[0].p, (uint64_t *)ctx->b_slot_keys[1].p,
                                (uint32_t *)ctx->b_slot_idx[0].p, (uint32_t *)ctx->b_slot_idx[1].p, total_slots,
                                ctx->ex.slot_end_bit));
    k_fix_ties<<<(unsigned)((total_slots + 255) / 256), 256, 0, st>>>((const uint64_t *)ctx->b_slot_keys[1].p, (uint32_t *)ctx->b_slot_idx[1].p,
                                                                     total_slots, (const wc_surfel_id *)ctx->b_slot_ids.p);
    const uint64_t gth = std::min<uint64_t>(total_slots, cap) * 10;
    if (gth)
      k_gather<<<(unsigned)((gth + 255) / 256), 256, 0, st>>>((const uint32_t *)ctx->b_slot_idx[1].p, (const wc_surfel *)ctx->b_slots.p,
                                                             (const wc_surfel_id *)ctx->b_slot_ids.p, status, d_out, d_ids, cap);
  }
  mark(5);
  // the time-bin path leaves count, queue length and flags in the pinned mailbox itself (k_slot_emit, raise_flag)
  if (!ctx->ex.fast_slots) WC_HIP(ctx, hipMemcpyAsync(ctx->h_status, status, 32, hipMemcpyDeviceToHost, st));
  WC_HIP(ctx, hipGetLastError());
  return WC_OK;
}

template <typename K>
int run_pipeline(wc_ctx *ctx, const wc_points &pts, double t_lo, double t_hi, wc_surfel *d_out, wc_surfel_id *d_ids,
                 uint64_t cap, bool fast, bool fast_order) {
  const wc_params &P = ctx->P;
  const uint64_t n = pts.n;
  hipStream_t st = ctx->stream;
  ExParams E;
  E.vs = (double)P.voxel_size;
  E.vs_f = P.voxel_size;
  E.max_layer = P.max_layer;
  E.min_points = P.min_points;
  E.thr = (double)P.planer_threshold;
  E.min_like = P.min_plane_likeness;
  for (int i = 0; i < 3; ++i) E.view[i] = P.view_point[i];
  E.gap = P.cluster_gap;
  E.cluster_min = P.cluster_min_points;
  E.t_lo_bits = ordered_bits_host(t_lo);
  E.dbg = ctx->dev.debug_skip;  // (development options: wc_ctx_set_dev_option)
  E.merge_min = ctx->dev.fx_merge_min;
  const uint64_t span = ordered_bits_host(t_hi) - E.t_lo_bits;
  E.t_span_bits = span;
  unsigned tbits = 1;
  while (tbits < 64 && (span >> tbits)) ++tbits;
  const unsigned slot_end_bit = tbits >= 63 ? 64 : tbits + 1;  // one extra bit so that ~0 (invalid) sorts last

  const uint64_t total_slots = (n * (uint64_t)(P.max_layer + 1)) / (uint64_t)P.cluster_min_points + 1;
  WC_TRY(wc_ensure(ctx, ctx->b_keys[0], n * sizeof(K)));
  WC_TRY(wc_ensure(ctx, ctx->b_keys[1], n * sizeof(K)));
  WC_TRY(wc_ensure(ctx, ctx->b_vals[0], n * 4));
  WC_TRY(wc_ensure(ctx, ctx->b_vals[1], n * 4));
  WC_TRY(wc_ensure(ctx, ctx->b_cand, total_slots * kMom * 8));
  WC_TRY(wc_ensure(ctx, ctx->b_cand_meta, total_slots * 4));
  WC_TRY(wc_ensure(ctx, ctx->b_slots, total_slots * sizeof(wc_surfel)));
  WC_TRY(wc_ensure(ctx, ctx->b_slot_ids, total_slots * sizeof(wc_surfel_id)));
  WC_TRY(wc_ensure(ctx, ctx->b_slot_keys[0], total_slots * 8));
  WC_TRY(wc_ensure(ctx, ctx->b_slot_keys[1], total_slots * 8));
  WC_TRY(wc_ensure(ctx, ctx->b_slot_idx[0], total_slots * 4));
  WC_TRY(wc_ensure(ctx, ctx->b_slot_idx[1], total_slots * 4));
  WC_TRY(wc_ensure(ctx, ctx->b_misc[0], (n / (uint64_t)(P.min_points + 1) + 2) * sizeof(HeadRec)));
  // the control block: cleared ahead of time by the previous finish() (see clear_ctrl) or right here
  const bool precleared = ctx->ex.precleared && ctx->b_ex_ctrl.p;
  ctx->ex.precleared = false;
  if (!precleared) WC_TRY(clear_ctrl(ctx));
  for (int q = 0; q < 16; ++q) ctx->h_status[q] = 0;
  uint32_t *status = (uint32_t *)ctx->b_ex_ctrl.p + kCtrlStatus;

  auto mark = [&](int i) {
    if (ctx->ex_prof && (ctx->ex_prof_mode != 2 || i == 1 || i == 5)) (void)hipEventRecord(ctx->ex_ev[i], st);
  };
  mark(0);
  const bool fast_slots = fast_order && tbits <= 31 && total_slots < (1ull << 31);
  // capacity of a time bucket's bin: twice the count every bucket would get if EVERY slot held a surfel, 64 at least
  uint32_t bin_cap = 64;
  while (bin_cap < kSlotBinMax && (uint64_t)bin_cap * kBuckets < 2 * total_slots) bin_cap *= 2;
  const unsigned g256 = (unsigned)((n + 255) / 256);
  // fast path (32-bit keys): bucket sort of (voxel key, index) composites; general path: rocPRIM radix sort
  const bool fast_pts = fast && sizeof(K) == 4;
  if (fast_pts || fast_slots) {
    WC_TRY(wc_ensure(ctx, ctx->b_slot_keys[1], (uint64_t)kBuckets * bin_cap * 8));  // slot bins (the general path's sort buffer)
  }
  const uint32_t nslots = (uint32_t)(n / (uint64_t)(P.min_points + 1) + 1);
  if (!fast_slots || !fast_pts) {  // the large fills of the radix-sort paths
    InitArgs I{};
    int r = 0;
    auto fill = [&](void *p, uint64_t words, uint32_t v) { I.p[r] = (uint32_t *)p, I.nw[r] = (uint32_t)words, I.val[r] = v, ++r; };
    if (!fast_slots) fill(ctx->b_slot_keys[0].p, total_slots * 2, 0xFFFFFFFFu);  // slot keys: ~0 = no surfel in the slot
    if (!fast_pts) fill(ctx->b_misc[0].p, (uint64_t)nslots * 4, 0xFFFFFFFFu);   // sparse head slot table: pos = ~0 = no live head
    uint32_t mx = 0;
    for (int q = 0; q < r; ++q) mx = std::max(mx, I.nw[q]);
    k_init<<<(mx + 255) / 256, 256, 0, st>>>(I);
  }
  mark(1);
  if (fast_pts) {
    WC_TRY(point_sort_runs(ctx, pts, E.vs, (HeadRec *)ctx->b_misc[0].p, P.min_points, status));  // also fills the head slot table
  } else {
    k_keygen<K><<<g256, 256, 0, st>>>(pts, E.vs, (K *)ctx->b_keys[0].p, (uint32_t *)ctx->b_vals[0].p, status);
    WC_TRY(sort_pairs<K>(ctx, (K *)ctx->b_keys[0].p, (K *)ctx->b_keys[1].p, (uint32_t *)ctx->b_vals[0].p,
                         (uint32_t *)ctx->b_vals[1].p, n, 3 * KeyTraits<K>::bits));
  }
  RootsArgs A;
  A.pts = pts;
  A.P = E;
  A.n = n;
  A.vals = (const uint32_t *)ctx->b_vals[1].p;
  A.cand = (double *)ctx->b_cand.p;
  A.cand_meta = (uint32_t *)ctx->b_cand_meta.p;
  A.slots = (wc_surfel *)ctx->b_slots.p;
  A.slot_ids = (wc_surfel_id *)ctx->b_slot_ids.p;
  A.slot_keys = (uint64_t *)ctx->b_slot_keys[0].p;
  A.total_slots = total_slots;
  A.status = status;
  A.slot_counts = fast_slots ? (uint32_t *)ctx->b_ex_ctrl.p + kCtrlBins : nullptr;
  A.slot_shift = tbits > 12 ? tbits - 12 : 0u;
  A.slot_bins = (uint64_t *)ctx->b_slot_keys[1].p;
  A.runs = (const uint64_t *)ctx->b_misc[1].p;
  A.run_off = (const uint32_t *)ctx->b_misc[3].p;
  A.run_cap = pt_bin_cap(n);
  A.root_cnt = (const uint32_t *)ctx->b_ex_ctrl.p + kCtrlRoots;
  A.root_first = A.root_cnt + kBuckets / 4;
  A.slot_bin_cap = bin_cap;
  A.heads = (const HeadRec *)ctx->b_misc[0].p;
  A.nslots = nslots;
  if (!fast_pts) k_heads<K><<<g256, 256, 0, st>>>((const K *)ctx->b_keys[1].p, n, P.min_points, (HeadRec *)ctx->b_misc[0].p);
  mark(2);
  WC_TRY(wc_ensure(ctx, ctx->b_misc[5], (size_t)A.nslots * sizeof(SplitJob)));
  A.split_jobs = (SplitJob *)ctx->b_misc[5].p;
  A.prof = nullptr;
#ifdef WC_PROF_ROOTS
  WC_TRY(wc_ensure(ctx, ctx->b_misc[6], (size_t)A.nslots * 32));
  WC_HIP(ctx, hipMemsetAsync(ctx->b_misc[6].p, 0, (size_t)A.nslots * 32, st));
  A.prof = (uint32_t *)ctx->b_misc[6].p;
#endif
  WC_TRY(wc_ensure(ctx, ctx->b_misc[7], (size_t)A.nslots * (9 * kMom * 8 + 4)));
  A.node_tot = (double *)ctx->b_misc[7].p;
  A.root_ncand = (uint32_t *)(A.node_tot + (size_t)A.nslots * 9 * kMom);
  const K *skeys = (const K *)ctx->b_keys[1].p;  // sorted keys (general path only; the run path never expands them)
  if (fast_pts) {
    if constexpr (sizeof(K) == 4) {
      if (ctx->ex.unordered)  // the previous sweep had no run structure: order-independent streaming
        k_roots_banks<K, 1, true><<<kRootsGrid, 64, 0, st>>>(A, skeys);
      else
        k_roots<K, 1, true><<<kRootsGrid, 64, 0, st>>>(A, skeys);  // stream root + layer 1
      mark(3);
      k_roots_emit<K, true><<<kEmitGrid, 64, 0, st>>>(A, skeys);  // node tests + emission
    }
  } else {
    k_roots_banks<K, 1, false><<<kRootsGrid, 64, 0, st>>>(A, skeys);  // order-independent streaming (no run structure on this path)
    mark(3);
    k_roots_emit<K, false><<<kEmitGrid, 64, 0, st>>>(A, skeys);
  }
  static_assert(sizeof(RootsArgs) <= sizeof(ctx->ex.roots_args), "ctx.h: roots_args too small");
  memcpy(ctx->ex.roots_args, &A, sizeof(A));
  ctx->ex.total_slots = total_slots;
  ctx->ex.bin_cap = bin_cap;
  ctx->ex.slot_end_bit = slot_end_bit;
  ctx->ex.fast_slots = fast_slots;
  ctx->ex.tail = fast_pts ? &pipeline_tail<K, (sizeof(K) == 4)> : &pipeline_tail<K, false>;
  // On regular scenes no root is queued for the layer-2 pass, and even an idle launch of it costs ~4.5 us: it is skipped
  // when the previous call queued nothing (consecutive sweeps look alike); finish() runs the tail again if that guess
  // was wrong.
  return ctx->ex.tail(ctx, ctx->ex.last_splits > 0);
}

}  // namespace

extern "C" int wc_voxel_keys(wc_ctx *ctx, const wc_points *pts, int32_t *d_keys_xyz) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !pts || !d_keys_xyz) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  if (pts->n == 0) return WC_OK;
  k_voxel_keys<<<(unsigned)((pts->n + 255) / 256), 256, 0, ctx->stream>>>(*pts, (double)ctx->P.voxel_size, d_keys_xyz);
  WC_HIP(ctx, hipGetLastError());
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WC_OK;
}

extern "C" int wc_extract_surfels_enqueue(wc_ctx *ctx, const wc_points *pts, double t_lo, double t_hi, wc_surfel *d_out,
                                          wc_surfel_id *d_ids, uint64_t cap) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !pts) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  if (pts->n >= (1ull << 32)) return wc_fail(ctx, WC_ERR_ARG, "at most 2^32-1 points per call");
  ctx->ex.active = true;
  ctx->ex.ticket = 0;  // (only this call's fx_tail may arm the completion ticket - a stale one would let finish skip its wait)
  ctx->ex.pts = *pts;
  ctx->ex.d_out = d_out;
  ctx->ex.d_ids = d_ids;
  ctx->ex.cap = cap;
  ctx->ex.wide = false;
  for (int q = 0; q < 16; ++q) ctx->h_status[q] = 0;
  if (pts->n == 0) return WC_OK;
  if (t_lo > t_hi) {  // no hint: read the first and last timestamp back (input is time ordered)
    WC_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[0], pts->time, 8, hipMemcpyDeviceToHost, ctx->stream));
    WC_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[1], (const char *)pts->time + (pts->n - 1) * pts->time_stride, 8,
                               hipMemcpyDeviceToHost, ctx->stream));
    WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    t_lo = ctx->h_mail[0];
    t_hi = ctx->h_mail[1];
  }
  {
    // A surfel's timestamp is sum(t) / n formed in fp64 (surfel_extraction.cc:36-51): with epoch-sized stamps the running sum
    // rounds at ~1e-5 s, so the mean of a cluster can fall a little outside [first, last] point time.  The hint is widened
    // by 4096 ulp of the larger bound (1 ms at 1.6e9 s) on both sides; a surfel outside the widened range is an error.
    const double mag = std::max(std::fabs(t_lo), std::fabs(t_hi));
    const double m = 4096.0 * (std::nextafter(mag, INFINITY) - mag);
    t_lo -= m;
    t_hi += m;
  }
  ctx->ex.t_lo = t_lo;
  ctx->ex.t_hi = t_hi;
  // after a bin overflow of the run-binned point sort (very many points in few voxels, or points in no spatial order)
  // the next calls go to the radix-sort path directly; the fast path is tried again every 16th call
  ctx->ex.general = ctx->ex.general_calls > 0 || ctx->dev.no_bucket_sort != 0;
  if (ctx->ex.general_calls > 0) --ctx->ex.general_calls;
  ctx->ex.order_general = false;
  ctx->ex.fx_active = fx_applicable(ctx, pts->n, t_lo, t_hi);
  if (ctx->ex.fx_active && ctx->ex.fx_skip_calls > 0) {  // recent sweeps had to be repeated on the exact path: go there directly for a while
    --ctx->ex.fx_skip_calls;
    ctx->ex.fx_active = false;
  }
  if (ctx->ex.fx_active) {
    const int rc = run_pipeline_fast(ctx, *pts, t_lo, t_hi, d_out, d_ids, cap);
    if (rc != WC_OK) ctx->ex.fx_dirty = true;
    return rc;
  }
  return run_pipeline<uint32_t>(ctx, *pts, t_lo, t_hi, d_out, d_ids, cap, !ctx->ex.general, true);
}

extern "C" int wc_extract_surfels_finish(wc_ctx *ctx, uint64_t *h_n_out) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !ctx->ex.active) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  ctx->ex.active = false;
  if (h_n_out) *h_n_out = 0;
  if (ctx->ex.pts.n == 0) return WC_OK;
  // (Round 3, tried: a completion ticket stored to the pinned mailbox by k_slot_emit's last workgroup, so that this wait reads host
  // memory instead of waiting for the stream - as the LM loop does with k_post_reduce's single workgroup.  "Last of 1 024
  // workgroups" needs a count with a device-scope release in front of every workgroup's increment, and on this chip - one L2 per
  // XCD - that fence writes the XCD's L2 back: 43 -> 70 us per sweep with a two-stage count, 130 us with one counter.  Without the
  // fences a copy on another stream could read the surfels before they have left L2.  The stream wait stays.)
  auto wait = [&]() -> int {  // sweep done (its ticket has arrived, or the stream is idle); fold the mailbox flag words (raise_flag) into h_status[1]
    bool arrived = false;
    if (const uint32_t ticket = ctx->ex.ticket) {  // (fx_tail: the ticket kernel was the last thing enqueued)
      ctx->ex.ticket = 0;
      const uint32_t *w = (const uint32_t *)ctx->h_status + kMailTicket;
      const auto t0 = std::chrono::steady_clock::now();
      for (uint32_t spin = 0; !arrived; ++spin) {
        arrived = __atomic_load_n(w, __ATOMIC_ACQUIRE) == ticket;
        if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
      }
    }
    if (!arrived) WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 7; ++i)
      if (ctx->h_status[8 + i]) ctx->h_status[1] |= 1u << i;
    return WC_OK;
  };
  WC_TRY(wait());
  if (ctx->ex.fx_active) {
    // layer-2 nodes were queued but the pair of launches was skipped (the previous sweep had none): run it now
    if (!ctx->ex.layer2_done && ctx->P.max_layer >= 2 && ctx->h_status[4] > 0 && !(ctx->h_status[1] & (kFlagFxFallback | kFlagKeyRange))) {
      ctx->ex.last_splits = ctx->h_status[4];
      WC_TRY(fx_tail(ctx, true));
      WC_TRY(wait());
    }
    const uint32_t fl = ctx->h_status[1];
    ctx->ex.fx_last_flags = fl;
    ctx->ex.last_splits = ctx->h_status[4];
    if (fl & (kFlagFxFallback | kFlagKeyRange | kFlagSlotBinOverflow | kFlagTimeRange)) {
      // a decision too close to its threshold, a table at capacity, a node spanning > 16 time bins, ...: the tables are put
      // back to zero and the sweep is repeated on the exact path (below)
      ++ctx->ex.fx_fallbacks;
      ctx->ex.fx_last_why = 0;
      for (int i = 0; i < 24; ++i)
        if (ctx->h_status[32 + i]) ctx->ex.fx_last_why |= 1u << i;
      if (ctx->ex.fx_last_why & (1u << 4)) ctx->ex.fx_spill_full = true;  // the spill pool overflowed: full size from now on
      // exponential back-off: a sweep the default path cannot finish (a node spanning more than 16 time bins, more roots than
      // blocks, ...) is usually followed by more of its kind; a gate that merely fell inside the noise band is not
      ctx->ex.fx_backoff = std::min(32u, std::max(1u, ctx->ex.fx_backoff * 2u));
      ctx->ex.fx_skip_calls = ctx->ex.fx_backoff - 1u;
      ctx->ex.fx_dirty = true;        // tables (a root at capacity, roots waiting for a layer-2 pass that never ran) ...
      ctx->ex.fx_ctrl_ready = false;  // ... and control blocks are set up anew by the next fast sweep
      ctx->ex.fx_active = false;
      ctx->ex.general = ctx->ex.general_calls > 0;
      WC_TRY(run_pipeline<uint32_t>(ctx, ctx->ex.pts, ctx->ex.t_lo, ctx->ex.t_hi, ctx->ex.d_out, ctx->ex.d_ids, ctx->ex.cap, !ctx->ex.general, true));
      WC_TRY(wait());
    } else {
      const uint32_t n_fast = ctx->h_status[0];
      ctx->ex.fx_long_lists = ctx->h_status[15] != 0u;  // lists of several records per (node, time slot): merged first next time
      ctx->ex.fx_long_lists2 = ctx->h_status[16] != 0u;  // ... among the layer-2 nodes' lists
      if (h_n_out) *h_n_out = n_fast;
      ctx->ex.fx_parity ^= 1;  // the other control block has been cleared by this sweep's k_slot_emit
      ctx->ex.fx_backoff = 0;
      if (n_fast > ctx->ex.cap)
        return wc_fail(ctx, WC_ERR_CAPACITY, "output capacity %llu < %u surfels", (unsigned long long)ctx->ex.cap, n_fast);
      return WC_OK;
    }
  }
  // a bucket had more runs than the LDS capacity of k_pt_bucket (but fits its bin): repeat one capacity up (sticky)
  while ((ctx->h_status[1] & kFlagLdsOverflow) && !(ctx->h_status[1] & (kFlagBucketOverflow | kFlagKeyRange)) && !ctx->ex.general &&
         ctx->ex.lds_cap < kPtBinMax) {
    ctx->ex.lds_cap *= 2;
    WC_TRY(run_pipeline<uint32_t>(ctx, ctx->ex.pts, ctx->ex.t_lo, ctx->ex.t_hi, ctx->ex.d_out, ctx->ex.d_ids, ctx->ex.cap, true, true));
    WC_TRY(wait());
  }
  if ((ctx->h_status[1] & kFlagLdsOverflow) && !ctx->ex.general) ctx->h_status[1] |= kFlagBucketOverflow;  // still too large
  if ((ctx->h_status[1] & kFlagBucketOverflow) && !(ctx->h_status[1] & kFlagKeyRange) && !ctx->ex.general) {
    // a bin of the run-binned point sort overflowed: redo with the general radix sort
    ctx->ex.general = true;
    ctx->ex.general_calls = 15;
    WC_TRY(run_pipeline<uint32_t>(ctx, ctx->ex.pts, ctx->ex.t_lo, ctx->ex.t_hi, ctx->ex.d_out, ctx->ex.d_ids, ctx->ex.cap, false, true));
    WC_TRY(wait());
  }
  if ((ctx->h_status[1] & kFlagKeyRange) && !ctx->ex.wide) {
    // the sweep spans more than +-512 root voxels around its first point: redo with 21-bit-per-axis keys
    ctx->ex.wide = true;
    WC_TRY(run_pipeline<uint64_t>(ctx, ctx->ex.pts, ctx->ex.t_lo, ctx->ex.t_hi, ctx->ex.d_out, ctx->ex.d_ids, ctx->ex.cap, false, true));
    WC_TRY(wait());
  }
  if ((ctx->h_status[1] & kFlagSlotBinOverflow) && !ctx->ex.order_general) {
    // very many surfels inside one 1/4096 of the sweep's time span: redo with the radix sort of the slot keys
    ctx->ex.order_general = true;
    if (ctx->ex.wide)
      WC_TRY(run_pipeline<uint64_t>(ctx, ctx->ex.pts, ctx->ex.t_lo, ctx->ex.t_hi, ctx->ex.d_out, ctx->ex.d_ids, ctx->ex.cap, false, false));
    else
      WC_TRY(run_pipeline<uint32_t>(ctx, ctx->ex.pts, ctx->ex.t_lo, ctx->ex.t_hi, ctx->ex.d_out, ctx->ex.d_ids, ctx->ex.cap, !ctx->ex.general, false));
    WC_TRY(wait());
  }
  if (!ctx->ex.layer2_done && ctx->h_status[4] > 0 && ctx->ex.tail) {  // roots were queued for the skipped layer-2 pass
    ctx->ex.last_splits = ctx->h_status[4];
    WC_TRY(ctx->ex.tail(ctx, true));
    WC_TRY(wait());
  }
  const uint32_t flags = ctx->h_status[1];
  const uint32_t n_out = ctx->h_status[0];
  ctx->ex.last_splits = ctx->h_status[4];
  // run statistics of the run-binned sort: fewer than four points per run on average = no run structure (a spinning
  // multi-beam sensor in firing order): the next sweep streams with k_roots_banks right away
  if (!ctx->ex.general && !ctx->ex.wide && ctx->h_status[5] > 0) ctx->ex.unordered = (uint64_t)ctx->h_status[5] * 4 > ctx->ex.pts.n;
  if (h_n_out) *h_n_out = n_out;
  // the control block of the NEXT call is cleared now, asynchronously: it runs while the host turns around
  if (clear_ctrl(ctx) == WC_OK) ctx->ex.precleared = true;
  if (flags & kFlagKeyRange) return wc_fail(ctx, WC_ERR_ARG, "point cloud extent exceeds 2^20 root voxels");
  if (flags & kFlagSlotOverflow) return wc_fail(ctx, WC_ERR_HIP, "internal: candidate slot overflow");
  if (flags & kFlagTimeRange) return wc_fail(ctx, WC_ERR_ARG, "surfel timestamp outside the [t_lo, t_hi] hint");
  if (n_out > ctx->ex.cap) return wc_fail(ctx, WC_ERR_CAPACITY, "output capacity %llu < %u surfels",
                                          (unsigned long long)ctx->ex.cap, n_out);
  return WC_OK;
}

extern "C" int wc_extract_surfels(wc_ctx *ctx, const wc_points *pts, double t_lo, double t_hi, wc_surfel *d_out,
                                  wc_surfel_id *d_ids, uint64_t cap, uint64_t *h_n_out) {
  wc_dev_guard dg_(ctx);
  WC_TRY(wc_extract_surfels_enqueue(ctx, pts, t_lo, t_hi, d_out, d_ids, cap));
  return wc_extract_surfels_finish(ctx, h_n_out);
}

// ---- K sweeps through ONE launch chain (include/wildcat_hip.h) ------------------------------------------------------------------
// A sweep of a million points is three launches of 5 - 18 us whose first microseconds are launch head and whose duration is one
// wavefront's chain of dependent steps: 488 node wavefronts on 1024 SIMDs.  K sweeps that are known together - a window replayed
// from a log, the sweeps of several sensors - share the chain: every sweep keeps its own tables (a sub-context each: the
// tables are zero at rest and cleaned behind the sweep, as for single sweeps), the kernels run once over all of them.
extern "C" int wc_extract_surfels_batch_enqueue(wc_ctx *ctx, const wc_sweep_job *jobs, int K) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !jobs || K < 1 || K > 64) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  while ((int)ctx->batch_subs.size() < K) {
    wc_ctx *sub = nullptr;
    const int rc = wc_ctx_create(&ctx->P, ctx->device, &sub);
    if (rc != WC_OK) return wc_fail(ctx, rc, "wc_extract_surfels_batch: no sub-context");
    ctx->batch_subs.push_back(sub);
  }
  hipStream_t st = ctx->stream;
  std::vector<FxArgs> args;
  std::vector<SlotEmitArgs> emit;
  std::vector<uint32_t> s_acc(1, 0u), s_nodes(1, 0u), s_g2(1, 0u), flags;
  bool any_l2 = false, any_merge = false;
  for (int k = 0; k < K; ++k) {
    wc_ctx *sub = ctx->batch_subs[k];
    if (sub->stream != st) WC_TRY(wc_ctx_set_stream(sub, st));
    if (std::memcmp(&sub->P, &ctx->P, sizeof(wc_params)) != 0) WC_TRY(wc_ctx_set_params(sub, &ctx->P));
    if (std::memcmp(&sub->dev, &ctx->dev, sizeof(wc_dev_opts)) != 0) {  // (the parent's development options reach the sweeps)
      sub->dev = ctx->dev;
      sub->ex.fx_backoff = sub->ex.fx_skip_calls = 0;
    }
    sub->ex.batch_defer = true;
    sub->ex.deferred = false;
    const int rc = wc_extract_surfels_enqueue(sub, &jobs[k].pts, jobs[k].t_lo, jobs[k].t_hi, jobs[k].d_out, jobs[k].d_ids, jobs[k].cap);
    sub->ex.batch_defer = false;
    if (rc != WC_OK) {
      // the sub-contexts prepared so far (tables and control blocks set up, their kernels never launched) are withdrawn: their
      // tables are cleared before the next use and a later finish() has nothing to wait for
      for (int j = 0; j <= k; ++j) {
        wc_ctx *pj = ctx->batch_subs[j];
        if (pj->ex.deferred || j == k) pj->ex.fx_dirty = true, pj->ex.active = false, pj->ex.deferred = false;
      }
      return wc_fail(ctx, rc, "wc_extract_surfels_batch (sweep %d): %s", k, wc_last_error(sub));
    }
    if (!sub->ex.deferred) continue;  // (empty, or a sweep the default path does not take: enqueued on its own, on the same stream)
    FxArgs A;
    std::memcpy(&A, sub->ex.roots_args, sizeof(A));
    args.push_back(A);
    const uint32_t g2 = std::min(std::min(256u * 8u, sub->ex.fx_ngrid), std::max(64u, sub->ex.last_splits));
    s_acc.push_back(s_acc.back() + sub->ex.fx_tiles);
    s_nodes.push_back(s_nodes.back() + sub->ex.fx_ngrid);
    s_g2.push_back(s_g2.back() + g2);
    const uint32_t f = (sub->ex.layer2_done ? 1u : 0u) | ((sub->ex.fx_long_lists || sub->ex.fx_long_lists2) ? 2u : 0u);  // (a batch merges both levels or none)
    flags.push_back(f);
    any_l2 = any_l2 || (f & 1u);
    any_merge = any_merge || (f & 2u);
    uint32_t *next_ctrl = (uint32_t *)sub->b_fx[5].p + (size_t)(sub->ex.fx_parity ^ 1) * kCtrlWords;
    emit.push_back(SlotEmitArgs{A.slot_counts, A.slot_bins, sub->ex.bin_cap, (const wc_surfel *)sub->b_slots.p, (const wc_surfel_id *)sub->b_slot_ids.p,
                                A.status, sub->ex.d_out, sub->ex.d_ids, sub->ex.cap, next_ctrl, (uint32_t)kCtrlWords});
  }
  const int Kd = (int)args.size();
  if (Kd == 0) return WC_OK;
  // one upload: {FxArgs[Kd] | SlotEmitArgs[Kd] | three prefix tables | flags}
  const size_t o_emit = (sizeof(FxArgs) * Kd + 15) & ~(size_t)15, o_tab = (o_emit + sizeof(SlotEmitArgs) * Kd + 15) & ~(size_t)15;
  const size_t bytes = o_tab + (size_t)(3 * (Kd + 1) + Kd) * 4;
  std::vector<unsigned char> host(bytes, 0);
  std::memcpy(host.data(), args.data(), sizeof(FxArgs) * Kd);
  std::memcpy(host.data() + o_emit, emit.data(), sizeof(SlotEmitArgs) * Kd);
  uint32_t *tab = (uint32_t *)(host.data() + o_tab);
  std::memcpy(tab, s_acc.data(), (Kd + 1) * 4);
  std::memcpy(tab + (Kd + 1), s_nodes.data(), (Kd + 1) * 4);
  std::memcpy(tab + 2 * (Kd + 1), s_g2.data(), (Kd + 1) * 4);
  std::memcpy(tab + 3 * (Kd + 1), flags.data(), Kd * 4);
  WC_TRY(wc_ensure(ctx, ctx->b_batch, bytes));
  WC_HIP(ctx, hipMemcpyAsync(ctx->b_batch.p, host.data(), bytes, hipMemcpyHostToDevice, st));
  WC_HIP(ctx, hipStreamSynchronize(st));  // (pageable staging: `host` dies with this scope; ~10 us once per K sweeps)
  const unsigned char *d = (const unsigned char *)ctx->b_batch.p;
  const uint32_t *dt = (const uint32_t *)(d + o_tab);
  FxBatch B;
  B.args = (const FxArgs *)d, B.K = Kd, B.flags = dt + 3 * (Kd + 1);
  B.start = dt;
  k_fx_acc_b<1><<<s_acc.back(), kFxThreads, 0, st>>>(B);
  B.start = dt + (Kd + 1);
  if (any_merge) k_fx_merge_b<1><<<s_nodes.back(), 128, 0, st>>>(B);
  k_fx_walk_b<1><<<s_nodes.back(), 64, 0, st>>>(B);
  k_fx_test_b<1><<<s_nodes.back(), 64, 0, st>>>(B);
  if (any_l2) {
    B.start = dt;
    k_fx_acc_b<2><<<s_acc.back(), kFxThreads, 0, st>>>(B);
    B.start = dt + 2 * (Kd + 1);
    if (any_merge) k_fx_merge_b<2><<<s_g2.back(), 128, 0, st>>>(B);
    k_fx_walk_b<2><<<s_g2.back(), 64, 0, st>>>(B);
    k_fx_test_b<2><<<s_g2.back(), 64, 0, st>>>(B);
  }
  k_slot_emit_b<<<(unsigned)Kd * (kBuckets / 4), 256, 0, st>>>((const SlotEmitArgs *)(d + o_emit), Kd);
  WC_HIP(ctx, hipGetLastError());
  return WC_OK;
}

extern "C" int wc_extract_surfels_batch_finish(wc_ctx *ctx, uint64_t *h_n_out, int K) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !h_n_out || K < 1 || K > (int)ctx->batch_subs.size()) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  int first_rc = WC_OK;
  for (int k = 0; k < K; ++k) {
    wc_ctx *sub = ctx->batch_subs[k];
    h_n_out[k] = 0;
    if (!sub->ex.active) continue;
    // (the sweep's own finish: waits for the shared stream, runs a layer-2 pass that turned out to be needed, repeats a sweep whose
    // gates fell inside the noise band on the exact path - all on the sub-context, as for a single sweep)
    const int rc = wc_extract_surfels_finish(sub, &h_n_out[k]);
    if (rc != WC_OK && first_rc == WC_OK) first_rc = wc_fail(ctx, rc, "wc_extract_surfels_batch (sweep %d): %s", k, wc_last_error(sub));
  }
  return first_rc;
}

extern "C" int wc_extract_profile(wc_ctx *ctx, int enable) {
  wc_dev_guard dg_(ctx);
  if (!ctx) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  if (enable && !ctx->ex_ev[0])
    for (int i = 0; i < 8; ++i) WC_HIP(ctx, hipEventCreate(&ctx->ex_ev[i]));
  ctx->ex_prof = enable != 0;
  ctx->ex_prof_mode = enable;  // 1: an event after every kernel group (each costs ~5 us of stream time); 2: first and last only
  return WC_OK;
}

extern "C" int wc_extract_stage_ms(wc_ctx *ctx, float *h_ms5) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !h_ms5 || !ctx->ex_ev[0]) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  WC_HIP(ctx, hipEventSynchronize(ctx->ex_ev[5]));
  if (ctx->ex_prof_mode == 2) {  // bracket only: the whole stage in slot 1
    for (int i = 0; i < 5; ++i) h_ms5[i] = 0.f;
    WC_HIP(ctx, hipEventElapsedTime(&h_ms5[1], ctx->ex_ev[1], ctx->ex_ev[5]));
    return WC_OK;
  }
  for (int i = 0; i < 5; ++i) WC_HIP(ctx, hipEventElapsedTime(&h_ms5[i], ctx->ex_ev[i], ctx->ex_ev[i + 1]));
  return WC_OK;
}

extern "C" int wc_debug_status(wc_ctx *ctx, uint32_t *h_out64) {
  wc_dev_guard dg_(ctx);  // profiling aid: status words [0..16) + section timers
  if (!ctx || !h_out64) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  // (the device copy has been cleared for the next call already: words 0..15 come from the host mailbox of the last call)
  for (int i = 0; i < 64; ++i) h_out64[i] = i < 16 ? ctx->h_status[i] : 0u;
  h_out64[59] = ctx->ex.fx_long_lists ? 1u : 0u;  // the next fast sweep runs k_fx_merge (the last one had long record lists)
  h_out64[60] = ctx->ex.fx_fallbacks;  // sweeps the fast path handed to the exact path so far
  h_out64[61] = ctx->ex.fx_active ? 1u : 0u;  // the last sweep was completed by the fast (integer-moment) path
  h_out64[62] = ctx->ex.fx_last_flags;
  h_out64[63] = ctx->ex.fx_last_why;  // bit i: call site i of fx_fallback() fired in the last sweep that fell back
#ifdef WC_PROF_ROOTS
  // average the per-root section timers into words [16, 24), number of timed roots in word 24
  const size_t nslots = ctx->ex.pts.n / (size_t)(ctx->P.min_points + 1) + 1;
  std::vector<uint32_t> rows(nslots * 8);
  WC_HIP(ctx, hipMemcpy(rows.data(), ctx->b_misc[6].p, nslots * 32, hipMemcpyDeviceToHost));
  double sum[8] = {0};
  uint32_t cnt = 0;
  for (size_t r = 0; r < nslots; ++r)
    if (rows[r * 8 + 1]) {
      ++cnt;
      for (int i = 0; i < 8; ++i) sum[i] += rows[r * 8 + i];
    }
  for (int i = 0; i < 8; ++i) h_out64[16 + i] = cnt ? (uint32_t)(sum[i] / cnt) : 0;
  h_out64[24] = cnt;
#endif
  return WC_OK;
}

// wc_ctx_warmup: loads this translation unit's code object (the runtime loads it at the first launch of any of its kernels)
int wc_touch_extract() {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, (const void *)k_voxel_keys) == hipSuccess ? WC_OK : WC_ERR_HIP;
}
