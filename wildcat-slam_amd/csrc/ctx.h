// ctx.h — internal definition of the opaque wc_ctx: one HIP stream, growable scratch buffers in HBM, pinned
// host mailbox for the few words that travel back (counts, status flags), last-error text.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/wildcat_hip.h"

struct wc_window_state;  // window.hip

struct wc_buf {
  void *p = nullptr;
  size_t cap = 0;
  bool pooled = false;  // from the device's stream-ordered memory pool (hipMallocAsync): released with hipFreeAsync
  bool plain = false;   // never from the pool: buffers handed to the communicator's collectives (RCCL sees ordinary hipMalloc memory)
};

// Knock-out bits of the extraction kernels (development option debug_skip) exist in a -DWC_DEV_KNOBS build only: the release kernels
// carry no profiling branch; the same switch gates the compile-time instrumentation of window.hip
// (WC_LIN_KNOCK, WC_GATHER_KNOCK, WC_PROF_LIN, WC_PROF_CHOL, WC_PCR_NOINV, WC_PCR_NOR).
#ifdef WC_DEV_KNOBS
#define WC_DBG(P, bit) ((P).dbg & (bit))
#else
#define WC_DBG(P, bit) 0
#undef WC_LIN_KNOCK
#undef WC_GATHER_KNOCK
#undef WC_PROF_LIN
#undef WC_PROF_CHOL
#undef WC_PCR_NOINV
#undef WC_PCR_NOR
#endif

// Development options of a context (wc_ctx_set_dev_option; include/wildcat_hip.h lists them).  They pin choices the library
// otherwise makes from the call's sizes, or knock parts of a kernel out for timing runs.  The release build reads NO environment
// variable that changes the executed path: a `-DWC_DEV_KNOBS` build (profiles/dev) seeds these fields from the WC_* variables of
// DESIGN 5.1 when a context is created; everything else goes through the explicit call.
struct wc_dev_opts {
  int exact_sums = 0;        // contexts behave as if wc_params.exact_sums were 1
  int debug_skip = 0;        // knock-out bits of the default extraction path (results are WRONG; timing runs only)
  int fx_merge_min = 3;      // list length from which the next sweep runs k_fx_merge
  int fx_split = -1;         // node stage of the default extraction: 0 fused, 1 two kernels, -1 by size
  int no_bucket_sort = 0;    // exact path: radix sort instead of the run-binned sort
  int ex_sync = 0;           // extraction: finish waits for the stream instead of the completion ticket
  int kd_leaf = 0;           // matcher: target leaf size of the kd-tree (0: 8)
  int knn_group = -1;        // matcher walk: 0 one lane per query, 1 eight lanes per query, -1 by size
  int knn_sort = -1;         // two-set searches order their queries by leaf: 0 never, 1 always, -1 by the rule in match.hip
  int knn_early = 1;         // two-set searches bound their walks by the nearest gate-passing candidate too (0: plain k-NN walks, rounds 4 - 5)
  int match_pair_serial = 0; // wc_match_pair runs its searches one after the other on the ctx
  int match_pair_swap = 0;   // the sliding-window search on the helper instead of the fixed-window one
  int match_pair_hold = 1;   // wc_match_pair: the fixed-window search's walk waits for the sliding-window search's tree (match.hip: wc_pair_sync)
  int lin_imu_apart = 0, lin_unary_apart = 0, lin_post_apart = 0;  // the linearisation's families / mailbox as launches of their own
  int lm_dense = 0;          // round 2's LM step: dense Cholesky of all 12 ns unknowns
  int lin_pair = 1;          // binary assembly pieces of at most 128 records two to a workgroup (0: one each, rounds 2 - 5)
  int lin_unary_chunks = 0;  // chunks of 256 records per unary piece (0: the library's 4; 1: rounds 2 - 5's pieces)
  int pcr_full_width = 0;    // bias elimination: every reduction level over all columns of the right-hand sides (rounds 3 - 5) instead of their bands
  int lm_side_stream = 1;    // two-collective form: the large collective on a side stream (1: with the in-library RCCL binding; 0: never; 2: always - tests)
  int lm_one_collective = 0; // sharded windows: rounds 3 - 5's ONE all-reduce per linearisation (IMU triples sharded too) instead of the two-collective form
  int dbg_lm = 0;            // experiment bits of the LM solve's kernels (timing runs of a development session; results may be WRONG)
  int lm_back_chunks = 0;    // rounds 2 - 5's back substitution (chunk solves + products) and tail launches instead of k_back_mul + the fused tail
  int lm_sync = 0;           // wait for the stream instead of the mailbox ticket
  int lm_eval_pass = 0;      // a cost-only pass for the candidate instead of a linearisation
  int pcr_ahead = 1;         // the bias elimination's level 0 of the NEXT iteration enqueued behind the candidate's linearisation (0: at the iteration's start)
  int lm_dense_radius = 10;  // iterations whose trust-region radius exceeds 10^value take the dense step (0: never)
};
// logging-only switches (they print; they never change a result): read once per process from the environment in every build
inline bool wc_log_env(const char *name) { return getenv(name) != nullptr; }

struct wc_ctx {
  int device = 0;
  wc_params P;
  wc_dev_opts dev;
  bool pool_ok = false;  // the device has a stream-ordered memory pool (wc_ensure allocates from it)
  hipMemPool_t pool = nullptr;  // the process's private pool on this device (ctx.hip: wc_pool_acquire), shared by its contexts
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;
  // scratch buffers (device), grown on demand and kept for the lifetime of the ctx
  wc_buf b_ex_ctrl;  // the extraction's control block (status words, bucket / bin counters): never shared, cleared ahead of time
  wc_buf b_keys[2], b_vals[2], b_sorttmp, b_slots, b_slot_ids, b_slot_keys[2], b_slot_idx[2], b_cand, b_cand_meta,
      b_status, b_misc[8], b_route[4], b_fx[10];
  // wc_match: what the last search's traversal touched (sampled, see k_knn_tree): wide nodes, leaves, points, exact distances, queries
  double match_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  wc_buf b_kd[10];  // match_tree.inc: planes (dim, value), bucket ids, padded counters, starts x 2, index lists x 2, box heap, leaf ranges
  std::vector<wc_ctx *> batch_subs;  // sub-contexts of wc_extract_surfels_batch_* (own scratch, the parent's stream)
  wc_buf b_batch;
  wc_buf b_match_half;  // match.hip: the sorted features in single precision, 32 bytes per target (the first look of k_knn_tree)
  hipEvent_t ev_knn[2] = {nullptr, nullptr};
  // multi-GPU: the job's communicator (wc_ctx_set_comm / wc_comm_rccl_init)
  wc_comm comm{};
  bool have_comm = false;
  void *rccl = nullptr;  // the in-library RCCL communicator (comm.hip), if any
  // pinned host mailbox
  uint32_t *h_status_dev = nullptr;  // the device's address of h_status
  uint32_t *h_status = nullptr;  // pinned, 128 words: [0] n_emitted, [1] flags, ... ; [64..95]: the matcher's read-backs (round words, walk statistics)
  unsigned long long mail_ticket = 0;  // last ticket handed to a k_post_reduce (window.hip: wait_mail)
  wc_buf b_stage;                 // wc_d2h_strided: the packed elements on the device
  void *h_stage = nullptr;        // ... and their pinned landing place
  size_t h_stage_cap = 0;
  double *h_mail = nullptr;      // pinned: 64 doubles of mailbox (costs etc.) + 4096 doubles of staging (the window's unknowns)
  // pending extraction (enqueue/finish split)
  struct {
    bool active = false;
    wc_points pts;
    double t_lo, t_hi;
    wc_surfel *d_out;
    wc_surfel_id *d_ids;
    uint64_t cap;
    bool wide;
    bool general;
    bool order_general = false;  // this call orders the surfels with the radix sort (a time bin overflowed)
    int general_calls = 0;       // upcoming calls that start on the radix-sort path right away
    bool bucket_attr_set = false;  // hipFuncSetAttribute(k_pt_bucket) done on this ctx's device
    uint32_t lds_cap = 256;      // runs per bucket k_pt_bucket sorts in LDS (256 / 512 / 1024, grows with the data)
    bool unordered = false;      // the previous sweep had (almost) no run structure: stream with k_roots_banks
    uint32_t last_splits = 256;  // roots the previous call queued for the layer-2 pass (sizes / gates that launch)
    // the tail of the pipeline (layer-2 pass, surfel order, status read-back) is re-run by finish() when the call skipped
    // the layer-2 launch and roots were queued for it after all
    uint32_t ticket = 0, ticket_seq = 0;  // completion ticket of the sweep in flight (0: none - finish waits for the stream)
    bool fx_active = false;      // this call runs on the fast (integer-moment) path
    bool fx_dirty = false;       // the fast path's tables may hold garbage (an aborted sweep): memset before the next use
    uint32_t fx_last_flags = 0, fx_fallbacks = 0, fx_last_why = 0;
    bool fx_spill_full = false;
    bool fx_split = false;
    // batched extraction (wc_extract_surfels_batch_*): a sub-context prepares its sweep - tables, control block, kernel arguments
    // in roots_args - and leaves the launches to the parent, which runs K sweeps' kernels as one launch chain
    bool batch_defer = false, deferred = false;
    unsigned fx_tiles = 0, fx_ngrid = 0;  // the node stage of the current sweep runs as k_fx_walk + k_fx_test (extract_split.inc)  // the spill pool of the fast path overflowed once: sized for the worst case from then on
    bool fx_long_lists = false;  // the last fast sweep walked long record lists: k_fx_merge runs before k_fx_nodes
    bool fx_long_lists2 = false;  // ... the same for the layer-2 pass
    uint32_t fx_backoff = 0, fx_skip_calls = 0;  // sweeps that go straight to the exact path after fall-backs (exponential)
    bool fx_ctrl_ready = false;  // the fast path's two control blocks are initialised
    int fx_parity = 0;           // which of them the next fast sweep uses
    bool precleared = false;     // the control block has been cleared (on the stream) by the previous finish()
    bool layer2_done = true;
    int (*tail)(wc_ctx *, bool) = nullptr;
    alignas(16) unsigned char roots_args[768];
    uint64_t total_slots;
    uint32_t bin_cap;
    unsigned slot_end_bit;
    bool fast_slots;
  } ex;
  wc_window_state *win = nullptr;
  wc_ctx *aux = nullptr;  // helper context of wc_match_pair (second stream + scratch), owned by this ctx
  hipEvent_t ev_aux = nullptr;  // orders the helper's stream behind the ctx stream
  hipEvent_t ev_pair = nullptr; // wc_match_pair: behind the sliding-window search's tree build (match.hip: wc_pair_sync)
  void *pair_worker = nullptr;  // wc_match_pair's helper thread (match.hip: wc_pair_worker), freed through pair_worker_free
  void (*pair_worker_free)(void *) = nullptr;
  void *pair_sync = nullptr;  // match.hip: wc_pair_sync of the wc_match_pair call this context's search belongs to (null: a search of its own)
  // optional per-stage HIP events of the extraction pipeline (wc_extract_profile)
  bool ex_prof = false;
  int ex_prof_mode = 0;
  hipEvent_t ex_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

// Every extern "C" entry point runs on the ctx's device whatever the calling thread's current device is (a ctx created
// for device 1 and driven from a thread whose current device is 0 would otherwise allocate scratch on GPU 0 and launch
// kernels on a stream of GPU 1); the thread's previous device is restored on return.
struct wc_dev_guard {
  int prev = -1;
  bool switched = false;
  explicit wc_dev_guard(const wc_ctx *ctx) {
    if (ctx && hipGetDevice(&prev) == hipSuccess && prev != ctx->device) switched = hipSetDevice(ctx->device) == hipSuccess;
  }
  ~wc_dev_guard() {
    if (switched) (void)hipSetDevice(prev);
  }
  wc_dev_guard(const wc_dev_guard &) = delete;
  wc_dev_guard &operator=(const wc_dev_guard &) = delete;
};

inline int wc_fail(wc_ctx *ctx, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

#define WC_HIP(ctx, call)                                                                                   \
  do {                                                                                                      \
    hipError_t e_ = (call);                                                                                 \
    if (e_ != hipSuccess)                                                                                   \
      return wc_fail(ctx, WC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// grow-only device buffer.  Memory comes from the process's private stream-ordered pool on the device (hipMallocFromPoolAsync on the ctx
// stream; the pool's release threshold keeps freed blocks until the last context on the device is destroyed): growing a buffer is an enqueue of microseconds, neither
// the device synchronisation of a hipFree nor the ~0.3 - 1 ms of a hipMalloc - a kernel trace of the facade's stream (round 5) showed a
// sweep of 15 ms among sweeps of 7 when the window's record buffers crossed their size together, and 19 ms for the first search of a
// helper context (25 buffers).  Falls back to hipMalloc where the pool is not available.
inline void wc_buf_release(wc_ctx *ctx, wc_buf &b) {
  if (!b.p) return;
  if (b.pooled && ctx && ctx->stream && hipFreeAsync(b.p, ctx->stream) == hipSuccess) {
  } else {
    if (ctx && ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(b.p);
  }
  b.p = nullptr, b.cap = 0, b.pooled = false;
}
inline int wc_ensure(wc_ctx *ctx, wc_buf &b, size_t bytes) {
  if (bytes <= b.cap) return WC_OK;
  wc_buf_release(ctx, b);  // (stream ordered: what is enqueued on the ctx stream still sees the old block)
  // (slack: a window that grows by a sweep per call - the facade's first seconds - re-allocated several buffers on EVERY call with an
  // eighth of slack; now a buffer that follows the data is never smaller than 16 MB, one below 64 MB doubles, a larger one grows by
  // half: 288 GB of HBM make the slack free.  The floor is for requests of 64 KB or more; the ~40 buffers of a context that hold
  // status words, plane tables or mailboxes stay small - with the floor on everything a fully used context pinned 1 - 1.5 GB, times
  // the helper and batch sub-contexts)
  size_t want = bytes < ((size_t)64 << 20) ? 2 * bytes : bytes + bytes / 2;
  if (bytes >= ((size_t)64 << 10) && want < ((size_t)16 << 20)) want = (size_t)16 << 20;
  if (want < 4096) want = 4096;
  static const bool alloc_dbg = wc_log_env("WC_ALLOC_DEBUG");  // (read once per process)
  if (alloc_dbg) fprintf(stderr, "[alloc] %zu bytes wanted -> %zu\n", bytes, want);
  if (ctx->pool_ok && !b.plain && hipMallocFromPoolAsync(&b.p, want, ctx->pool, ctx->stream) == hipSuccess) {
    b.pooled = true;
  } else {
    (void)hipGetLastError();
    b.p = nullptr;
    WC_HIP(ctx, hipMalloc(&b.p, want));
    b.pooled = false;
  }
  b.cap = want;
  return WC_OK;
}
int wc_rccl_allreduce_on(wc_ctx *ctx, double *d_buf, uint64_t count, hipStream_t st);  // comm.hip: the RCCL binding on a given stream (-1: not installed)
#define WC_TRY(expr)            \
  do {                          \
    int rc_ = (expr);           \
    if (rc_ != WC_OK) return rc_; \
  } while (0)
