/*
 * wildcat_hip.h — C-ABI of libwildcat_hip.so, the MI355X (gfx950) implementation of the sliding-window
 * odometry hot path of kekeliu-whu/Wildcat-SLAM (reference call site src/odometry/lidar_odometry.cc:523-566).
 *
 * The reference has no FFI / plugin boundary: the seam is a set of C++ free functions and classes inside
 * src/odometry.  Each entry point below names the reference interface it replaces; INTEGRATION.md shows the
 * C++ glue a maintainer adds inside LidarOdometry::AddLidarScan to call them.
 *
 * Conventions
 *   - every function returns an int status: WC_OK (0) or a WC_ERR_* code; wc_last_error() gives the text.
 *     (The reference aborts through glog CHECKs instead; the host facade turns non-zero into a fatal log.)
 *   - pointers named d_* are DEVICE (HBM) pointers, h_* are host pointers.  Records are the PODs of wc_types.h.
 *   - one caller thread per wc_ctx; a ctx owns one HIP stream (replaceable through wc_ctx_set_stream) and all
 *     scratch memory.  Calls are synchronous on return unless stated otherwise.
 *   - no C++ types, exceptions or torch types cross this boundary.
 */
#ifndef WILDCAT_HIP_H_
#define WILDCAT_HIP_H_

#include "wc_types.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
  WC_OK = 0,
  WC_ERR_CAPACITY = 1,  /* output buffer too small; *n_out holds the needed count                   */
  WC_ERR_RANGE = 2,     /* timestamp outside the IMU / sample-state range (reference CHECKs)        */
  WC_ERR_ORDER = 3,     /* correspondence not (older, newer)  (CHECK_LT lidar_odometry.cc:256,301)  */
  WC_ERR_HIP = 10,      /* HIP runtime error                                                        */
  WC_ERR_ARG = 11,      /* bad argument                                                             */
  WC_ERR_NOGPU = 12,    /* no gfx950 device visible                                                 */
  WC_ERR_NUMERIC = 13   /* linear solve failed                                                      */
};

typedef struct wc_ctx wc_ctx;

/* library / device --------------------------------------------------------------------------------------------- */
const char *wc_version(void);
int wc_device_count(void);
void wc_params_default(wc_params *p); /* SURVEY.md §2.1 values: surfel_extraction.cc:327, knn_surfel_matcher.h:37-41,
                                         lio_config.h:10-14,32-45 */
int wc_ctx_create(const wc_params *params, int device, wc_ctx **out);
void wc_ctx_destroy(wc_ctx *ctx);
const char *wc_last_error(const wc_ctx *ctx);
int wc_ctx_set_stream(wc_ctx *ctx, void *hip_stream); /* NULL = the ctx's own stream */
int wc_ctx_set_params(wc_ctx *ctx, const wc_params *params);
/* Development options of ONE context - the only way to change what the release library executes besides wc_params.  The release
 * build reads no environment variable that alters a result or a code path (a stray WC_* variable cannot change what a node runs);
 * the variables it does read only PRINT: WC_ALLOC_DEBUG, WC_FX_DEBUG, WC_MATCH_DEBUG, WC_MATCH_TIMING, WC_WIN_DEBUG, WC_DEBUG_GATHER
 * (libwildcat_hip.so) and WC_ODOM_DEBUG (the facade).  A `-DWC_DEV_KNOBS` build (profiles/dev) additionally seeds the options below
 * from the upper-case WC_<NAME> variables when a context is created.  Options (value 0 / 1 unless stated; -1 = the library decides):
 *   exact_sums         contexts behave as if wc_params.exact_sums were 1
 *   debug_skip         knock-out bits of the default extraction path (timing runs only: results are wrong; the bits exist in a
 *                      -DWC_DEV_KNOBS build only - the release kernels carry no profiling branch and ignore the option)
 *   fx_merge_min       list length from which the next sweep merges record lists first (default 3)
 *   fx_split           node stage of the default extraction: 0 fused kernel, 1 two kernels, -1 by size
 *   no_bucket_sort     exact path: radix sort instead of the run-binned sort
 *   ex_sync            wc_extract_surfels_finish waits for the stream instead of the sweep's completion ticket
 *   kd_leaf            target leaf size of the matcher's kd-tree (0 = 8)
 *   knn_group          matcher walk: 0 one lane per query, 1 eight lanes per query, -1 by the call's sizes
 *   knn_early          1 (default): the matcher's walks are bounded by the nearest gate-passing candidate as well as by the k-th distance
 *                      (same pair lists; not when the neighbour lists themselves are asked for); 0: plain k-NN walks; 2: two-set searches only
 *   knn_sort           leaf-order sort of a two-set search's queries: 0 never, 1 always, -1 (default): from 40 000 queries when the plain
 *                      k-NN walks run (neighbour lists asked for, knn_early = 0), never with the early bound
 *   match_pair_hold    1 (default): wc_match_pair holds the fixed-window search's walk back until the other search's tree is built
 *   match_pair_serial  wc_match_pair runs its two searches one after the other on the ctx
 *   match_pair_swap    the sliding-window search on the helper context instead of the fixed-window one
 *   lin_imu_apart, lin_unary_apart, lin_post_apart   factor families / mailbox of a linearisation as launches of their own
 *   lm_dense, lm_eval_pass, lm_sync, lm_back_chunks   earlier forms of the LM step kept for A/B runs
 *   pcr_ahead (1)                                     0: the bias elimination's level 0 at the start of an iteration (rounds 3 - 4)
 *   pcr_full_width     bias elimination: every reduction level over all columns of its right-hand sides (rounds 3 - 5) instead of their bands
 *   lin_pair (1)       binary assembly pieces of at most 128 records two to a workgroup (0: one each - rounds 2 - 5; the same bits)
 *   lin_unary_chunks   chunks of 256 records per unary assembly piece (0 / 1: one - the default; 2 .. 4: long pieces, a launch of their own)
 *   lm_one_collective  sharded windows: rounds 3 - 5's ONE all-reduce per linearisation (IMU triples sharded too) instead of the
 *                      two-collective form (IMU factors replicated; 16-byte cost collective, then the pose corners)
 *   lm_side_stream (1) two-collective form: the large collective on a side stream beside the bias elimination (1: with the in-library RCCL
 *                      binding, 0: never, 2: always - the choreography with a communicator of callbacks, for tests)
 *   dbg_lm             experiment bits of the LM solve's kernels (development sessions; 0 in every measured or tested run)
 *   lm_dense_radius    iterations whose trust-region radius exceeds 10^value take the dense step (default 10; 0 = never)
 * Tests use it to run both forms of a choice on the same data.  Unknown names return WC_ERR_ARG. */
int wc_ctx_set_dev_option(wc_ctx *ctx, const char *name, int value);
/* Optional, for long-running callers (the facade calls it from its constructor): takes one-time costs out of the first calls - loads
 * the code object of every translation unit of the library, creates wc_match_pair's helper context and host thread, and takes
 * `reserve_bytes` (0: nothing) of HBM into the device's stream-ordered memory pool, from which every scratch buffer of a context grows
 * (growing a buffer is then an enqueue of microseconds, not a hipFree + hipMalloc).  No reference counterpart. */
int wc_ctx_warmup(wc_ctx *ctx, size_t reserve_bytes);

/* device memory helpers so that a host program needs no HIP headers ---------------------------------------------- */
int wc_dev_alloc(wc_ctx *ctx, size_t bytes, void **d_ptr);
int wc_dev_free(wc_ctx *ctx, void *d_ptr);
int wc_h2d(wc_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int wc_d2h(wc_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int wc_d2d(wc_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
int wc_memset(wc_ctx *ctx, void *d_dst, int value, size_t bytes);
/* one field of every record: n elements of elem_bytes, src_stride bytes apart on the device, packed on the host */
int wc_d2h_strided(wc_ctx *ctx, void *h_dst, const void *d_src, size_t elem_bytes, size_t src_stride, size_t n);
int wc_sync(wc_ctx *ctx);
/* HIP-event timing on the ctx stream (used by bench.py: torch.cuda.Event only sees torch's stream) */
int wc_timer_start(wc_ctx *ctx);
int wc_timer_stop_ms(wc_ctx *ctx, float *h_ms);

/* known-answer hooks for the library's own math (csrc/dmath.h), so that the reference's KATs (src/common/utils_test.cc:5-21)
 * run against the product code: on_device != 0 evaluates in a one-thread kernel on the ctx's GPU, on_device == 0 with the
 * host instantiation of the same header (ctx may be NULL).
 *   wc_selftest_so3 : out52 = Exp(v) quat (w,x,y,z) | Log(Exp(v)) | Jl | Jl_inv | Jr | Jr_inv | Hat   (3x3 row-major)
 *   wc_selftest_eig3: a9 symmetric row-major -> out12 = ascending eigenvalues | eigenvectors in columns (row-major)
 *   wc_selftest_quat: in12 = a(4) b(4) f p(3) -> out11 = slerp(a,f,b) | a*p (rotation) | a*b
 *   wc_selftest_so3_fused (device only): the fused forms the factor kernels evaluate (csrc/so3_fused.h) ->
 *                     out25 = Exp(v) quat | Jr(v) | Log(Exp(v)) | Jr_inv(Log(Exp(v))) */
int wc_selftest_so3(wc_ctx *ctx, const double v[3], int on_device, double out52[52]);
int wc_selftest_so3_fused(wc_ctx *ctx, const double v[3], double out25[25]);
int wc_selftest_eig3(wc_ctx *ctx, const double a9[9], int on_device, double out12[12]);
int wc_selftest_quat(wc_ctx *ctx, const double in12[12], int on_device, double out11[11]);
/* the damped solve's diagonal-block kernel on its own (tests/test_kat_gpu.py, profiles/dev/factor32.py): Cholesky factor L and L^-1 of a
 * 32 x 32 SPD matrix (row-major, lower part read); variant 0 = the library's block form (256 threads, fp64 matrix-core rank-4 updates);
 * h_clk[0] = shader clocks of the fastest of `reps` runs, h_clk[1] = 1 when every pivot was positive */
int wc_selftest_factor32(wc_ctx *ctx, int variant, int reps, const double *h_A, double *h_L, double *h_X, long long *h_clk);

/* surfel extraction --------------------------------------------------------------------------------------------- */
/* Replaces BuildSurfels(const std::vector<hilti_ros::Point>&, std::deque<Surfel::Ptr>&, GlobalMap&)
 * (src/odometry/surfel_extraction.h:145-147, .cc:316-337; call site lidar_odometry.cc:523-525).
 *   pts          descriptor with DEVICE pointers (time ascending, as the reference CHECKs at lidar_odometry.cc:491)
 *   d_out/d_ids  caller-allocated, capacity `cap` records; d_ids may be NULL
 *   h_n_out      number of surfels, sorted by ascending timestamp (surfel_extraction.cc:334), ties by id
 *   t_lo, t_hi   optional hint: all point timestamps lie in [t_lo, t_hi] (pass t_lo > t_hi to let the
 *                library read the first/last timestamp back from the device) */
int wc_extract_surfels(wc_ctx *ctx, const wc_points *pts, double t_lo, double t_hi, wc_surfel *d_out,
                       wc_surfel_id *d_ids, uint64_t cap, uint64_t *h_n_out);
/* asynchronous split of the same call for pipelined / benchmarked use: enqueue does no host synchronisation,
 * finish waits for the stream and returns the count and status. */
int wc_extract_surfels_enqueue(wc_ctx *ctx, const wc_points *pts, double t_lo, double t_hi, wc_surfel *d_out,
                               wc_surfel_id *d_ids, uint64_t cap);
int wc_extract_surfels_finish(wc_ctx *ctx, uint64_t *h_n_out);
/* K sweeps (1 <= K <= 64) through ONE launch chain: BuildSurfels is called once per sweep with a fresh GlobalMap
 * (lidar_odometry.cc:523-525), so sweeps are independent; when several are known together (a window replayed from a log, C3 / C4's
 * 5 / 20-sweep windows, several sensors) their kernels run as one launch each over all sweeps instead of three launches per
 * sweep - a 1 M-point sweep is launch-latency bound on its own.  Every sweep gets its own output buffers, count and - should a
 * gate fall inside the noise band - its own repetition on the exact path; results are those of K wc_extract_surfels calls, byte for
 * byte.  enqueue returns without waiting; finish fills h_n_out[K]. */
int wc_extract_surfels_batch_enqueue(wc_ctx *ctx, const wc_sweep_job *jobs, int K);
int wc_extract_surfels_batch_finish(wc_ctx *ctx, uint64_t *h_n_out, int K);
/* per-stage device time of the LAST enqueued extraction, measured with HIP events on the ctx stream:
 * h_ms5 = {per-call fills, voxel grouping of the points, root + layer-1 streaming, node tests + emission (+ layer 2),
 * time ordering + gather of the surfels}.  Enable first: 1 = an event after every kernel group (each event costs ~5 us of
 * stream time), 2 = one pair of events around the whole stage (the stage's time is then reported in h_ms5[1], the other
 * entries are 0), 0 = off. */
int wc_extract_profile(wc_ctx *ctx, int enable);
int wc_extract_stage_ms(wc_ctx *ctx, float *h_ms5);
/* raw status words of the last extraction (profiling aid; words 16.. hold per-section cycle sums when the library
 * is built with -DWC_PROF_ROOTS) */
int wc_debug_status(wc_ctx *ctx, uint32_t *h_out64);
/* root-voxel index of every point: VoxelLoc (src/odometry/surfel_extraction.h:55-64); d_keys_xyz = 3 int32 per point */
int wc_voxel_keys(wc_ctx *ctx, const wc_points *pts, int32_t *d_keys_xyz);

/* one cloud over several GPUs (BASELINE config 5) --------------------------------------------------- */
/* The reference has no counterpart (single thread): root voxels are independent after binning (surfel_extraction.cc:217-219,
 * :330-332) but each needs all its points in time order (:22-29), so a cloud shards by root voxel - see csrc/route.hip. */
/* Installing a communicator makes NO existing entry point a collective: only the *_sharded calls (wc_extract_surfels_sharded,
 * wc_gather_surfels, wc_match_sharded, wc_match_pair_sharded, wc_window_build_sharded and the solve of a problem built by it) use it. */
int wc_ctx_set_comm(wc_ctx *ctx, const wc_comm *comm); /* NULL removes it */
/* the in-library RCCL communicator (csrc/comm.hip; librccl.so is dlopen()ed): rank 0 creates the 128-byte unique id, the
 * launcher hands it to every rank, each rank calls wc_comm_rccl_init on its ctx.  Collectives run on the ctx's stream. */
int wc_comm_rccl_unique_id(char out128[128]);
int wc_comm_rccl_init(wc_ctx *ctx, int rank, int world, const char id128[128]);
int wc_comm_rccl_destroy(wc_ctx *ctx);
/* measurement helper (bench.py): microseconds per in-place all-reduce of `count` doubles through the ctx's communicator, `reps` of
 * them enqueued back to back on the ctx stream between two HIP events; a collective - every rank of the communicator calls it */
int wc_comm_allreduce_probe(wc_ctx *ctx, uint64_t count, int reps, double *h_us);
/* owner rank of a root voxel (VoxelLoc index, surfel_extraction.h:59-64): hash(kx,ky,kz) mod world; needs no GPU */
int wc_route_owner(int32_t kx, int32_t ky, int32_t kz, int world);
/* stable partition of this rank's points by owner: d_send (capacity pts->n records) receives `world` consecutive segments
 * (owner 0, 1, ...), time order preserved inside each; h_counts[world] = their lengths */
int wc_route_partition(wc_ctx *ctx, const wc_points *pts, int world, wc_route_point *d_send, uint64_t *h_counts);
/* the whole sharded call on one rank: partition the local time-contiguous slice, ONE all-to-all of 24-byte records through
 * the ctx's communicator, wc_extract_surfels on the points of the voxels this rank owns.  t_lo <= t_hi: the time range of
 * the WHOLE cloud.  Output: this rank's surfels (disjoint from the other ranks' by voxel), time sorted.
 * h_n_points_owned (may be NULL): points this rank received. */
int wc_extract_surfels_sharded(wc_ctx *ctx, const wc_points *pts, double t_lo, double t_hi, wc_surfel *d_out, wc_surfel_id *d_ids,
                               uint64_t cap, uint64_t *h_n_out, uint64_t *h_n_points_owned);
/* k time-sorted surfel lists, concatenated in d_in (h_counts[k] lengths) -> one list in the extraction's canonical order
 * (timestamp, ties by root voxel index and node id; without ids: timestamp, ties by list).  d_out must not alias d_in. */
int wc_merge_surfels(wc_ctx *ctx, const wc_surfel *d_in, const wc_surfel_id *d_in_ids, const uint64_t *h_counts, int k,
                     wc_surfel *d_out, wc_surfel_id *d_out_ids);
/* all-gather of every rank's surfel list + merge: every rank ends with the unsharded call's output (replicated window) */
int wc_gather_surfels(wc_ctx *ctx, const wc_surfel *d_local, const wc_surfel_id *d_local_ids, uint64_t n_local, wc_surfel *d_out,
                      wc_surfel_id *d_out_ids, uint64_t cap, uint64_t *h_n_out);

/* sweep preparation ("next" row f-1 of SURVEY.md §8: the per-point stages right in front of the hot path) ------------ */
/* Replaces the per-point loop of LidarOdometry::AddLidarScan (src/odometry/lidar_odometry.cc:489-496): lidar->imu
 * extrinsic in double (quat = w,x,y,z), cast to float, drop points with |p| < min_range, |p| > max_range or inside the
 * blind box; survivors keep their order.  Records are the 48-byte hilti_ros::Point (common.h:12-28). */
int wc_prefilter_points(wc_ctx *ctx, const void *d_pts_in, uint64_t n, const double ext_quat[4], const double ext_t[3],
                        double min_range, double max_range, const double blind_min[3], const double blind_max[3],
                        void *d_pts_out, uint64_t cap, uint64_t *h_n_out);
/* The same loop including its CHECK(points_buff_.empty() || pt.time >= points_buff_.back().time) (:491), evaluated on the device
 * for EVERY incoming point against the last point buffered at that moment: prev_time = stamp of the last buffered point before this
 * message (-INFINITY: buffer empty).  *h_monotonic = 0 when the CHECK would have fired.  d_kept_times (may be NULL, capacity
 * `cap`): the survivors' stamps, packed - the only thing a host-side window bookkeeping needs back. */
int wc_prefilter_points_checked(wc_ctx *ctx, const void *d_pts_in, uint64_t n, const double ext_quat[4], const double ext_t[3],
                                double min_range, double max_range, const double blind_min[3], const double blind_max[3],
                                void *d_pts_out, uint64_t cap, uint64_t *h_n_out, double prev_time, double *d_kept_times,
                                int *h_monotonic);
/* Replaces UndistortSweep(sweep_in, imu_states, sweep_out) (src/odometry/lidar_odometry.cc:143-158).
 * WC_ERR_RANGE mirrors the CHECK at :149. */
int wc_undistort_sweep(wc_ctx *ctx, const void *d_pts_in, uint64_t n, const wc_imu_state *d_imu, uint64_t n_imu, void *d_pts_out);
/* The same, but the sweep leaves as the 20 bytes per point BuildSurfels reads (surfel_extraction.cc:317-324 copies x, y, z, time
 * and nothing else): d_xyz_out = n x 3 floats, d_time_out = n doubles, i.e. the wc_points {d_xyz_out, d_time_out, 12, 8, n}.
 * The undistorted 48-byte records are never written or read: 48 + 20 + 20 bytes of traffic per point up to and including the
 * extraction's pass instead of 48 + 48 + 48. */
int wc_undistort_sweep_packed(wc_ctx *ctx, const void *d_pts_in, uint64_t n, const wc_imu_state *d_imu, uint64_t n_imu, float *d_xyz_out,
                              double *d_time_out);

/* surfel pose update --------------------------------------------------------------------------------------------- */
/* Replaces UpdateSurfelPoses(const std::deque<ImuState>&, std::deque<Surfel::Ptr>&) (src/odometry/lidar_odometry.cc:160-170)
 * + Surfel::UpdatePose (src/odometry/surfel.h:48-58).  d_in_body[i] == 0 marks a surfel still in the world frame
 * (fresh from wc_extract_surfels); it is converted to the body frame and the flag set.  WC_ERR_RANGE mirrors the
 * CHECK at lidar_odometry.cc:164. */
int wc_update_surfel_poses(wc_ctx *ctx, const wc_imu_state *d_imu, uint64_t n_imu, wc_surfel *d_surf, wc_pose *d_pose,
                           uint8_t *d_in_body, uint64_t n);

/* ShrinkToFit (src/odometry/lidar_odometry.cc:243-246) moves the oldest sliding-window surfels to the fixed window with
 * push_front, oldest first: the fixed window is kept NEWEST-first.  dst[j] = src[n-1-j]; asynchronous on the
 * ctx stream. */
int wc_reverse_copy_surfels(wc_ctx *ctx, const wc_surfel *d_src_surf, const wc_pose *d_src_pose, uint64_t n,
                            wc_surfel *d_dst_surf, wc_pose *d_dst_pose);

/* correspondence -------------------------------------------------------------------------------------------------- */
/* Replaces KnnSurfelMatcher::BuildIndex(const std::deque<Surfel::Ptr>&) + Match(std::deque<Surfel::Ptr>&,
 * std::vector<SurfelCorrespondence>&) (src/odometry/knn_surfel_matcher.h:17-19, .cc:3-49; calls lidar_odometry.cc:532-538).
 *   same_set != 0 : sliding-window matcher, the targets ARE the queries (pass the same arrays twice); pairs are
 *                   (older, newer) indices into that set, at most one per query, in query order
 *   same_set == 0 : fixed-window matcher; pair.first indexes the targets (fixed window), pair.second the queries
 *   d_knn_idx / d_knn_d2 (may be NULL): the raw exact k nearest neighbours per query (k = wc_params.knn_k), the output
 *                   of FLANNKNearestSearch (cc:75-89), for known-answer tests.  With them NULL a walk is bounded by the nearest
 *                   gate-passing candidate as well as by the k-th distance - Match takes the FIRST of the k neighbours that passes
 *                   the gates (cc:24-46), so the pair list is the same, byte for byte (development option knn_early = 0: off)
 * Surfels must be in the body frame with poses attached (wc_update_surfel_poses). */
int wc_match(wc_ctx *ctx, const wc_surfel *d_q_surf, const wc_pose *d_q_pose, uint64_t nq, const wc_surfel *d_t_surf,
             const wc_pose *d_t_pose, uint64_t nt, int same_set, wc_pair *d_pairs, uint64_t cap, uint64_t *h_n_pairs,
             uint32_t *d_knn_idx, double *d_knn_d2);

/* What the walk of the last wc_match on this context touched (the index is a 6-D kd-tree with bounding boxes, csrc/match_tree.inc),
 * from a sample of its wavefronts: h_out[0 .. 3] = wide nodes opened, leaves scanned, points given the fp32 first look, exact
 * fp64 distances - sums over h_out[4] sampled queries; h_out[5] = depth of the tree, h_out[6] = levels above the buckets,
 * h_out[7] = targets.  Measurement only (bench.py's candidates-per-query figure). */
int wc_match_stats(wc_ctx *ctx, double h_out[8]);

/* Multi-GPU form of wc_match (the queries are independent, knn_surfel_matcher.cc:22-48): a COLLECTIVE of the ctx's
 * communicator - every rank calls it with the same replicated arguments; each searches a contiguous share of the queries (in
 * tree-leaf order) and ONE all-gather of the gated neighbour lists (4 k bytes per query) gives every rank the whole table, on
 * which the order-dependent de-duplication (cc:35-38) runs replicated: every rank ends with the unsharded call's pairs, byte for
 * byte.  wc_match itself is never a collective, whatever is installed on the ctx.  Without a communicator (or a world of one) this
 * is wc_match.  A rank that fails locally BEFORE the all-gather (an allocation, a HIP error) returns its error without entering it:
 * treat a non-zero return of any rank as fatal for the job (wc_window_build_sharded, whose local part can fail on its arguments,
 * joins its share check with a poisoned share instead). */
int wc_match_sharded(wc_ctx *ctx, const wc_surfel *d_q_surf, const wc_pose *d_q_pose, uint64_t nq, const wc_surfel *d_t_surf,
                     const wc_pose *d_t_pose, uint64_t nt, int same_set, wc_pair *d_pairs, uint64_t cap, uint64_t *h_n_pairs);
/* both searches of an outer iteration as collectives (one after the other: their all-gathers share the ctx stream) */
int wc_match_pair_sharded(wc_ctx *ctx, const wc_surfel *d_sld_surf, const wc_pose *d_sld_pose, uint64_t n_sld,
                          const wc_surfel *d_fix_surf, const wc_pose *d_fix_pose, uint64_t n_fix, wc_pair *d_pairs_sld, uint64_t cap_sld,
                          uint64_t *h_n_pairs_sld, wc_pair *d_pairs_fix, uint64_t cap_fix, uint64_t *h_n_pairs_fix);

/* Both correspondence searches of one outer iteration (the two KnnSurfelMatcher objects of lidar_odometry.cc:530-538) at
 * once.  Same results, bit for bit, as
 *   wc_match(ctx, sld, sld_pose, n_sld, sld, sld_pose, n_sld, 1, d_pairs_sld, ...) followed by
 *   wc_match(ctx, sld, sld_pose, n_sld, fix, fix_pose, n_fix, 0, d_pairs_fix, ...),
 * but the fixed-window search runs on a helper context of its own (own stream and scratch, created on first use, ordered
 * behind the work already enqueued on the ctx stream): the builds of the two trees are chains of small launches and a search is a
 * single round of wavefronts of unequal length, so each fills the slots the other leaves.  Never a collective
 * (wc_match_pair_sharded is). */
int wc_match_pair(wc_ctx *ctx, const wc_surfel *d_sld_surf, const wc_pose *d_sld_pose, uint64_t n_sld,
                  const wc_surfel *d_fix_surf, const wc_pose *d_fix_pose, uint64_t n_fix, wc_pair *d_pairs_sld, uint64_t cap_sld,
                  uint64_t *h_n_pairs_sld, wc_pair *d_pairs_fix, uint64_t cap_fix, uint64_t *h_n_pairs_fix);

/* window problem: factors + Levenberg-Marquardt ----------------------------------------------------------------- */
/* Replaces the ceres::Problem construction of lidar_odometry.cc:541-545:
 *   BuildSldWinLidarResiduals (cc:254-297) — d_pairs_sld index the sliding-window surfels (older, newer),
 *   BuildFixWinLidarResiduals (cc:299-317) — d_pairs_fix: first = fixed-window surfel, second = sliding-window surfel,
 *   BuildImuResiduals         (cc:319-363) — h_imu: the window's IMU states (host; a few thousand records),
 * with SampleState timestamps h_sample_times[ns] (ascending), gravity of the last sample state and the
 * SubsetParameterization gauge flag (cc:556-560).  Surfels must already carry poses (wc_update_surfel_poses).
 * The packed per-correspondence records are built once here; the calls below reuse them.
 * 2 <= ns <= 340 (dense normal equations of 12 ns unknowns; the reference's default 6.5 s window has 82 sample states),
 * otherwise WC_ERR_ARG.
 * Ordering: every host argument (h_imu, h_sample_times, h_grav) has been consumed when the call returns, and argument errors the
 * device finds (a surfel stamp outside the sample-state range, a correspondence that is not (older, newer)) are reported by it;
 * the call's LAST device work - the records and one copy of host-built lists - may still be in flight on the ctx stream.  The
 * calls that use the problem (wc_window_solve / _linearize / _evaluate) are enqueued behind it on that stream; a caller that reads
 * the device arguments' buffers on ANOTHER stream orders itself with an event on the ctx stream, as after any asynchronous call. */
int wc_window_build(wc_ctx *ctx, const wc_surfel *d_sld_surf, const wc_pose *d_sld_pose, const wc_pair *d_pairs_sld,
                    uint64_t n_pairs_sld, const wc_surfel *d_fix_surf, const wc_pose *d_fix_pose, const wc_pair *d_pairs_fix,
                    uint64_t n_pairs_fix, const wc_imu_state *h_imu, uint64_t n_imu, const double *h_sample_times, uint64_t ns,
                    const double *h_grav, int fix_first_pos);
/* Multi-GPU form (correspondences and IMU factors sharded, unknowns replicated): a COLLECTIVE of the ctx's
 * communicator.  EVERY rank passes the SAME replicated arguments as it would to wc_window_build; the library takes this rank's
 * contiguous share of both correspondence lists - the IMU factors, a few hundred to two thousand, stay on EVERY rank -, and one small
 * all-reduce checks that the shares add up to the whole problem (WC_ERR_ARG otherwise).  From then on wc_window_linearize / _evaluate /
 * _solve on this problem are collectives, two per linearisation (round 6): {cost of the surfel factors, spare} - 16 bytes, all the
 * trust-region decision waits for - and {6 x 6 pose corner of every upper block pair, pose half of g} (surfel factors reach nothing
 * else: 0.60 MB at 64 sample states, 2.35 MB at 127), which the in-library RCCL binding runs on a side stream beside the bias
 * elimination; every rank takes identical accept / reject decisions on identical numbers.  (Development option lm_one_collective:
 * rounds 3 - 5's ONE all-reduce of {upper block pairs of H, g, cost} with the IMU triples sharded too: 0.76 / 2.7 MB.)
 * A problem built with wc_window_build is never a collective, whatever is installed on the ctx (a caller that shards the factors
 * itself opts in with wc_window_set_allreduce).  Without a communicator (or a world of one) this is wc_window_build. */
int wc_window_build_sharded(wc_ctx *ctx, const wc_surfel *d_sld_surf, const wc_pose *d_sld_pose, const wc_pair *d_pairs_sld,
                            uint64_t n_pairs_sld, const wc_surfel *d_fix_surf, const wc_pose *d_fix_pose, const wc_pair *d_pairs_fix,
                            uint64_t n_pairs_fix, const wc_imu_state *h_imu, uint64_t n_imu, const double *h_sample_times, uint64_t ns,
                            const double *h_grav, int fix_first_pos);
/* counts[4] = {binary factors, unary factors, imu factors, assembly pieces} (of THIS rank's share for a sharded problem);
 * wc_window_reduce_bytes: bytes one linearisation's collectives carry (0: the problem is not sharded).  A problem built by
 * wc_window_build_sharded has two per linearisation since round 6 - {cost of the surfel factors, spare} (16 bytes) and {6 x 6 pose corner of
 * every upper block pair, pose half of g} -, the IMU factors being replicated; the figure is their sum */
uint64_t wc_window_reduce_bytes(wc_ctx *ctx);
int wc_window_counts(wc_ctx *ctx, uint64_t counts[4]);
/* problem.Evaluate(apply_loss_function = true) (lidar_odometry.cc:62-65): cost = 1/2 sum rho; d_residuals (may be NULL)
 * receives the loss-corrected residuals in the reference's block order: binary, unary, 12 per IMU factor.
 * h_x = the 12*ns correction blocks (SampleState::data_cor, surfel.h:13-17). */
int wc_window_evaluate(wc_ctx *ctx, const double *h_x, double *h_cost, double *d_residuals);
/* one linearisation: dense row-major H = J^T J (12ns x 12ns) and g = J^T r, loss-corrected, gauge columns zeroed */
int wc_window_linearize(wc_ctx *ctx, const double *h_x, double *d_H, double *d_g, double *h_cost);
/* measurement: `reps` linearisations at h_x back to back on the ctx stream between two HIP events; *h_ms = device time of one
 * (the kernels and the gap between them - no upload, no wait for the mailbox, no caller between two of them) */
int wc_window_linearize_timed(wc_ctx *ctx, const double *h_x, int reps, float *h_ms_per_linearisation);
/* ceres::Solve with the reference's options (lidar_odometry.cc:551-561): trust-region LM, <= max_iterations.
 * h_x_inout: corrections in / optimised corrections out; h_first_step (may be NULL) receives the first LM increment. */
int wc_window_solve(wc_ctx *ctx, double *h_x_inout, wc_solve_summary *summary, double *h_first_step);
/* multi-GPU, for a caller that shards the factors ITSELF (each rank builds its own slices with wc_window_build): install a
 * "sum this device buffer over all ranks" callback; called once per linearisation on the packed {upper block pairs of H, g, cost}
 * buffer (layout: wc_window_build_sharded above) and once per candidate-cost evaluation (one double).  NULL removes it. */
int wc_window_set_allreduce(wc_ctx *ctx, int (*fn)(void *user, double *d_buf, uint64_t count), void *user);

#ifdef __cplusplus
}
#endif
#endif /* WILDCAT_HIP_H_ */
