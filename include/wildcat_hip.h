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

/* device memory helpers so that a host program needs no HIP headers ---------------------------------------------- */
int wc_dev_alloc(wc_ctx *ctx, size_t bytes, void **d_ptr);
int wc_dev_free(wc_ctx *ctx, void *d_ptr);
int wc_h2d(wc_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int wc_d2h(wc_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int wc_memset(wc_ctx *ctx, void *d_dst, int value, size_t bytes);
int wc_sync(wc_ctx *ctx);
/* HIP-event timing on the ctx stream (used by bench.py: torch.cuda.Event only sees torch's stream) */
int wc_timer_start(wc_ctx *ctx);
int wc_timer_stop_ms(wc_ctx *ctx, float *h_ms);

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
/* per-stage device time of the LAST enqueued extraction, measured with HIP events on the ctx stream:
 * h_ms5 = {key generation, point radix sort, k_roots (octree + PCA), surfel-slot sort, gather}.  Enable first. */
int wc_extract_profile(wc_ctx *ctx, int enable);
int wc_extract_stage_ms(wc_ctx *ctx, float *h_ms5);
/* root-voxel index of every point: VoxelLoc (src/odometry/surfel_extraction.h:55-64); d_keys_xyz = 3 int32 per point */
int wc_voxel_keys(wc_ctx *ctx, const wc_points *pts, int32_t *d_keys_xyz);

#ifdef __cplusplus
}
#endif
#endif /* WILDCAT_HIP_H_ */
