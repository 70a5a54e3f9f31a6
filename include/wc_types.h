/*
 * wc_types.h — plain-old-data records that cross the C-ABI of the MI355X hot path.
 *
 * Every record is a flat run of doubles / ints so that it can live in HBM, in a numpy array, or
 * in a C++ host struct without translation.  Each one names the reference type it stands for
 * (paths relative to the reference checkout).
 */
#ifndef WC_TYPES_H_
#define WC_TYPES_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Input point buffer.  The reference hands `std::vector<hilti_ros::Point>` (src/common/common.h:12-28,
 * a 48-byte AoS record: float x,y,z,pad @0, float intensity @16, double time @24, uint16 ring @32) to
 * BuildSurfels (src/odometry/surfel_extraction.cc:316).  The descriptor accepts that record in place
 * (xyz = base, xyz_stride = 48, time = base + 24, time_stride = 48) or a packed SoA
 * (xyz_stride = 12 or 16, time_stride = 8).  Pointers are DEVICE pointers for the wc_* entry points and
 * host pointers for the oracle. */
typedef struct wc_points {
  const void *xyz;      /* first float of point 0 (x, y, z consecutive floats)            */
  const void *time;     /* first double (timestamp of point 0), ascending                  */
  uint32_t xyz_stride;  /* bytes between consecutive points' x                             */
  uint32_t time_stride; /* bytes between consecutive points' timestamps                    */
  uint64_t n;           /* number of points                                                */
} wc_points;

#define WC_HILTI_POINT_BYTES 48
#define WC_HILTI_POINT_TIME_OFFSET 24

/* One surfel as `BuildSurfels` emits it: reference `Surfel` ctor arguments
 * (src/odometry/surfel.h:39-40, src/odometry/surfel_extraction.cc:62).  144 bytes.
 * Right after extraction center/cov/normal are in the world frame; after the first pose update they are in
 * the body frame (surfel.h:48-58). */
typedef struct wc_surfel {
  double t;          /* mean timestamp of the cluster                       */
  double center[3];  /* mean point                                          */
  double cov[9];     /* population covariance, row-major 3x3 (symmetric)    */
  double normal[3];  /* eigenvector of the smallest eigenvalue, view-flipped */
  double resolution; /* 4 * quarter_length of the emitting octree node      */
  double sigma;      /* sqrt(lambda_min)                                    */
} wc_surfel;

/* Identity of the octree node + temporal cluster a surfel came from.  Not a reference type: the
 * reference's emission order is hash-map order, so parity is checked by matching surfels
 * on this id.  node = layer | o1 << 2 | o2 << 5 | cluster << 8, where o1/o2 are the octant codes
 * 4*[x>cx] + 2*[y>cy] + [z>cz] (surfel_extraction.cc:147-158) of the layer-1 / layer-2 node and
 * `cluster` is the ordinal of the temporal cluster inside the node (dropped clusters count). */
typedef struct wc_surfel_id {
  int32_t kx, ky, kz; /* root voxel index, floor(p / voxel_size)  (surfel_extraction.h:59-64) */
  uint32_t node;
} wc_surfel_id;

/* Body->world pose attached to a surfel (surfel.h:114-115). quat = (w, x, y, z). 56 bytes. */
typedef struct wc_pose {
  double pos[3];
  double quat[4];
} wc_pose;

/* Reference `ImuState` (src/odometry/surfel.h:25-33) flattened: 112 bytes. quat = (w, x, y, z). */
typedef struct wc_imu_state {
  double t;
  double pos[3];
  double quat[4];
  double acc[3];
  double gyr[3];
} wc_imu_state;

/* Index pair produced by the matcher: reference `SurfelCorrespondence{s1, s2}`
 * (src/odometry/surfel.h:124-127) with s1 the older surfel (knn_surfel_matcher.cc:41-45).
 * For the sliding-window matcher both are indices into the query/target set (the same set);
 * for the fixed-window matcher `first` indexes the fixed-window (target) set and `second` the query set. */
typedef struct wc_pair {
  int32_t first;
  int32_t second;
} wc_pair;

/* Record of the ONE exchange step of the sharded extraction: the 20 used bytes of a point
 * (float xyz + double time) padded to 24.  A received buffer is a valid wc_points input (xyz_stride = time_stride = 24). */
typedef struct wc_route_point {
  float x, y, z;
  uint32_t src; /* spare (keeps t 8-aligned) */
  double t;
} wc_route_point;

/* Communicator of a multi-GPU job: one process (and one wc_ctx) per GPU.  The library calls these for its few collectives;
 * wc_comm_rccl_init() installs an in-library RCCL implementation, tests / other runtimes install callbacks.
 * All buffers are DEVICE pointers on the ctx's GPU; a callback returns 0 on success and must have completed (or be
 * stream-ordered on the ctx's stream) when it returns.  Byte counts are per rank, data consecutive in rank order. */
typedef struct wc_comm {
  void *user;
  int32_t rank, world;
  int (*allreduce_f64)(void *user, double *d_buf, uint64_t count); /* in-place sum over ranks */
  int (*alltoallv)(void *user, const void *d_send, const uint64_t *send_bytes, void *d_recv, const uint64_t *recv_bytes);
  int (*allgatherv)(void *user, const void *d_send, uint64_t send_bytes, void *d_recv, const uint64_t *recv_bytes);
  int32_t stream_ordered; /* 1: the callbacks enqueue on the ctx's stream themselves (the in-library RCCL binding): the library
                             does not synchronise around them.  0: the library drains its stream before every call and the
                             callback must have completed when it returns */
} wc_comm;

/* One sweep of a batched extraction (wc_extract_surfels_batch_*): the arguments of wc_extract_surfels_enqueue. */
typedef struct wc_sweep_job {
  wc_points pts;
  double t_lo, t_hi;   /* time range hint of THIS sweep (t_lo > t_hi: read back from the device) */
  wc_surfel *d_out;
  wc_surfel_id *d_ids; /* may be NULL */
  uint64_t cap;
} wc_sweep_job;

/* Hard-coded reference parameters of the path.  wc_params_default() fills the
 * reference values; tests may override them to reach edge cases. */
typedef struct wc_params {
  /* extraction: src/odometry/surfel_extraction.cc:327 */
  float voxel_size;          /* 0.8f  (float in the reference signature)               */
  int32_t max_layer;         /* 2                                                     */
  int32_t min_points;        /* node is tested when n > min_points (20)               */
  float planer_threshold;    /* 0.01f                                                 */
  double min_plane_likeness; /* 0.1                                                   */
  double view_point[3];      /* (0,0,0)                                               */
  double cluster_gap;        /* 0.05 s  (surfel_extraction.cc:24)                     */
  int32_t cluster_min_points;/* clusters with fewer points are dropped (20, cc:33)    */
  /* matcher: src/odometry/knn_surfel_matcher.h:37-41 */
  double center_scale;       /* 1.0                                                   */
  double angular_scale;      /* 5 deg in rad                                          */
  double surfel_dist_max;    /* 0.1                                                   */
  int32_t knn_k;             /* 10                                                    */
  double time_diff_min;      /* 0.06                                                  */
  /* factors / solver: src/odometry/lio_config.h:10-14,32-45, lidar_odometry.cc:270,551-560 */
  double surfel_sigma0;      /* 0.05 / 6 (cost_functor.h:24)                          */
  double cauchy_a;           /* 0.4                                                   */
  double w_gyr, w_acc, w_bg, w_ba; /* IMU cost weights                                */
  double imu_dt;             /* 1 / imu_rate = 0.005                                  */
  int32_t max_iterations;    /* 100                                                   */
  int32_t reference_quirks;  /* 1: reproduce Q1 (Jacobian overwrite) and Q3           */
  /* extraction arithmetic (not a reference parameter).  0 (default): order-independent integer moments - counts / ids exact,
   * geometry ~1e-9, surfel time stamp = the correctly rounded mean, sweeps with a gate inside the reference's own rounding
   * noise are repeated on the exact path.  1: every sum formed in the reference's order (bit-identical to the CPU path,
   * including the output order among surfels whose stamps differ by less than the running sum's rounding). */
  int32_t exact_sums;
} wc_params;

/* Solver summary (subset of ceres::Solver::Summary the reference logs, lidar_odometry.cc:562). */
typedef struct wc_solve_summary {
  double initial_cost;
  double final_cost;
  int32_t iterations;            /* LM iterations performed (iteration 0 excluded)   */
  int32_t successful_steps;
  int32_t unsuccessful_steps;
  int32_t termination;           /* 0 = convergence, 1 = no convergence (max iters), 2 = failure */
  int32_t n_linearizations;
  int32_t n_cost_evaluations;
  double first_step[16];         /* [0] first_step_norm; [1] steps re-formed by the dense factorisation (a rejected or
                                  * invalid step of the bias elimination, wildcat_hip.h: wc_window_solve); rest unused */
} wc_solve_summary;

#ifdef __cplusplus
}
#endif
#endif /* WC_TYPES_H_ */
