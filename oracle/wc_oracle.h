/*
 * oracle/wc_oracle.h — TEST INFRASTRUCTURE.  CPU restatement ("oracle") of the reference hot path
 * (src/odometry of kekeliu-whu/Wildcat-SLAM, call site lidar_odometry.cc:523-566).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * shipped product path (libwildcat_hip.so) never links, loads or calls it.
 *
 * PARITY STATUS: the reference cannot be compiled in this image (every hot-path translation unit needs
 * Eigen + glog + Ceres/PCL/abseil, none installed, no network) and its own tests hold no vectors for
 * extraction, matching, factors or the solve.  What IS pinned against the reference's tests:
 *   - SO(3) Jacobians        : src/common/utils_test.cc:5-21         (tests/test_oracle_kat.py)
 *   - exact k-NN, k = 10     : src/odometry/knn_surfel_matcher_test.cc:19-43
 *   - cubic B-spline fit     : src/odometry/spline_interpolation_test.cc:79-96 (+ closed forms :10-48)
 * Extraction, factor and LM-solve parity are "parity unpinned": restated line-by-line with citations and
 * cross-checked by an independent numpy/scipy restatement (oracle/np_check.py), nothing more.
 *
 * All pointers are HOST pointers.
 */
#ifndef WC_ORACLE_H_
#define WC_ORACLE_H_

#include "../include/wc_types.h"

#ifdef __cplusplus
extern "C" {
#endif

void wco_params_default(wc_params *p);

/* ---- math known-answer hooks (utils.h:15-67, so3.hpp) ---- */
void wco_so3_exp(const double w[3], double quat_wxyz[4]);
void wco_so3_log(const double quat_wxyz[4], double w[3]);
void wco_so3_jl(const double v[3], double out9[9]);
void wco_so3_jl_inv(const double v[3], double out9[9]);
void wco_so3_jr(const double v[3], double out9[9]);
void wco_so3_jr_inv(const double v[3], double out9[9]);
void wco_eig3(const double a9[9], double evals[3], double evecs9[9]); /* columns = eigenvectors */

/* ---- extraction (surfel_extraction.cc) ---- */
typedef struct wco_extract_stats {
  uint64_t root_voxels;
  uint64_t nodes_tested[4]; /* per layer */
  uint64_t nodes_plane[4];
  uint64_t clusters_total;
  uint64_t clusters_rejected;
  uint64_t surfels;
  double min_gate_margin; /* smallest |quantity - threshold| over every gate evaluated */
} wco_extract_stats;

int wco_voxel_keys(const wc_points *pts, const wc_params *P, int32_t *keys_xyz);
int wco_extract_surfels(const wc_points *pts, const wc_params *P, wc_surfel *out, wc_surfel_id *out_ids, uint64_t cap,
                        uint64_t *n_out, wco_extract_stats *stats);

/* ---- surfel pose update (lidar_odometry.cc:160-170, surfel.h:48-58) ---- */
int wco_update_surfel_poses(const wc_imu_state *imu, uint64_t n_imu, wc_surfel *surf, wc_pose *pose, uint8_t *in_body,
                            uint64_t n);

/* ---- "next" row f-1: point pre-filter (lidar_odometry.cc:489-496) + UndistortSweep (lidar_odometry.cc:143-158) ----
 * pts_*: arrays of the 48-byte hilti_ros::Point record */
int wco_prefilter_points(const void *pts_in, uint64_t n, const double ext_quat[4], const double ext_t[3], double min_range,
                         double max_range, const double blind_min[3], const double blind_max[3], void *pts_out, uint64_t *n_out);
int wco_undistort_sweep(const void *pts_in, uint64_t n, const wc_imu_state *imu, uint64_t n_imu, void *pts_out);

/* ---- correspondence (knn_surfel_matcher.cc) ---- */
/* exact k nearest neighbours in the raw 6-D feature space (FLANNKNearestSearch, cc:75-89) */
int wco_knn6(const double *cloud6, uint64_t n, const double *query6, uint64_t nq, int k, int32_t *idx, double *dist2);
/* Match (cc:16-49). same_set != 0: targets are the query set itself (sliding-window matcher);
 * pairs are (older, newer) index pairs, see wc_pair. */
int wco_match(const wc_params *P, const wc_surfel *q_surf, const wc_pose *q_pose, uint64_t nq, const wc_surfel *t_surf,
              const wc_pose *t_pose, uint64_t nt, int same_set, wc_pair *pairs, uint64_t cap, uint64_t *n_pairs);

/* ---- window problem: factors + LM (cost_functor.h, lidar_odometry.cc:254-363,541-562) ---- */
typedef struct wco_window wco_window;

/* sample_times[ns]; x is 12*ns (data_cor blocks, surfel.h:13-17). grav[3] from the last sample state
 * (lidar_odometry.cc:341,355).  fix_first_pos: SubsetParameterization(12,{3,4,5}) on block 0 (cc:556-560). */
wco_window *wco_window_create(const wc_params *P, const double *sample_times, uint64_t ns, const double grav[3],
                              int fix_first_pos);
void wco_window_destroy(wco_window *w);
/* BuildSldWinLidarResiduals (cc:254-297): pairs index sld surfels */
int wco_window_add_binary(wco_window *w, const wc_surfel *surf, const wc_pose *pose, const wc_pair *pairs, uint64_t n);
/* BuildFixWinLidarResiduals (cc:299-317): pair.first indexes fix_*, pair.second indexes sld_* */
int wco_window_add_unary(wco_window *w, const wc_surfel *fix_surf, const wc_pose *fix_pose, const wc_surfel *sld_surf,
                         const wc_pose *sld_pose, const wc_pair *pairs, uint64_t n);
/* BuildImuResiduals (cc:319-363) */
int wco_window_add_imu(wco_window *w, const wc_imu_state *imu, uint64_t n_imu);
uint64_t wco_window_num_residuals(const wco_window *w);
void wco_window_counts(const wco_window *w, uint64_t counts[6]); /* bin mode0,1,2, unary, imu mode0, mode1 */
/* problem.Evaluate(apply_loss_function = true) (cc:62-65): cost = 1/2 sum rho; residuals loss-corrected */
int wco_window_evaluate(const wco_window *w, const double *x, double *cost, double *residuals_or_null);
/* one linearisation: dense H = J^T J (row-major 12ns x 12ns), g = J^T r, loss-corrected, no Jacobi scaling,
 * gauge columns (if any) zeroed */
int wco_window_linearize(const wco_window *w, const double *x, double *H, double *g, double *cost);
/* ceres::Solve with the reference's options (cc:551-561); Ceres-default trust-region LM restated */
int wco_window_solve(const wco_window *w, double *x_inout, wc_solve_summary *summary, double *first_step_or_null);

/* ---- post-solve (lidar_odometry.cc:22-54,172-215; spline_interpolation.h:42-113) ---- */
int wco_bspline_fit_eval(const double *timestamps, const double *points3, uint64_t np, const double *query_t,
                         uint64_t nq, double *out3, uint8_t *valid);
int wco_update_imu_poses(const double *sample_times, const double *x, uint64_t ns, const double ba[3],
                         const double bg[3], const double grav[3], wc_imu_state *imu, uint64_t n_imu);

/* ---- the orchestrated odometry (lidar_odometry.cc:365-611): see oracle/odometry.cc ---- */
typedef struct wco_odom wco_odom;
wco_odom *wco_odom_create(void);
void wco_odom_destroy(wco_odom *o);
int wco_odom_error(const wco_odom *o); /* non-zero: a CHECK of the reference would have aborted */
void wco_odom_add_imu(wco_odom *o, double t, const double acc[3], const double gyr[3]);  /* AddImuData cc:607-611 */
void wco_odom_add_scan(wco_odom *o, const void *points48, uint64_t n);                  /* AddLidarScan cc:487-605 */
void wco_odom_set_quirks(wco_odom *o, int on); /* 0: Q1/Q3 Jacobians accumulated, fixed window trimmed (not a reference mode) */
void wco_odom_export_state(const wco_odom *o, double *samples23, uint64_t cap_s, wc_imu_state *imu, uint64_t cap_i,
                           uint64_t counts[2]);
int wco_odom_sweeps(const wco_odom *o);
uint64_t wco_odom_num_samples(const wco_odom *o);
int wco_odom_sample(const wco_odom *o, uint64_t i, double *out15);
void wco_odom_stats(const wco_odom *o, double *stats10);
uint64_t wco_odom_window_times(const wco_odom *o, int fixed, double *out, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
