// oracle/odometry.cc — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// Orchestrated restatement of the reference's LidarOdometry (src/odometry/lidar_odometry.cc) on top of the oracle's
// stage functions, so that the host facade (wildcat-slam_amd/host/lidar_odometry.cc) can be held to it sweep by sweep:
//   AddImuData                        cc:607-611
//   AddLidarScan                      cc:487-605   (ROS publishing :582-602 left out)
//   SyncHeadingMsgs                   cc:457-485
//   PredictImuStatesAndSampleStates   cc:365-455   + PredictPoseOfNewImuState cc:106-123
//   BuildSweep                        cc:134-141
//   UpdateSamplePoses                 cc:172-179
//   ShrinkToFit                       cc:228-250   incl. Q11 (fixed window: push_front, never trimmed)
// Configuration values: src/odometry/lio_config.h:8-46.  The stage functions (pre-filter, undistortion, BuildSurfels,
// UpdateSurfelPoses, KnnSurfelMatcher, the Ceres problem + solve, UpdateImuPoses) are the oracle's own (wc_oracle.h).
// PARITY: unpinned like the stages themselves (the reference has no test or fixture for AddLidarScan).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <vector>

#include "math3.h"
#include "wc_oracle.h"

using namespace wco;

namespace {

struct Pt48 {  // hilti_ros::Point, src/common/common.h:12-28
  float x, y, z, pad;
  float intensity, pad1;
  double time;
  uint16_t ring;
  uint8_t tail[14];
};
static_assert(sizeof(Pt48) == 48, "48-byte point record");

struct ImuMsg {  // ImuData, common.h:31-35
  double t;
  double acc[3], gyr[3];
};

struct Sample {  // SampleState, surfel.h:9-23
  double timestamp;
  double cor[12];  // rot_cor, pos_cor, bg, ba
  double grav[3];
  double quat[4];
  double pos[3];
};

struct Surf {
  wc_surfel s;
  wc_pose p;
  uint8_t in_body;
};

inline V3 v3(const double *p) { return V3{p[0], p[1], p[2]}; }
inline Q4 q4(const double *p) { return Q4{p[0], p[1], p[2], p[3]}; }
inline void st3(double *d, V3 v) { d[0] = v.x, d[1] = v.y, d[2] = v.z; }
inline void stq(double *d, Q4 q) { d[0] = q.w, d[1] = q.x, d[2] = q.y, d[3] = q.z; }

// Eigen::Quaterniond(Matrix3d) (upstream Eigen, not in the reference tree): Shepperd's branches, no normalisation
Q4 quat_from_matrix_eigen(const double m[9]) {
  auto M = [&](int r, int c) { return m[3 * r + c]; };
  double q[4];  // x, y, z, w
  double t = M(0, 0) + M(1, 1) + M(2, 2);
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (M(2, 1) - M(1, 2)) * t;
    q[1] = (M(0, 2) - M(2, 0)) * t;
    q[2] = (M(1, 0) - M(0, 1)) * t;
  } else {
    int i = 0;
    if (M(1, 1) > M(0, 0)) i = 1;
    if (M(2, 2) > M(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (M(k, j) - M(j, k)) * t;
    q[j] = (M(j, i) + M(i, j)) * t;
    q[k] = (M(k, i) + M(i, k)) * t;
  }
  return Q4{q[3], q[0], q[1], q[2]};
}

}  // namespace

struct wco_odom {
  // lio_config.h:17-41
  double max_range = 120, min_range = 0.3;
  double blind_min[3] = {-0.8, -0.5, -0.4}, blind_max[3] = {0.3, 0.5, 0.4};
  double ext_t[3] = {-0.001, -0.00855, 0.055};
  double ext_q[4];
  double imu_rate = 200, sample_dt = 0.08, fixed_window_duration = 20.0, sliding_window_duration = 6.0, sweep_duration = 0.5;
  double gravity_norm = 9.81;
  int outer_iter_num_max = 1, inner_iter_num_max = 100;
  wc_params P;

  std::deque<ImuMsg> imu_buff;
  std::deque<Pt48> points_buff;
  std::deque<Sample> samples;
  std::deque<wc_imu_state> imu_states;
  std::deque<Surf> sld, fix;
  bool init_sld_win = false, sync_done = false, first_known = false;
  double first_sample_time = 0;
  int sweep_id = 0;
  int error = 0;  // a CHECK of the reference would have fired (code = source line of the restatement)
  wc_solve_summary last_summary{};
  uint64_t last_corr[2] = {0, 0};
  std::vector<double> pair_stamps[2];  // (first, second) surfel stamps of the last sweep's correspondences (test read-out)
  uint64_t last_new_surfels = 0;
};

#define ORACLE_CHECK(o, cond)                 \
  do {                                        \
    if (!(cond)) {                            \
      (o)->error = __LINE__;                  \
      return;                                 \
    }                                         \
  } while (0)

namespace {

// cc:106-123
void predict_pose(const wc_imu_state &i1, const wc_imu_state &i2, V3 ba, V3 bg, V3 grav, wc_imu_state &i3) {
  const double dt = i3.t - i2.t;
  stq(i3.quat, qmul(q4(i2.quat), so3_exp(((v3(i2.gyr) + v3(i3.gyr)) / 2 - bg) * dt)));
  st3(i3.pos, ((qrot(q4(i1.quat), v3(i1.acc) - ba) + grav) * dt) * dt + 2 * v3(i2.pos) - v3(i1.pos));
}

// cc:457-485
bool sync_heading(wco_odom *o) {
  if (o->sync_done) return true;
  if (o->imu_buff.empty() || o->points_buff.empty()) return false;
  if (o->imu_buff.back().t < o->points_buff.front().time) return false;
  while (o->imu_buff.front().t < o->points_buff.front().time) {
    o->imu_buff.pop_front();
    if (o->imu_buff.empty()) {
      o->error = __LINE__;
      return false;
    }
  }
  while (o->points_buff.front().time < o->imu_buff.front().t) {
    o->points_buff.pop_front();
    if (o->points_buff.empty()) {
      o->error = __LINE__;
      return false;
    }
  }
  o->sync_done = true;
  return true;
}

// cc:365-455
void predict_states(wco_odom *o, double end_time) {
  ORACLE_CHECK(o, o->imu_buff.size() >= 2);
  const double dt = 1 / o->imu_rate;
  if (!o->init_sld_win) {
    for (int i = 0; i < 2; ++i) {
      const ImuMsg m = o->imu_buff.front();
      o->imu_buff.pop_front();
      wc_imu_state s{};
      s.t = m.t;
      for (int d = 0; d < 3; ++d) s.acc[d] = m.acc[d], s.gyr[d] = m.gyr[d], s.pos[d] = 0;
      if (i == 0)
        stq(s.quat, Q4{1, 0, 0, 0});
      else
        stq(s.quat, so3_exp((v3(o->imu_states.back().gyr) + v3(s.gyr)) / 2 * dt));
      o->imu_states.push_back(s);
    }
    Sample ss{};
    ss.timestamp = o->imu_states.front().t;
    const V3 a0 = v3(o->imu_states.front().acc);
    st3(ss.grav, (-o->gravity_norm) * (a0 / norm(a0)));
    std::memcpy(ss.quat, o->imu_states.front().quat, 32);
    std::memcpy(ss.pos, o->imu_states.front().pos, 24);
    o->samples.push_back(ss);
    o->first_sample_time = ss.timestamp;
    o->first_known = true;
    o->init_sld_win = true;
  }
  const double old_last = o->samples.back().timestamp;
  const int add_size = (int)((end_time - old_last) / o->sample_dt);
  const double add_last = old_last + o->sample_dt * add_size;
  const V3 ba = v3(o->samples.back().cor + 9), bg = v3(o->samples.back().cor + 6), grav = v3(o->samples.back().grav);
  while (!o->imu_buff.empty()) {
    const size_t size = o->imu_states.size();
    const ImuMsg m = o->imu_buff.front();
    o->imu_buff.pop_front();
    wc_imu_state s{};
    s.t = m.t;
    for (int d = 0; d < 3; ++d) s.acc[d] = m.acc[d], s.gyr[d] = m.gyr[d];
    // CHECK_NEAR(i3.t - i2.t, i2.t - i1.t, 1e-6)  cc:119
    ORACLE_CHECK(o, std::fabs((s.t - o->imu_states[size - 1].t) - (o->imu_states[size - 1].t - o->imu_states[size - 2].t)) <= 1e-6);
    predict_pose(o->imu_states[size - 2], o->imu_states[size - 1], ba, bg, grav, s);
    o->imu_states.push_back(s);
    if (s.t >= add_last) break;
  }
  for (int i = 1; i <= add_size; ++i) {
    const double t = old_last + i * o->sample_dt;
    Sample ss{};
    ss.timestamp = t;
    st3(ss.cor + 9, ba), st3(ss.cor + 6, bg), st3(ss.grav, grav);
    size_t idx = 0;
    while (idx < o->imu_states.size() && o->imu_states[idx].t < t) ++idx;  // std::lower_bound
    ORACLE_CHECK(o, idx != 0 && idx != o->imu_states.size());
    const wc_imu_state &a = o->imu_states[idx - 1], &b = o->imu_states[idx];
    const double f = (t - a.t) / (b.t - a.t);
    stq(ss.quat, qslerp(q4(a.quat), f, q4(b.quat)));
    st3(ss.pos, (1 - f) * v3(a.pos) + f * v3(b.pos));
    ORACLE_CHECK(o, f >= 0 && f <= 1);
    o->samples.push_back(ss);
  }
}

// UpdateSurfelPoses over the sliding window (cc:160-170)
void update_surfel_poses(wco_odom *o) {
  std::vector<wc_imu_state> imu(o->imu_states.begin(), o->imu_states.end());
  for (Surf &s : o->sld) {
    const int rc = wco_update_surfel_poses(imu.data(), imu.size(), &s.s, &s.p, &s.in_body, 1);
    ORACLE_CHECK(o, rc == 0);
  }
}

// cc:228-250
void shrink_to_fit(wco_odom *o) {
  if (o->samples.empty() || o->samples.back().timestamp - o->samples.front().timestamp <= o->sliding_window_duration) return;
  while (o->samples.back().timestamp - o->samples.front().timestamp > o->sliding_window_duration) o->samples.pop_front();
  while (o->imu_states.front().t < o->samples.front().timestamp) o->imu_states.pop_front();
  while (!o->sld.empty() && o->sld.front().s.t < o->imu_states.front().t) {
    o->fix.push_front(o->sld.front());  // oldest first, each to the FRONT: the fixed window ends up newest-first (Q11)
    o->sld.pop_front();
  }
  // cc:247-249 compares back() with itself: the fixed window is never trimmed (Q11).  With the quirks switched off (not a
  // reference mode: the intended behaviour the facade's reference_quirks = false implements) it is cut from its old end.
  if (!o->P.reference_quirks)
    while (!o->fix.empty() && o->fix.front().s.t - o->fix.back().s.t > o->fixed_window_duration) o->fix.pop_back();
}

}  // namespace

extern "C" wco_odom *wco_odom_create(void) {
  wco_odom *o = new wco_odom;
  wco_params_default(&o->P);
  o->P.max_iterations = o->inner_iter_num_max;
  const double rot[9] = {-5.32125e-08, -1, 0, -1, -5.32125e-08, -0, 0, 0, -1};  // lio_config.h:25-28
  stq(o->ext_q, quat_from_matrix_eigen(rot));
  return o;
}
extern "C" void wco_odom_destroy(wco_odom *o) { delete o; }
extern "C" int wco_odom_error(const wco_odom *o) { return o->error; }

extern "C" void wco_odom_add_imu(wco_odom *o, double t, const double acc[3], const double gyr[3]) {
  ImuMsg m;
  m.t = t;
  for (int d = 0; d < 3; ++d) m.acc[d] = acc[d], m.gyr[d] = gyr[d];
  o->imu_buff.push_back(m);
}

extern "C" void wco_odom_add_scan(wco_odom *o, const void *points48, uint64_t n) {
  if (o->error) return;
  // lidar -> imu frame, range / blind-box filter (cc:489-496)
  std::vector<Pt48> kept(n);
  uint64_t m = 0;
  wco_prefilter_points(points48, n, o->ext_q, o->ext_t, o->min_range, o->max_range, o->blind_min, o->blind_max, kept.data(), &m);
  {
    // the CHECK at cc:491 runs on every incoming point, filtered or not
    const Pt48 *in = (const Pt48 *)points48;
    double prev = o->points_buff.empty() ? -INFINITY : o->points_buff.back().time;
    uint64_t k = 0;
    for (uint64_t i = 0; i < n; ++i) {
      ORACLE_CHECK(o, in[i].time >= prev);  // against the last BUFFERED point
      // the pre-filter copies a surviving record and only rewrites xyz: bytes 16.. identify it
      if (k < m && std::memcmp((const char *)&kept[k] + 16, (const char *)&in[i] + 16, 32) == 0) {
        prev = in[i].time;
        ++k;
      }
    }
  }
  for (uint64_t i = 0; i < m; ++i) o->points_buff.push_back(kept[i]);
  if (!sync_heading(o)) return;

  // 1. collect scan to sweep (cc:501-509)
  double sweep_endtime = o->points_buff.front().time + o->sweep_duration;
  if (o->points_buff.back().time < sweep_endtime || o->imu_buff.empty() || o->imu_buff.back().t < sweep_endtime) return;

  // 2. integrate IMU poses in windows (cc:512-513)
  predict_states(o, sweep_endtime);
  if (o->error) return;
  sweep_endtime = o->samples.back().timestamp;
  std::vector<Pt48> sweep;  // BuildSweep cc:134-141
  while (!o->points_buff.empty() && o->points_buff.front().time < sweep_endtime) {
    sweep.push_back(o->points_buff.front());
    o->points_buff.pop_front();
  }
  ORACLE_CHECK(o, !sweep.empty());

  // 3. undistort (cc:519-520)
  std::vector<wc_imu_state> imu(o->imu_states.begin(), o->imu_states.end());
  std::vector<Pt48> und(sweep.size());
  ORACLE_CHECK(o, wco_undistort_sweep(sweep.data(), sweep.size(), imu.data(), imu.size(), und.data()) == 0);

  // 4. BuildSurfels + UpdateSurfelPoses (cc:523-527)
  {
    const uint64_t cap = (3 * und.size()) / 20 + 1;
    std::vector<wc_surfel> out(cap);
    uint64_t n_new = 0;
    wc_points d{und.data(), (const char *)und.data() + 24, 48, 48, und.size()};
    ORACLE_CHECK(o, wco_extract_surfels(&d, &o->P, out.data(), nullptr, cap, &n_new, nullptr) == 0);
    o->last_new_surfels = n_new;
    for (uint64_t i = 0; i < n_new; ++i) {
      Surf s{};
      s.s = out[i];
      o->sld.push_back(s);
    }
  }
  update_surfel_poses(o);
  if (o->error) return;

  for (int iter = 0; iter < o->outer_iter_num_max; ++iter) {
    // correspondences (cc:530-538)
    std::vector<wc_surfel> ss, fs;
    std::vector<wc_pose> sp, fp;
    for (const Surf &s : o->sld) ss.push_back(s.s), sp.push_back(s.p);
    for (const Surf &s : o->fix) fs.push_back(s.s), fp.push_back(s.p);
    std::vector<wc_pair> pb(ss.size() + 1), pu(ss.size() + 1);
    uint64_t nb = 0, nu = 0;
    ORACLE_CHECK(o, wco_match(&o->P, ss.data(), sp.data(), ss.size(), ss.data(), sp.data(), ss.size(), 1, pb.data(), pb.size(), &nb) == 0);
    if (!fs.empty())
      ORACLE_CHECK(o, wco_match(&o->P, ss.data(), sp.data(), ss.size(), fs.data(), fp.data(), fs.size(), 0, pu.data(), pu.size(), &nu) == 0);
    o->last_corr[0] = nb, o->last_corr[1] = nu;
    for (int which = 0; which < 2; ++which) {
      o->pair_stamps[which].clear();
      const std::vector<wc_pair> &pr = which ? pu : pb;
      for (uint64_t i = 0; i < (which ? nu : nb); ++i) {
        o->pair_stamps[which].push_back(which ? fs[(size_t)pr[i].first].t : ss[(size_t)pr[i].first].t);
        o->pair_stamps[which].push_back(ss[(size_t)pr[i].second].t);
      }
    }
    // 5. the problem + solve (cc:541-562)
    std::vector<double> ts, x;
    for (const Sample &s : o->samples) {
      ts.push_back(s.timestamp);
      x.insert(x.end(), s.cor, s.cor + 12);
    }
    const bool fix_first = o->first_known && o->samples.front().timestamp == o->first_sample_time;  // cc:556-560
    wco_window *w = wco_window_create(&o->P, ts.data(), ts.size(), o->samples.back().grav, fix_first ? 1 : 0);
    int rc = wco_window_add_binary(w, ss.data(), sp.data(), pb.data(), nb);
    if (!rc && nu) rc = wco_window_add_unary(w, fs.data(), fp.data(), ss.data(), sp.data(), pu.data(), nu);
    imu.assign(o->imu_states.begin(), o->imu_states.end());
    if (!rc) rc = wco_window_add_imu(w, imu.data(), imu.size());
    if (!rc) rc = wco_window_solve(w, x.data(), &o->last_summary, nullptr);
    wco_window_destroy(w);
    ORACLE_CHECK(o, rc == 0);
    for (size_t i = 0; i < o->samples.size(); ++i) std::memcpy(o->samples[i].cor, &x[12 * i], 96);
    // state update (cc:564-566)
    const Sample &b = o->samples.back();
    ORACLE_CHECK(o, wco_update_imu_poses(ts.data(), x.data(), ts.size(), b.cor + 9, b.cor + 6, b.grav, imu.data(), imu.size()) == 0);
    for (size_t i = 0; i < imu.size(); ++i) o->imu_states[i] = imu[i];
    update_surfel_poses(o);
    if (o->error) return;
    for (Sample &s : o->samples) {  // UpdateSamplePoses cc:172-179
      stq(s.quat, qmul(so3_exp(v3(s.cor)), q4(s.quat)));
      st3(s.pos, v3(s.cor + 3) + v3(s.pos));
      for (int d = 0; d < 6; ++d) s.cor[d] = 0;
    }
  }
  shrink_to_fit(o);  // cc:574-580
  ++o->sweep_id;
}

extern "C" void wco_odom_set_quirks(wco_odom *o, int on) { o->P.reference_quirks = on ? 1 : 0; }
// the window's states for a re-synchronised comparison: 23 doubles per sample state (timestamp, cor[12], grav[3], quat[4],
// pos[3]) and the IMU states; counts[2] = their numbers (nothing is written beyond the capacities)
extern "C" void wco_odom_export_state(const wco_odom *o, double *samples23, uint64_t cap_s, wc_imu_state *imu, uint64_t cap_i,
                                      uint64_t counts[2]) {
  counts[0] = o->samples.size(), counts[1] = o->imu_states.size();
  for (uint64_t i = 0; i < o->samples.size() && i < cap_s; ++i) {
    const Sample &s = o->samples[i];
    double *p = samples23 + 23 * i;
    p[0] = s.timestamp;
    std::memcpy(p + 1, s.cor, 96);
    std::memcpy(p + 13, s.grav, 24);
    std::memcpy(p + 16, s.quat, 32);
    std::memcpy(p + 20, s.pos, 24);
  }
  for (uint64_t i = 0; i < o->imu_states.size() && i < cap_i; ++i) imu[i] = o->imu_states[i];
}
extern "C" int wco_odom_sweeps(const wco_odom *o) { return o->sweep_id; }
extern "C" uint64_t wco_odom_num_samples(const wco_odom *o) { return o->samples.size(); }
// out[15] = t, pos[3], quat[4] (w,x,y,z), bg[3], ba[3], spare
extern "C" int wco_odom_sample(const wco_odom *o, uint64_t i, double *out) {
  if (i >= o->samples.size()) return 1;
  const Sample &s = o->samples[i];
  out[0] = s.timestamp;
  std::memcpy(out + 1, s.pos, 24);
  std::memcpy(out + 4, s.quat, 32);
  std::memcpy(out + 8, s.cor + 6, 24);
  std::memcpy(out + 11, s.cor + 9, 24);
  out[14] = 0;
  return 0;
}
// stats[10] = sliding surfels, fixed surfels, binary corr, unary corr, LM iterations, initial cost, final cost, termination,
//             surfels of the last sweep, imu states
extern "C" void wco_odom_stats(const wco_odom *o, double *stats) {
  stats[0] = (double)o->sld.size();
  stats[1] = (double)o->fix.size();
  stats[2] = (double)o->last_corr[0];
  stats[3] = (double)o->last_corr[1];
  stats[4] = o->last_summary.iterations;
  stats[5] = o->last_summary.initial_cost;
  stats[6] = o->last_summary.final_cost;
  stats[7] = o->last_summary.termination;
  stats[8] = (double)o->last_new_surfels;
  stats[9] = (double)o->imu_states.size();
}
extern "C" uint64_t wco_odom_pair_stamps(const wco_odom *o, int which, double *out, uint64_t cap) {
  const std::vector<double> &v = o->pair_stamps[which ? 1 : 0];
  for (uint64_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
  return v.size();
}
// timestamps of the fixed window in its deque order (newest first, Q11) and of the sliding window
extern "C" uint64_t wco_odom_window_times(const wco_odom *o, int fixed, double *out, uint64_t cap) {
  const std::deque<Surf> &d = fixed ? o->fix : o->sld;
  for (uint64_t i = 0; i < d.size() && i < cap; ++i) out[i] = d[i].s.t;
  return d.size();
}
