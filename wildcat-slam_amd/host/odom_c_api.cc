// odom_c_api.cc — a flat C wrapper around the LidarOdometry facade so that scripts (tests, bench) can drive the same
// object wildcat_slam_node.cc would drive through its C++ interface.
#include <cstring>

#include "lidar_odometry.h"

extern "C" {

void *wc_odom_create(int device) { return new LidarOdometry(device); }
void wc_odom_destroy(void *h) { delete (LidarOdometry *)h; }

void wc_odom_add_imu(void *h, double t, const double acc[3], const double gyr[3]) {
  ImuData d;
  d.timestamp = t;
  for (int i = 0; i < 3; ++i) d.linear_acceleration[i] = acc[i], d.angular_velocity[i] = gyr[i];
  ((LidarOdometry *)h)->AddImuData(d);
}

// points: n records of the 48-byte hilti_ros::Point layout, in the LIDAR frame, time ascending
void wc_odom_add_scan(void *h, const void *points, uint64_t n) {
  auto cloud = std::make_shared<pcl::PointCloud<hilti_ros::Point>>();
  cloud->points.resize(n);
  if (n) std::memcpy(cloud->points.data(), points, n * sizeof(hilti_ros::Point));
  ((LidarOdometry *)h)->AddLidarScan(cloud);
}

int wc_odom_sweeps(void *h) { return ((LidarOdometry *)h)->sweeps_done(); }
uint64_t wc_odom_num_samples(void *h) { return ((LidarOdometry *)h)->num_sample_states(); }

// out[15] = t, pos[3], quat[4] (w,x,y,z), bg[3], ba[3], spare
int wc_odom_sample(void *h, uint64_t i, double *out) {
  LidarOdometry::SampleStateView v;
  if (!((LidarOdometry *)h)->sample_state(i, &v)) return 1;
  out[0] = v.timestamp;
  std::memcpy(out + 1, v.pos, 24);
  std::memcpy(out + 4, v.quat, 32);
  std::memcpy(out + 8, v.bg, 24);
  std::memcpy(out + 11, v.ba, 24);
  return 0;
}

// stats[8] = sliding surfels, fixed surfels, binary corr, unary corr, LM iterations, initial cost, final cost, termination
void wc_odom_stats(void *h, double *stats) {
  LidarOdometry *o = (LidarOdometry *)h;
  stats[0] = (double)o->sliding_window_surfels();
  stats[1] = (double)o->fixed_window_surfels();
  stats[2] = (double)o->last_correspondences(0);
  stats[3] = (double)o->last_correspondences(1);
  stats[4] = o->last_solve().iterations;
  stats[5] = o->last_solve().initial_cost;
  stats[6] = o->last_solve().final_cost;
  stats[7] = o->last_solve().termination;
}
}
