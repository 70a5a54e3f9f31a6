// shim/common.h — minimal stand-ins for the reference's PCL / Eigen based types so that the host facade builds on a
// box without ROS, PCL or Eigen.  Layouts mirror src/common/common.h:12-35 of the reference:
//   hilti_ros::Point : 48-byte record (float x,y,z,pad @0, float intensity @16, double time @24, uint16 ring @32)
//   ImuData          : timestamp + linear_acceleration + angular_velocity
//   pcl::PointCloud<T>::Ptr : shared_ptr to a container of points (begin/end/size/push_back)
// Where the real headers exist (ROS build) include the reference's common/common.h instead and define
// WC_HAVE_REFERENCE_TYPES before including lidar_odometry.h.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

struct Vec3d {
  double v[3];
  double &operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
  double x() const { return v[0]; }
  double y() const { return v[1]; }
  double z() const { return v[2]; }
};

namespace hilti_ros {
struct alignas(16) Point {
  float x, y, z, pad;
  float intensity;
  float pad1;
  double time;
  std::uint16_t ring;
};
static_assert(sizeof(Point) == 48, "hilti_ros::Point must stay 48 bytes (common.h:12-28)");
}  // namespace hilti_ros

struct ImuData {
  double timestamp;
  Vec3d linear_acceleration;
  Vec3d angular_velocity;
};

namespace pcl {
template <typename T>
struct PointCloud {
  using Ptr = std::shared_ptr<PointCloud<T>>;
  std::vector<T> points;
  typename std::vector<T>::const_iterator begin() const { return points.begin(); }
  typename std::vector<T>::const_iterator end() const { return points.end(); }
  size_t size() const { return points.size(); }
  void push_back(const T &p) { points.push_back(p); }
};
}  // namespace pcl
