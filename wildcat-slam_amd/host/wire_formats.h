// wire_formats.h — the ROS-free half of the reference's wire / visualisation formats: everything between a
// ROS message's bytes and the records of the hot path that does not need ROS itself.
//
//   sensor_msgs/PointCloud2  <->  hilti_ros::Point     what pcl::fromROSMsg / pcl::toROSMsg do for the point type registered at
//                                                      src/common/common.h:12-28 (fields x, y, z, intensity, timestamp, ring):
//                                                      HandleLidarMessage (wildcat_slam_node.cc:46-52) and the sweep the reference
//                                                      publishes after every solve (lidar_odometry.cc:584-595)
//   surfel -> RViz SPHERE marker                       PubSurfels (surfel_extraction.cc:340-434): eigen-frame of the world
//                                                      covariance made right-handed (makeRightHanded :340-358), 3-sigma scales,
//                                                      colour from the world normal
//   last sample state -> world->imu_link transform     lidar_odometry.cc:596-602 (tf quaternion order x, y, z, w)
// A node that links ROS copies these plain structs into the message types field by field; nothing here includes a ROS header.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/wc_types.h"
#include "../csrc/dmath.h"
#ifndef WC_HAVE_REFERENCE_TYPES
#include "shim/common.h"
#endif

namespace wc_wire {

// sensor_msgs/PointField datatypes (sensor_msgs/PointField.msg)
enum : uint8_t { INT8 = 1, UINT8 = 2, INT16 = 3, UINT16 = 4, INT32 = 5, UINT32 = 6, FLOAT32 = 7, FLOAT64 = 8 };

struct PointField {
  std::string name;
  uint32_t offset;
  uint8_t datatype;
  uint32_t count;
};

struct PointCloud2 {  // the fields of sensor_msgs/PointCloud2 that describe the payload
  uint32_t height = 1, width = 0;
  std::vector<PointField> fields;
  bool is_bigendian = false;
  uint32_t point_step = 0, row_step = 0;
  std::vector<uint8_t> data;
  bool is_dense = true;
};

// the registration of hilti_ros::Point (common.h:21-28): tag name, datatype, offset inside the 48-byte struct
struct Registered {
  const char *name;
  uint8_t datatype;
  uint32_t offset, size;
};
inline const Registered *registered_fields() {
  static const Registered r[6] = {{"x", FLOAT32, 0, 4},          {"y", FLOAT32, 4, 4},           {"z", FLOAT32, 8, 4},
                                  {"intensity", FLOAT32, 16, 4}, {"timestamp", FLOAT64, 24, 8}, {"ring", UINT16, 32, 2}};
  return r;
}

// pcl::fromROSMsg for hilti_ros::Point: a registered field is filled from the message field of the SAME name, datatype and
// count 1 (pcl::FieldMatches); a field the message lacks keeps the value of a default-constructed point (PCL warns "Failed to
// find match for field" and goes on: zero here - hilti_ros::Point is a plain struct without a constructor, common.h:13-19, and
// pcl::fromROSMsg value-initialises the points, so the padding word data[3] stays 0 too); of two fields with one name the
// FIRST is taken, as PCL's mapping does.  Returns the number of
// registered fields that found their match (6 = complete); out gets width * height points.  A big-endian payload, which PCL
// refuses as well, returns -1.
inline int PointsFromCloud2(const PointCloud2 &msg, std::vector<hilti_ros::Point> &out) {
  static_assert(sizeof(hilti_ros::Point) == 48, "48-byte record");
  const size_t n = (size_t)msg.width * msg.height;
  out.assign(n, hilti_ros::Point{});
  if (msg.is_bigendian) return -1;
  const uint32_t row_step = msg.row_step ? msg.row_step : msg.point_step * msg.width;
  if (n && msg.data.size() < (size_t)(msg.height - 1) * row_step + (size_t)msg.width * msg.point_step) return -1;  // shorter than described
  int matched = 0;
  const Registered *reg = registered_fields();
  for (int f = 0; f < 6; ++f) {
    const PointField *src = nullptr;
    for (const PointField &pf : msg.fields)
      if (pf.name == reg[f].name && pf.datatype == reg[f].datatype && (pf.count == 1 || pf.count == 0)) {
        src = &pf;
        break;
      }
    if (!src) continue;
    ++matched;
    for (uint32_t r = 0; r < msg.height; ++r)
      for (uint32_t c = 0; c < msg.width; ++c) {
        const size_t at = (size_t)r * row_step + (size_t)c * msg.point_step + src->offset;
        if (at + reg[f].size > msg.data.size()) return -1;  // a payload shorter than its description
        std::memcpy((char *)&out[(size_t)r * msg.width + c] + reg[f].offset, &msg.data[at], reg[f].size);
      }
  }
  return matched;
}

// pcl::toROSMsg for hilti_ros::Point: the structs' bytes as they lie (point_step = 48), fields in registration order
inline void Cloud2FromPoints(const hilti_ros::Point *pts, size_t n, PointCloud2 &msg) {
  msg.height = 1, msg.width = (uint32_t)n;
  msg.fields.clear();
  const Registered *reg = registered_fields();
  for (int f = 0; f < 6; ++f) msg.fields.push_back({reg[f].name, reg[f].offset, reg[f].datatype, 1});
  msg.is_bigendian = false;
  msg.point_step = 48, msg.row_step = (uint32_t)(48 * n);
  msg.data.resize(48 * n);
  if (n) std::memcpy(msg.data.data(), pts, 48 * n);
  msg.is_dense = true;
}

// makeRightHanded (surfel_extraction.cc:340-358): columns normalised; if (c0 x c1) . c2 < 0 the first two columns - and their
// eigenvalues - change places.  evec: row-major 3 x 3, eigenvectors in the columns.
inline void MakeRightHanded(double evec[9], double eval[3]) {
  using namespace wc;
  V3 c[3];
  for (int k = 0; k < 3; ++k) {
    c[k] = mk3(evec[k], evec[3 + k], evec[6 + k]);
    c[k] = c[k] / norm(c[k]);
  }
  if (dot(cross(c[0], c[1]), c[2]) < 0) {
    const V3 t = c[0];
    c[0] = c[1], c[1] = t;
    const double e = eval[0];
    eval[0] = eval[1], eval[1] = e;
  }
  for (int k = 0; k < 3; ++k) evec[k] = c[k].x, evec[3 + k] = c[k].y, evec[6 + k] = c[k].z;
}

struct SurfelMarker {  // visualization_msgs::Marker of PubSurfels (:381-408): type SPHERE, ns "plane", frame "world"
  double position[3];
  double orientation[4];  // w, x, y, z
  double scale[3];        // 3 sqrt(eigenvalue)
  float color[4];         // r, g, b, a
};

// one surfel of the window (body frame + pose, the records of include/wc_types.h) -> its marker.  Eigen's
// SelfAdjointEigenSolver (ascending eigenvalues, unit eigenvectors, lower triangle) is stood in for by the library's symmetric
// 3 x 3 solver (csrc/dmath.h: same conventions); an eigenvector's SIGN is the solver's choice in Eigen too, so the orientation
// agrees with the reference's up to half-turns about the ellipsoid's own axes - the same ellipsoid on the screen.
inline SurfelMarker MarkerFromSurfel(const wc_surfel &s, const wc_pose &p) {
  using namespace wc;
  const Q4 q{p.quat[0], p.quat[1], p.quat[2], p.quat[3]};
  const M3 R = qmat(q);
  M3 C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C.m[i][j] = s.cov[3 * i + j];
  const M3 Cw = (R * C) * transpose(R);  // GetCovarianceInWorld (surfel.h:89-91)
  const V3 cw = qrot(q, mk3(s.center[0], s.center[1], s.center[2])) + mk3(p.pos[0], p.pos[1], p.pos[2]);  // surfel.h:67-69
  const V3 nw = qrot(q, mk3(s.normal[0], s.normal[1], s.normal[2]));                                      // surfel.h:78-80
  double ev[3];
  M3 V;
  eig3_sym(Cw, ev, V);
  double evec[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) evec[3 * i + j] = V.m[i][j];
  MakeRightHanded(evec, ev);
  // Quaterniond qq{rot} (:378): rotation matrix -> quaternion (Shepperd's branches, as Eigen forms it)
  auto M = [&](int r, int c) { return evec[3 * r + c]; };
  double qx, qy, qz, qw, t = M(0, 0) + M(1, 1) + M(2, 2);
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    qw = 0.5 * t;
    t = 0.5 / t;
    qx = (M(2, 1) - M(1, 2)) * t, qy = (M(0, 2) - M(2, 0)) * t, qz = (M(1, 0) - M(0, 1)) * t;
  } else {
    int i = 0;
    if (M(1, 1) > M(0, 0)) i = 1;
    if (M(2, 2) > M(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    qw = (M(k, j) - M(j, k)) * t;
    v[j] = (M(j, i) + M(i, j)) * t;
    v[k] = (M(k, i) + M(i, k)) * t;
    qx = v[0], qy = v[1], qz = v[2];
  }
  SurfelMarker m;
  m.position[0] = cw.x, m.position[1] = cw.y, m.position[2] = cw.z;
  m.orientation[0] = qw, m.orientation[1] = qx, m.orientation[2] = qy, m.orientation[3] = qz;
  for (int k = 0; k < 3; ++k) m.scale[k] = 3 * std::sqrt(ev[k]);  // (:396-398; a rounding-negative eigenvalue gives NaN there too)
  m.color[0] = (float)((nw.x + 1) / 2), m.color[1] = (float)((nw.y + 1) / 2), m.color[2] = (float)((nw.z + 1) / 2), m.color[3] = 1.f;
  return m;
}

struct StampedTransform {  // tf::StampedTransform(world -> imu_link) of lidar_odometry.cc:596-602
  double stamp;
  double origin[3];
  double rotation_xyzw[4];  // tf::Quaternion(x, y, z, w)
};

}  // namespace wc_wire
