// MOCK of tf/transform_datatypes.h (test infrastructure, see tests/ros_mock/README.md): the quaternion -> roll / pitch /
// yaw conversion the shell's imuCallback uses, stated from the published formula of tf::Matrix3x3::getEulerYPR
// (rotation matrix of a unit quaternion; pitch = -asin(m20), roll = atan2(m21, m22), yaw = atan2(m10, m00)).
#pragma once
#include <cmath>

#include "../sensor_msgs/Imu.h"
namespace tf {
struct Quaternion {
  double x = 0, y = 0, z = 0, w = 1;
};
inline void quaternionMsgToTF(const geometry_msgs::Quaternion &m, Quaternion &q) { q.x = m.x, q.y = m.y, q.z = m.z, q.w = m.w; }
class Matrix3x3 {
 public:
  explicit Matrix3x3(const Quaternion &q) {
    const double d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w, s = 2.0 / d;
    const double xs = q.x * s, ys = q.y * s, zs = q.z * s;
    const double wx = q.w * xs, wy = q.w * ys, wz = q.w * zs, xx = q.x * xs, xy = q.x * ys, xz = q.x * zs, yy = q.y * ys,
                 yz = q.y * zs, zz = q.z * zs;
    m[0][0] = 1.0 - (yy + zz), m[0][1] = xy - wz, m[0][2] = xz + wy;
    m[1][0] = xy + wz, m[1][1] = 1.0 - (xx + zz), m[1][2] = yz - wx;
    m[2][0] = xz - wy, m[2][1] = yz + wx, m[2][2] = 1.0 - (xx + yy);
  }
  void getRPY(double &roll, double &pitch, double &yaw) const {
    if (std::fabs(m[2][0]) >= 1.0) {  // gimbal lock
      yaw = 0.0;
      const double delta = std::atan2(m[2][1], m[2][2]);
      if (m[2][0] < 0) {
        pitch = M_PI / 2.0;
        roll = delta;
      } else {
        pitch = -M_PI / 2.0;
        roll = delta;
      }
      return;
    }
    pitch = -std::asin(m[2][0]);
    const double c = std::cos(pitch);
    roll = std::atan2(m[2][1] / c, m[2][2] / c);
    yaw = std::atan2(m[1][0] / c, m[0][0] / c);
  }

 private:
  double m[3][3];
};
}  // namespace tf
