// MOCK of sensor_msgs/Imu (test infrastructure, see tests/ros_mock/README.md): only the orientation the shell reads
#pragma once
#include <memory>

#include "../std_msgs/Header.h"
namespace geometry_msgs {
struct Quaternion {
  double x = 0, y = 0, z = 0, w = 1;
};
}  // namespace geometry_msgs
namespace sensor_msgs {
struct Imu {
  std_msgs::Header header;
  geometry_msgs::Quaternion orientation;
};
typedef std::shared_ptr<const Imu> ImuConstPtr;
}  // namespace sensor_msgs
