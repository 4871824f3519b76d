// MOCK of std_msgs/Header (test infrastructure, see tests/ros_mock/README.md)
#pragma once
#include <cstdint>
#include <string>
namespace ros {
struct Time {
  uint32_t sec = 0, nsec = 0;
};
}  // namespace ros
namespace std_msgs {
struct Header {
  uint32_t seq = 0;
  ros::Time stamp;
  std::string frame_id;
};
}  // namespace std_msgs
