// MOCK of sensor_msgs/PointField (test infrastructure, see tests/ros_mock/README.md)
#pragma once
#include <cstdint>
#include <string>
namespace sensor_msgs {
struct PointField {
  enum { INT8 = 1, UINT8 = 2, INT16 = 3, UINT16 = 4, INT32 = 5, UINT32 = 6, FLOAT32 = 7, FLOAT64 = 8 };
  std::string name;
  uint32_t offset = 0;
  uint8_t datatype = 0;
  uint32_t count = 0;
};
}  // namespace sensor_msgs
