// MOCK of sensor_msgs/PointCloud2 (test infrastructure, see tests/ros_mock/README.md)
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "../std_msgs/Header.h"
#include "PointField.h"
namespace sensor_msgs {
struct PointCloud2 {
  std_msgs::Header header;
  uint32_t height = 0, width = 0;
  std::vector<PointField> fields;
  bool is_bigendian = false;
  uint32_t point_step = 0, row_step = 0;
  std::vector<uint8_t> data;
  bool is_dense = false;
};
typedef std::shared_ptr<const PointCloud2> PointCloud2ConstPtr;
}  // namespace sensor_msgs
