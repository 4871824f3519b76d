// ROS1 shell around fx::FeatureExtractionNode.
//
// Same node name, private parameters, topics and message layouts as the reference node
// (ref: src/feature_extraction_node.cpp:3-50 constructor, :57-70 imuCallback, :72-145 cloudCallback,
// :379-387 main), with the per-scan numerics behind libfx_hip.so.  The shell needs roscpp,
// sensor_msgs and tf only: no PCL, pcl_ros or pcl_conversions — it reads and writes the PointCloud2
// byte layouts those libraries would produce.
//
// ROS is not installable in the development image (no network), so this file has never been built against roscpp:
// the test-suite compiles it against the stand-in headers of tests/ros_mock (a mock, clearly not ROS) and drives
// imuCallback / cloudCallback with a driver-style PointCloud2 and an Imu (tests/test_ros_shell.py: the four published
// byte buffers against the oracle and against fx_pack_features / fx_pack_pointxyzi).  ros/CMakeLists.txt builds it
// where catkin, roscpp, sensor_msgs and tf exist.
#include <ros/ros.h>
#include <sensor_msgs/Imu.h>
#include <sensor_msgs/PointCloud2.h>
#include <sensor_msgs/PointField.h>
#include <tf/transform_datatypes.h>

#include <cstring>
#include <string>

#include "../feature_extraction_amd/csrc/fx_node.hpp"

namespace {

sensor_msgs::PointField field(const std::string &name, uint32_t offset, uint32_t count) {
  sensor_msgs::PointField f;
  f.name = name;
  f.offset = offset;
  f.datatype = sensor_msgs::PointField::FLOAT32;
  f.count = count;
  return f;
}

// pcl::PointXYZI as pcl_ros publishes it: 32-byte records, x y z at 0 4 8, intensity at 16
void toMsg(const fx::PointCloud &cloud, const std_msgs::Header &header, sensor_msgs::PointCloud2 &msg) {
  msg.header = header;
  msg.height = 1;
  msg.width = (uint32_t)cloud.size();
  msg.fields = {field("x", 0, 1), field("y", 4, 1), field("z", 8, 1), field("intensity", 16, 1)};
  msg.is_bigendian = false;
  msg.point_step = 32;
  msg.row_step = msg.point_step * msg.width;
  msg.is_dense = true;
  msg.data.assign((size_t)msg.row_step, 0);
  for (size_t i = 0; i < cloud.size(); ++i) {
    uint8_t *rec = &msg.data[i * 32];
    const float one = 1.0f;
    std::memcpy(rec, &cloud[i], 12);
    std::memcpy(rec + 12, &one, 4);  // PCL_ADD_POINT4D's fourth coordinate
    std::memcpy(rec + 16, &cloud[i].intensity, 4);
  }
}

// pcl::PointDescriptor (ref: node.h:35-53): x y z at 0 4 8, intensity at 16, shape_context[1980] at 20, rf[9] at 7940
void toMsg(const fx::PointCloud &keypoints, const fx::DescriptorCloud &descriptors, const std_msgs::Header &header,
           sensor_msgs::PointCloud2 &msg) {
  msg.header = header;
  msg.height = 1;
  msg.width = (uint32_t)descriptors.size();
  msg.fields = {field("x", 0, 1), field("y", 4, 1), field("z", 8, 1), field("intensity", 16, 1),
                field("shape_context", 20, FX_DESC_BINS), field("rf", 7940, FX_DESC_RF)};
  msg.is_bigendian = false;
  msg.point_step = FX_FEATURE_RECORD_BYTES;
  msg.row_step = msg.point_step * msg.width;
  msg.is_dense = true;
  msg.data.assign((size_t)msg.row_step, 0);
  for (size_t i = 0; i < descriptors.size(); ++i) {
    uint8_t *rec = &msg.data[i * FX_FEATURE_RECORD_BYTES];
    const float one = 1.0f;
    std::memcpy(rec, &keypoints[i], 12);
    std::memcpy(rec + 12, &one, 4);
    std::memcpy(rec + 16, &keypoints[i].intensity, 4);
    std::memcpy(rec + 20, descriptors[i].descriptor, sizeof(float) * FX_DESC_BINS);
    std::memcpy(rec + 7940, descriptors[i].rf, sizeof(float) * FX_DESC_RF);
  }
}

// what pcl_conversions::toPCL + pcl::fromPCLPointCloud2 keep of a driver message (ref: node.cpp:79-81):
// the float32 fields x, y, z by name; any other field (ring, time, ...) is ignored; intensity is
// overwritten by getElevationAngles anyway (ref: node.cpp:154)
bool fromMsg(const sensor_msgs::PointCloud2 &msg, fx::PointCloud &cloud) {
  int ox = -1, oy = -1, oz = -1;
  for (const sensor_msgs::PointField &f : msg.fields) {
    if (f.datatype != sensor_msgs::PointField::FLOAT32) continue;
    if (f.name == "x") ox = (int)f.offset;
    if (f.name == "y") oy = (int)f.offset;
    if (f.name == "z") oz = (int)f.offset;
  }
  if (ox < 0 || oy < 0 || oz < 0 || msg.is_bigendian) return false;
  const size_t n = (size_t)msg.width * msg.height;
  cloud.resize(n);
  for (size_t i = 0; i < n; ++i) {
    const uint8_t *rec = &msg.data[(i / msg.width) * msg.row_step + (i % msg.width) * msg.point_step];
    std::memcpy(&cloud[i].x, rec + ox, 4);
    std::memcpy(&cloud[i].y, rec + oy, 4);
    std::memcpy(&cloud[i].z, rec + oz, 4);
    cloud[i].intensity = 0.0f;
  }
  return true;
}

class Shell {
 public:
  Shell() : nh_("~") {
    // ref: node.cpp:9-34 — same private parameter names and defaults
    nh_.param("cloud_leveling", node_.levelCloud, true);
    nh_.param("x_min", node_.xMin, 0.0);
    nh_.param("x_max", node_.xMax, 75.0);
    nh_.param("y_min", node_.yMin, -30.0);
    nh_.param("y_max", node_.yMax, 30.0);
    nh_.param("z_min", node_.zMin, -1.5);
    nh_.param("z_max", node_.zMax, 5.0);
    nh_.param("cluster_tolerance", node_.clusterTolerance, 0.65);
    nh_.param("cluster_min_count", node_.clusterMinCount, 5);
    nh_.param("cluster_max_count", node_.clusterMaxCount, 50);
    nh_.param("cluster_radius_threshold", node_.clusterRadiusThreshold, 0.15);
    nh_.param("number_detection_channels", node_.detectionChannelThreshold, 1);
    nh_.param("estimate_descriptors", node_.descriptorEstimation, true);
    nh_.param("descriptor_radius", node_.descriptorRadius, 2.5);
    // ref: node.cpp:41-48 — same topics, queue size 0
    kp_pub_ = nh_.advertise<sensor_msgs::PointCloud2>("keypoints", 0);
    kpc_pub_ = nh_.advertise<sensor_msgs::PointCloud2>("keypoint_cloud", 0);
    cloud_pub_ = nh_.advertise<sensor_msgs::PointCloud2>("cloud", 0);
    if (node_.descriptorEstimation) feature_pub_ = nh_.advertise<sensor_msgs::PointCloud2>("features", 0);
    pc_sub_ = nh_.subscribe("/velodyne_points", 0, &Shell::cloudCallback, this);
    imu_sub_ = nh_.subscribe("/xsens/data", 0, &Shell::imuCallback, this);
  }

 private:
  // ref: node.cpp:57-70
  void imuCallback(const sensor_msgs::ImuConstPtr &msg) {
    tf::Quaternion quat;
    double imu_roll, imu_pitch, yaw;
    tf::quaternionMsgToTF(msg->orientation, quat);
    tf::Matrix3x3(quat).getRPY(imu_roll, imu_pitch, yaw);
    node_.imuCallback(imu_roll, imu_pitch);
  }

  // ref: node.cpp:72-145
  void cloudCallback(const sensor_msgs::PointCloud2ConstPtr &msg) {
    if (!fromMsg(*msg, cloud_full_)) {
      ROS_ERROR_THROTTLE(5.0, "feature_extraction: PointCloud2 without little-endian float32 x, y, z fields");
      return;
    }
    try {
      node_.cloudCallback(cloud_full_, cloud_, keypoints_, keypoint_cloud_, descriptors_);
    } catch (const std::exception &e) {
      ROS_ERROR("feature_extraction: %s", e.what());
      return;
    }
    if (node_.lastFlags()) ROS_WARN_THROTTLE(5.0, "feature_extraction: capacity flags 0x%x (see include/fx.h)", node_.lastFlags());
    sensor_msgs::PointCloud2 out;
    if (node_.descriptorEstimation) {  // ref: node.cpp:113-124
      toMsg(keypoints_, descriptors_, msg->header, out);
      feature_pub_.publish(out);
    }
    toMsg(keypoints_, msg->header, out);  // ref: node.cpp:129-139
    kp_pub_.publish(out);
    toMsg(keypoint_cloud_, msg->header, out);
    kpc_pub_.publish(out);
    toMsg(cloud_, msg->header, out);
    cloud_pub_.publish(out);
  }

  ros::NodeHandle nh_;
  ros::Publisher kp_pub_, kpc_pub_, cloud_pub_, feature_pub_;
  ros::Subscriber pc_sub_, imu_sub_;
  fx::FeatureExtractionNode node_;
  fx::PointCloud cloud_full_, cloud_, keypoints_, keypoint_cloud_;
  fx::DescriptorCloud descriptors_;
};

}  // namespace

int main(int argc, char **argv) {
  ros::init(argc, argv, "feature_extraction_node");  // ref: node.cpp:382
  Shell shell;
  ros::spin();
  return 0;
}
