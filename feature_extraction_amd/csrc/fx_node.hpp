// fx_node.hpp — C++ host mirror of the reference's FeatureExtractionNode (no ROS, no PCL).
//
// Same parameter names, defaults and call order as the reference class
// (ref: include/feature_extraction/feature_extraction_node.h:58-132,
//       src/feature_extraction_node.cpp:3-145), with the per-scan numerics behind the C-ABI of
// include/fx.h.  A ROS shell only has to convert messages and call cloudCallback().
#ifndef FX_NODE_HPP_
#define FX_NODE_HPP_
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/fx.h"

namespace fx {

struct Point {  // pcl::PointXYZI payload; `intensity` carries the elevation angle after the callback (ref: node.cpp:154)
  float x, y, z, intensity;
};
typedef std::vector<Point> PointCloud;
struct Descriptor {  // pcl::ShapeContext1980 (ref: node.h:75)
  float descriptor[FX_DESC_BINS];
  float rf[FX_DESC_RF];
};
typedef std::vector<Descriptor> DescriptorCloud;

class FeatureExtractionNode {
 public:
  // --- the node's members (ref: node.h:115-127), defaults of the constructor (ref: node.cpp:9-34)
  double zMin = -1.5, zMax = 5.0, xMin = 0.0, xMax = 75.0, yMin = -30.0, yMax = 30.0;
  double roll = 0.0, pitch = 0.0;  // the reference leaves these uninitialised until the first IMU message
  bool levelCloud = true;
  double clusterTolerance = 0.65;
  int clusterMinCount = 5, clusterMaxCount = 50;
  double clusterRadiusThreshold = 0.15;
  int detectionChannelThreshold = 1;
  double descriptorRadius = 2.5;
  bool descriptorEstimation = true;
  // --- hard-coded in the reference (ref: node.cpp:195, 200, 227)
  int nRings = 16;
  double el0Deg = -15.0, elStepDeg = 2.0;
  int secondaryMax = 16;
  // --- capacities of the device context (non-zero fields override fx_limits_default(1, max_points); call reset() after a change)
  fx_limits limitsOverride{};

  explicit FeatureExtractionNode(int device = 0, uint32_t max_points = 1u << 18) : device_(device), max_points_(max_points) {}
  ~FeatureExtractionNode() {
    if (ctx_) fx_destroy(ctx_);
  }
  FeatureExtractionNode(const FeatureExtractionNode &) = delete;
  FeatureExtractionNode &operator=(const FeatureExtractionNode &) = delete;

  // the launch file's preset (ref: launch/keypoint_playback.launch:17-33)
  void useLaunchPreset() {
    clusterTolerance = 1.0, clusterMinCount = 1, clusterMaxCount = 1000, clusterRadiusThreshold = 0.2;
    detectionChannelThreshold = 2, xMax = 100.0, xMin = 0.0, yMax = 50.0, yMin = -50.0, zMax = 4.0, zMin = -1.5;
    descriptorRadius = 2.5;
    reset();
  }
  // call after changing any parameter
  void reset() {
    if (ctx_) fx_destroy(ctx_);
    ctx_ = nullptr;
  }

  // ref: node.cpp:57-70 (tf::Matrix3x3(quat).getRPY is the shell's job; this takes its result)
  void imuCallback(double imu_roll, double imu_pitch) {
    if (levelCloud) {
      pitch = imu_pitch;
      roll = imu_roll - M_PI;
    } else {
      roll = 0.0;
      pitch = 0.0;
    }
  }

  // ref: node.cpp:72-145 — body between fromPCLPointCloud2 (:81) and the publishers (:117-139).
  // cloud_full: input scan; on return its intensity is NOT modified (the elevation-tagged, rotated
  // copy lives on the device only).  cloud = ~cloud, keypoints = ~keypoints,
  // keypoint_cloud = ~keypoint_cloud, descriptors = ShapeContext1980 per keypoint (~features).
  void cloudCallback(const PointCloud &cloud_full, PointCloud &cloud, PointCloud &keypoints, PointCloud &keypoint_cloud,
                     DescriptorCloud &descriptors) {
    ensure();
    fx_scan_desc scan;
    scan.points = cloud_full.data();
    scan.n_points = (uint32_t)cloud_full.size();
    scan.stride_bytes = sizeof(Point);
    scan.roll = roll;
    scan.pitch = pitch;
    fx_batch_view v;
    check(fx_process_batch(ctx_, &scan, 1, FX_OUT_HOST | FX_OUT_CLOUDS, &v));
    last_flags_ = v.h_flags[0];
    copy_cloud(v.h_filtered, v.h_n_filtered[0], cloud);
    copy_cloud(v.h_kpc, v.h_n_kpc[0], keypoint_cloud);
    copy_cloud(v.h_keypoints, v.h_n_keypoints[0], keypoints);
    descriptors.resize(descriptorEstimation ? v.h_n_keypoints[0] : 0);
    if (!descriptors.empty()) std::memcpy(descriptors.data(), v.h_descriptors, descriptors.size() * sizeof(Descriptor));
  }

  // ref: node.h:84 — the one public stage of the reference class
  void filterCloud(PointCloud &cloud) {
    PointCloud filtered, kp, kpc;
    DescriptorCloud d;
    const bool de = descriptorEstimation;
    if (de) {
      descriptorEstimation = false;
      reset();
    }
    cloudCallback(cloud, filtered, kp, kpc, d);
    if (de) {
      descriptorEstimation = true;
      reset();
    }
    cloud.swap(filtered);
  }

  uint32_t lastFlags() const { return last_flags_; }
  fx_ctx *context() {
    ensure();
    return ctx_;
  }

 private:
  void ensure() {
    if (ctx_) return;
    fx_params p;
    fx_params_default(&p);
    p.cloud_leveling = levelCloud;
    p.x_min = xMin, p.x_max = xMax, p.y_min = yMin, p.y_max = yMax, p.z_min = zMin, p.z_max = zMax;
    p.cluster_tolerance = clusterTolerance;
    p.cluster_min_count = clusterMinCount;
    p.cluster_max_count = clusterMaxCount;
    p.cluster_radius_threshold = clusterRadiusThreshold;
    p.number_detection_channels = detectionChannelThreshold;
    p.estimate_descriptors = descriptorEstimation;
    p.descriptor_radius = descriptorRadius;
    p.n_rings = nRings, p.el0_deg = el0Deg, p.el_step_deg = elStepDeg, p.secondary_max = secondaryMax;
    fx_limits l;
    fx_limits_default(&l, 1, max_points_);
    l.max_kpc_points = max_points_;
    {
      const uint32_t *ov = reinterpret_cast<const uint32_t *>(&limitsOverride);
      uint32_t *dst = reinterpret_cast<uint32_t *>(&l);
      for (size_t i = 2; i < sizeof(fx_limits) / 4; ++i)
        if (ov[i]) dst[i] = ov[i];
    }
    if (!limitsOverride.max_total_keypoints) l.max_total_keypoints = l.max_keypoints;
    check(FX_CHECK_ABI());  // (this translation unit's fx.h against the library's)
    check(fx_create(&p, &l, device_, &ctx_));
  }
  static void check(fx_status s) {
    if (s != FX_OK) throw std::runtime_error(std::string(fx_status_str(s)) + ": " + fx_last_error());
  }
  static void copy_cloud(const float *src, uint32_t n, PointCloud &dst) {
    dst.resize(n);
    if (n) std::memcpy(dst.data(), src, (size_t)n * sizeof(Point));
  }
  int device_;
  uint32_t max_points_;
  fx_ctx *ctx_ = nullptr;
  uint32_t last_flags_ = 0;
};

}  // namespace fx
#endif
