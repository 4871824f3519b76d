// fx_pcl_crosscheck — runs the reference node's sequence of PCL calls (PCL >= 1.8; no ROS) on the exported golden inputs
// and compares with the oracle's outputs.  See README.md.  NOT compiled in the development image (no PCL there).
//
// Every PCL object below is configured exactly as the reference configures it
// (GAVLab/feature_extraction src/feature_extraction_node.cpp: the line numbers are cited at each call); the three
// constants the reference hard-codes for the VLP-16 come from the fixture's .txt so that the 64- / 128-ring fixtures run.
#include <pcl/common/transforms.h>
#include <pcl/features/3dsc.h>
#include <pcl/filters/passthrough.h>
#include <pcl/io/pcd_io.h>
#include <pcl/point_types.h>
#include <pcl/search/kdtree.h>
#include <pcl/segmentation/extract_clusters.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

typedef pcl::PointXYZI Pt;
typedef pcl::PointCloud<Pt> Cloud;

struct Cfg {
  std::map<std::string, double> v;
  double operator[](const char *k) const { return v.at(k); }
};

static Cfg read_cfg(const std::string &path) {
  Cfg c;
  std::ifstream in(path);
  std::string k;
  double x;
  while (in >> k >> x) c.v[k] = x;
  return c;
}

// clusters of `cloud`, as (tolerance, min, max) make them (ref: node.cpp:269-276 and :222-229)
static std::vector<pcl::PointIndices> clusters_of(const Cloud::Ptr &cloud, double tolerance, int min_size, int max_size) {
  std::vector<pcl::PointIndices> out;
  pcl::EuclideanClusterExtraction<Pt> ec;
  ec.setInputCloud(cloud);
  ec.setClusterTolerance(tolerance);
  ec.setMinClusterSize(min_size);
  ec.setMaxClusterSize(max_size);
  ec.extract(out);
  return out;
}

static uint32_t bits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}
// bit-equal except that +0 and -0 are the same value
static size_t differing(const float *a, const float *b, size_t n) {
  size_t bad = 0;
  for (size_t i = 0; i < n; ++i) bad += (bits(a[i]) != bits(b[i]) && !(a[i] == 0.0f && b[i] == 0.0f)) ? 1 : 0;
  return bad;
}

static bool run_fixture(const std::string &dir, const std::string &name) {
  const Cfg P = read_cfg(dir + "/" + name + ".txt");
  Cloud::Ptr full(new Cloud);
  if (pcl::io::loadPCDFile<Pt>(dir + "/" + name + ".pcd", *full) != 0) return false;

  // ---- getElevationAngles (ref: node.cpp:147-156): into `intensity`, before the rotation
  for (auto &p : full->points) {
    const double x = p.x, y = p.y, z = p.z;
    const double az = std::atan2(y, x);
    const double xp = std::cos(az) * x + std::sin(az) * y;
    p.intensity = std::atan2(z, xp) * 180 / M_PI;
  }
  // ---- rotateCloud (ref: node.cpp:159-167)
  {
    Eigen::Affine3f t = Eigen::Affine3f::Identity();
    t.translation() << 0.0, 0.0, 0.0;
    t.rotate(Eigen::AngleAxisf(P["pitch"], Eigen::Vector3f::UnitY()) * Eigen::AngleAxisf(P["roll"], Eigen::Vector3f::UnitX()));
    Cloud rotated;
    pcl::transformPointCloud(*full, rotated, t);
    *full = rotated;
  }
  // ---- filterCloud (ref: node.cpp:169-183): the copy that is filtered, z then y then x
  Cloud::Ptr cloud(new Cloud(*full));
  {
    pcl::PassThrough<Pt> f;
    f.setInputCloud(cloud);
    f.setFilterFieldName("z");
    f.setFilterLimits(P["z_min"], P["z_max"]);
    f.filter(*cloud);
    f.setFilterFieldName("y");
    f.setFilterLimits(P["y_min"], P["y_max"]);
    f.filter(*cloud);
    f.setFilterFieldName("x");
    f.setFilterLimits(P["x_min"], P["x_max"]);
    f.filter(*cloud);
  }
  // ---- estimateKeypoints (ref: node.cpp:185-259)
  Cloud::Ptr keypoints_full(new Cloud), keypoints(new Cloud);
  {
    pcl::PassThrough<Pt> ring;
    ring.setInputCloud(cloud);
    ring.setFilterFieldName("intensity");
    const int n_rings = (int)P["n_rings"];
    for (int i = 0; i < n_rings; ++i) {
      const double centre = P["el0_deg"] + i * P["el_step_deg"];  // (i-7)*2-1 for the VLP-16 (ref: node.cpp:200)
      const double half = P["el_step_deg"] / 2.0;
      Cloud::Ptr channel(new Cloud);
      ring.setFilterLimits(centre - half, centre + half);
      ring.filter(*channel);
      if (channel->points.empty()) continue;
      // getCylinderSegments (ref: node.cpp:261-327)
      for (const pcl::PointIndices &c : clusters_of(channel, P["cluster_tolerance"], (int)P["cluster_min_count"], (int)P["cluster_max_count"])) {
        double sx = 0, sy = 0, sz = 0, minx = 1000.0, maxx = -1000.0, miny = 1000.0, maxy = -1000.0;
        for (int j : c.indices) {
          const double x = channel->points[j].x, y = channel->points[j].y, z = channel->points[j].z;
          sx += x, sy += y, sz += z;
          if (x < minx) minx = x;
          if (y < miny) miny = y;
          if (x > maxx) maxx = x;
          if (y > maxy) maxy = y;
        }
        const double diameter = std::pow(std::pow(maxx - minx, 2) + std::pow(maxy - miny, 2), 0.5);
        if (diameter < 2 * P["cluster_radius_threshold"]) {
          Pt centroid;
          const double n = (double)c.indices.size();
          centroid.x = sx / n, centroid.y = sy / n, centroid.z = sz / n;
          centroid.intensity = channel->points[c.indices[0]].intensity;
          keypoints_full->points.push_back(centroid);
        }
      }
    }
    if (!keypoints_full->points.empty()) {
      // secondary merge (ref: node.cpp:209-257): z replaced by the scaled elevation while clustering
      std::vector<double> zhold(keypoints_full->points.size());
      for (size_t i = 0; i < zhold.size(); ++i) {
        zhold[i] = keypoints_full->points[i].z;
        keypoints_full->points[i].z = keypoints_full->points[i].intensity * 0.75 * P["cluster_radius_threshold"] / 2;
      }
      const std::vector<pcl::PointIndices> merged =
          clusters_of(keypoints_full, P["cluster_radius_threshold"], (int)P["number_detection_channels"], (int)P["secondary_max"]);
      for (size_t i = 0; i < zhold.size(); ++i) keypoints_full->points[i].z = zhold[i];
      for (const pcl::PointIndices &c : merged) {
        double sx = 0, sy = 0, sz = 0;
        for (int j : c.indices) sx += keypoints_full->points[j].x, sy += keypoints_full->points[j].y, sz += keypoints_full->points[j].z;
        Pt centroid;
        const double n = (double)c.indices.size();
        centroid.x = sx / n, centroid.y = sy / n, centroid.z = sz / n;
        centroid.intensity = keypoints_full->points[c.indices[0]].intensity;
        keypoints->points.push_back(centroid);
      }
    }
  }
  // ---- estimateDescriptors (ref: node.cpp:329-355)
  pcl::PointCloud<pcl::ShapeContext1980> descriptors;
  if (!keypoints->points.empty()) {
    pcl::PointCloud<pcl::Normal>::Ptr normals(new pcl::PointCloud<pcl::Normal>);
    pcl::Normal up;
    up.normal_x = 0.0f, up.normal_y = 0.0f, up.normal_z = 1.0f;
    for (size_t i = 0; i < full->points.size(); ++i) normals->points.push_back(up);
    pcl::search::KdTree<Pt>::Ptr tree(new pcl::search::KdTree<Pt>);
    pcl::ShapeContext3DEstimation<Pt, pcl::Normal, pcl::ShapeContext1980> sc;
    sc.setInputCloud(keypoints);
    sc.setSearchSurface(full);
    sc.setInputNormals(normals);
    sc.setSearchMethod(tree);
    sc.setRadiusSearch(P["descriptor_radius"]);
    sc.setMinimalRadius(P["descriptor_radius"] / 10.0);
    sc.setPointDensityRadius(P["descriptor_radius"] / 5.0);
    sc.compute(descriptors);
  }

  // ---- compare with the oracle's outputs
  std::ifstream in(dir + "/" + name + "_expected.bin", std::ios::binary);
  uint32_t cnt[3];
  in.read((char *)cnt, 12);
  std::vector<float> e_filt((size_t)cnt[0] * 4), e_cand((size_t)cnt[1] * 4), e_kp((size_t)cnt[2] * 4), e_desc((size_t)cnt[2] * 1989);
  std::vector<uint32_t> e_nb(cnt[2]);
  in.read((char *)e_filt.data(), e_filt.size() * 4);
  in.read((char *)e_cand.data(), e_cand.size() * 4);
  in.read((char *)e_kp.data(), e_kp.size() * 4);
  in.read((char *)e_nb.data(), e_nb.size() * 4);
  in.read((char *)e_desc.data(), e_desc.size() * 4);
  auto flat = [](const Cloud &c) {
    std::vector<float> v;
    for (const Pt &p : c.points) v.insert(v.end(), {p.x, p.y, p.z, p.intensity});
    return v;
  };
  bool ok = true;
  auto check_cloud = [&](const char *what, const Cloud &got, const std::vector<float> &want) {
    const std::vector<float> g = flat(got);
    const size_t bad = g.size() == want.size() ? differing(g.data(), want.data(), g.size()) : (size_t)-1;
    std::printf("  %-16s %zu points (expected %zu): %s\n", what, got.points.size(), want.size() / 4,
                bad == 0 ? "bit-identical" : (bad == (size_t)-1 ? "COUNT DIFFERS" : "VALUES DIFFER"));
    ok = ok && bad == 0;
  };
  std::printf("%s\n", name.c_str());
  check_cloud("~cloud", *cloud, e_filt);
  check_cloud("keypoints_full", *keypoints_full, e_cand);
  check_cloud("~keypoints", *keypoints, e_kp);
  if (descriptors.points.size() == cnt[2]) {
    size_t beyond = 0, nan_mismatch = 0;
    double worst = 0;
    for (size_t k = 0; k < descriptors.points.size(); ++k)
      for (int b = 0; b < 1989; ++b) {
        const float g = b < 1980 ? descriptors.points[k].descriptor[b] : descriptors.points[k].rf[b - 1980], w = e_desc[k * 1989 + b];
        if (std::isnan(g) != std::isnan(w)) {
          ++nan_mismatch;
          continue;
        }
        if (std::isnan(g)) continue;
        const double d = std::fabs((double)g - (double)w);
        if (d > worst) worst = d;
        if (d > 1e-5) ++beyond;
      }
    std::printf("  %-16s %zu x 1989 values: %zu beyond 1e-5, %zu NaN mismatches, largest difference %.3g\n", "descriptors",
                descriptors.points.size(), beyond, nan_mismatch, worst);
    ok = ok && beyond == 0 && nan_mismatch == 0;
  } else if (cnt[2]) {
    std::printf("  descriptors      COUNT DIFFERS (%zu, expected %u)\n", descriptors.points.size(), cnt[2]);
    ok = false;
  }
  std::printf("  => %s\n", ok ? "PASS" : "FAIL");
  return ok;
}

int main(int argc, char **argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: fx_pcl_crosscheck DIR [fixture names...]   (DIR: output of export_fixtures.py)\n");
    return 2;
  }
  const std::string dir = argv[1];
  std::vector<std::string> names(argv + 2, argv + argc);
  if (names.empty())
    names = {"vlp16_default_seed1000", "vlp16_launch_seed1000", "vlp16_launch_seed1001_unleveled", "hdl64_64x2048_launch_seed10",
             "dense_128x2048_R2m_launch_seed10"};
  bool all = true;
  for (const std::string &n : names) all = run_fixture(dir, n) && all;
  return all ? 0 : 1;
}
