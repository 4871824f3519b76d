// fx_cli — one sweep from a .pcd file through the detector + descriptor on the GPU
// (BASELINE config 1 without ROS): the reference's cloudCallback with files instead of topics.
//
//   fx_cli scan.pcd [--launch] [--roll R] [--pitch P] [--out PREFIX] [--device N]
//   fx_cli --synth SEED out.pcd        write a synthetic VLP-16 sweep
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

#include "fx_node.hpp"
#include "fx_pcd.hpp"

int main(int argc, char **argv) {
  try {
    if (argc >= 4 && !std::strcmp(argv[1], "--synth")) {
      fx_synth_cfg cfg;
      fx_synth_cfg_vlp16(&cfg, std::strtoull(argv[2], nullptr, 10));
      fx::PointCloud c(cfg.n_rings * cfg.n_az);
      fx_synth_scan(&cfg, &c[0].x, (uint32_t)c.size());
      fx::write_pcd(argv[3], c, true);
      std::printf("wrote %zu points to %s\n", c.size(), argv[3]);
      return 0;
    }
    if (argc < 2) {
      std::fprintf(stderr, "usage: fx_cli scan.pcd [--launch] [--roll R] [--pitch P] [--out PREFIX] [--device N]\n"
                           "       fx_cli --synth SEED out.pcd\n");
      return 2;
    }
    std::string in = argv[1], out;
    bool launch = false;
    double roll = 0.0, pitch = 0.0;
    int device = 0;
    for (int i = 2; i < argc; ++i) {
      if (!std::strcmp(argv[i], "--launch")) launch = true;
      else if (!std::strcmp(argv[i], "--roll") && i + 1 < argc) roll = std::atof(argv[++i]);
      else if (!std::strcmp(argv[i], "--pitch") && i + 1 < argc) pitch = std::atof(argv[++i]);
      else if (!std::strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
      else if (!std::strcmp(argv[i], "--device") && i + 1 < argc) device = std::atoi(argv[++i]);
      else {
        std::fprintf(stderr, "unknown argument %s\n", argv[i]);
        return 2;
      }
    }
    fx::PointCloud cloud_full = fx::read_pcd(in);
    fx::FeatureExtractionNode node(device, (uint32_t)std::max<size_t>(cloud_full.size(), 1024));
    if (launch) node.useLaunchPreset();
    node.roll = roll;
    node.pitch = pitch;
    fx::PointCloud cloud, keypoints, keypoint_cloud;
    fx::DescriptorCloud descriptors;
    node.cloudCallback(cloud_full, cloud, keypoints, keypoint_cloud, descriptors);
    std::printf("points %zu  filtered %zu  keypoint_cloud %zu  keypoints %zu  flags 0x%x\n", cloud_full.size(), cloud.size(),
                keypoint_cloud.size(), keypoints.size(), node.lastFlags());
    for (size_t k = 0; k < keypoints.size(); ++k) {
      double mass = 0;
      if (k < descriptors.size())
        for (float v : descriptors[k].descriptor) mass += v;
      std::printf("kp %3zu  %10.5f %10.5f %10.5f  el %6.2f  |desc|_1 %.6g\n", k, keypoints[k].x, keypoints[k].y,
                  keypoints[k].z, keypoints[k].intensity, mass);
    }
    if (!out.empty()) {
      fx::write_pcd(out + "_cloud.pcd", cloud);
      fx::write_pcd(out + "_keypoints.pcd", keypoints);
      fx::write_pcd(out + "_keypoint_cloud.pcd", keypoint_cloud);
      std::ofstream f(out + "_descriptors.f32", std::ios::binary);
      f.write(reinterpret_cast<const char *>(descriptors.data()), (std::streamsize)(descriptors.size() * sizeof(fx::Descriptor)));
    }
    return 0;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "fx_cli: %s\n", e.what());
    return 1;
  }
}
