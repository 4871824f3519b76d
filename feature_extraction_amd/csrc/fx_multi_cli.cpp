// fx_multi_cli — the C++ multi-GPU driver (fx_multi.hpp) on synthetic VLP-16 scans: frame-shards a batch over the
// visible GPUs (or --devices N of them), gathers the keypoint records over RCCL, checks the gathered table against the
// per-rank results and prints scans/s.  On a 1-GPU box it runs with one rank: RCCL initialises and the collective runs.
//   fx_multi_cli [--devices N] [--batch B] [--steps K] [--launch]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "fx_multi.hpp"

int main(int argc, char **argv) {
  try {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
      std::fprintf(stderr, "fx_multi_cli: no HIP device (the hot path has no CPU fallback)\n");
      return 3;
    }
    int want = n_dev;
    uint32_t batch = 256, steps = 5;
    bool launch = true;
    for (int i = 1; i < argc; ++i) {
      if (!std::strcmp(argv[i], "--devices") && i + 1 < argc) want = std::atoi(argv[++i]);
      else if (!std::strcmp(argv[i], "--batch") && i + 1 < argc) batch = (uint32_t)std::atoi(argv[++i]);
      else if (!std::strcmp(argv[i], "--steps") && i + 1 < argc) steps = (uint32_t)std::atoi(argv[++i]);
      else if (!std::strcmp(argv[i], "--default")) launch = false;
    }
    if (want < 1 || want > n_dev) want = n_dev;
    std::vector<int> devices;
    for (int d = 0; d < want; ++d) devices.push_back(d);
    fx_params p;
    if (launch) fx_params_launch(&p); else fx_params_default(&p);
    fx_synth_cfg cfg;
    fx_synth_cfg_vlp16(&cfg, 0);
    const uint32_t N = cfg.n_rings * cfg.n_az;
    std::vector<float> host((size_t)batch * N * 4);
    std::vector<fx_scan_desc> scans(batch);
    for (uint32_t b = 0; b < batch; ++b) {
      fx_synth_cfg_vlp16(&cfg, 1000 + b);
      fx_synth_scan(&cfg, &host[(size_t)b * N * 4], N);
      scans[b] = fx_scan_desc{&host[(size_t)b * N * 4], N, 16, 0.02, -0.015};
    }
    fx::MultiGpu multi(p, devices, batch, N);
    std::printf("fx_multi_cli: %d device(s) visible, %u rank(s), %u scans per batch (%u per rank), RCCL communicators up\n", n_dev,
                multi.world(), batch, multi.scans_per_rank());
    std::vector<float> table;
    std::vector<fx_batch_view> views;
    multi.process(scans.data(), batch, FX_OUT_HOST, &table, &views);  // warm-up + the checked batch
    // the gathered table against every rank's own results
    uint64_t kp_total = 0;
    for (uint32_t b = 0; b < batch; ++b) {
      const uint32_t r = fx::owner_of(b, batch, multi.world());
      const uint32_t local = (uint32_t)(b - fx::shard_range(batch, multi.world(), r).first);
      const fx_batch_view &v = views[r];
      const fx::KeypointRecordView rec = multi.record(table, b, batch);
      const uint32_t K = v.h_n_keypoints[local];
      const uint32_t Kr = K < fx::kRecKeypoints ? K : fx::kRecKeypoints;
      if (rec.n_keypoints() != Kr) throw std::runtime_error("gathered keypoint count differs from the producing rank's");
      if (std::memcmp(rec.keypoint(0), v.h_keypoints + (size_t)local * v.max_keypoints * 4, (size_t)Kr * 16) != 0)
        throw std::runtime_error("gathered keypoints differ from the producing rank's");
      kp_total += K;
    }
    // every rank holds the same table
    std::vector<float> other(table.size());
    for (uint32_t r = 1; r < multi.world(); ++r) {
      if (hipSetDevice(devices[r]) != hipSuccess ||
          hipMemcpy(other.data(), multi.device_table(r), other.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        throw std::runtime_error("hipMemcpy of a rank's table failed");
      if (std::memcmp(other.data(), table.data(), table.size() * sizeof(float)) != 0)
        throw std::runtime_error("ranks hold different gathered tables");
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t s = 0; s < steps; ++s) multi.process(scans.data(), batch, 0, nullptr);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("fx_multi_cli: gathered table == per-rank results (%llu keypoints in %u scans); host-input batches: %.0f scans/s over %u rank(s)\n",
                (unsigned long long)kp_total, batch, (double)batch * steps / dt, multi.world());
    return 0;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "fx_multi_cli: %s\n", e.what());
    return 1;
  }
}
