// fx_batcher_cli — S simulated sensors (one producer thread each) push synthetic VLP-16 scans at HZ for SECONDS
// through fx::StreamBatcher (fx_batcher.hpp); every scan's keypoints and descriptors are written to OUT for the test
// to compare with the oracle, and the latency distribution / batch sizes are printed.
//   fx_batcher_cli [--sensors S] [--hz HZ] [--seconds T] [--burst N] [--out FILE] [--default] [--poles P] [--max-batch B]
// Scan q of sensor s is fx_synth_scan(seed 1000 + 1000 s + q), roll 0.02, pitch -0.015.
// OUT: per scan {u32 sensor, u32 seq, u32 flags, u32 K, K x float4 keypoints, K x 1989 float descriptors}; a scan of a batch
// that FAILED has flags = 0x80000000 | fx_status and K = 0.
//   fx_batcher_cli --files A,B,... [--node] [--big-limits] --out FILE
// pushes the scans of the given files (raw float32 x y z i records) ONE AT A TIME through one warm context — the batcher's,
// or with --node fx::FeatureExtractionNode::cloudCallback (launch preset, roll = pitch = 0) — as sensor 0, seq 0, 1, ...
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "fx_batcher.hpp"

int main(int argc, char **argv) {
  try {
    uint32_t sensors = 4, burst = 0, poles = 0, max_batch = 64;
    double hz = 10.0, seconds = 2.0;
    bool launch = true, node = false, big_limits = false;
    const char *out_path = nullptr, *files = nullptr;
    for (int i = 1; i < argc; ++i) {
      if (!std::strcmp(argv[i], "--sensors") && i + 1 < argc) sensors = (uint32_t)std::atoi(argv[++i]);
      else if (!std::strcmp(argv[i], "--hz") && i + 1 < argc) hz = std::atof(argv[++i]);
      else if (!std::strcmp(argv[i], "--seconds") && i + 1 < argc) seconds = std::atof(argv[++i]);
      else if (!std::strcmp(argv[i], "--burst") && i + 1 < argc) burst = (uint32_t)std::atoi(argv[++i]);
      else if (!std::strcmp(argv[i], "--out") && i + 1 < argc) out_path = argv[++i];
      else if (!std::strcmp(argv[i], "--default")) launch = false;
      else if (!std::strcmp(argv[i], "--poles") && i + 1 < argc) poles = (uint32_t)std::atoi(argv[++i]);
      else if (!std::strcmp(argv[i], "--max-batch") && i + 1 < argc) max_batch = (uint32_t)std::atoi(argv[++i]);
      else if (!std::strcmp(argv[i], "--files") && i + 1 < argc) files = argv[++i];
      else if (!std::strcmp(argv[i], "--node")) node = true;
      else if (!std::strcmp(argv[i], "--big-limits")) big_limits = true;
    }
    fx_params p;
    if (launch) fx_params_launch(&p); else fx_params_default(&p);
    if (files) {
      // ---- given scans, one at a time through one warm context
      std::vector<std::vector<float>> in;
      for (const char *q = files; *q;) {
        const char *e = std::strchr(q, ',');
        const std::string path = e ? std::string(q, e) : std::string(q);
        FILE *f = std::fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open " + path);
        std::fseek(f, 0, SEEK_END);
        const long bytes = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        std::vector<float> v((size_t)bytes / 4);
        if (bytes && std::fread(v.data(), 1, (size_t)bytes, f) != (size_t)bytes) throw std::runtime_error("short read of " + path);
        std::fclose(f);
        in.push_back(std::move(v));
        q = e ? e + 1 : q + std::strlen(q);
      }
      fx_limits big{};  // capacities of tests/test_gpu_front.py's scans beyond the LDS tiers
      big.max_ring_candidates = 2048, big.max_candidates = 4096, big.max_keypoints = 512, big.max_total_keypoints = 1024, big.max_kpc_points = 8192;
      FILE *out = out_path ? std::fopen(out_path, "wb") : nullptr;
      auto emit = [&](uint32_t seq, uint32_t flags, const fx::PointCloud &kp, const fx::DescriptorCloud &d) {
        if (!out) return;
        const uint32_t hdr[4] = {0u, seq, flags, (uint32_t)kp.size()};
        std::fwrite(hdr, 4, 4, out);
        std::fwrite(kp.data(), sizeof(fx::Point), kp.size(), out);
        std::fwrite(d.data(), sizeof(fx::Descriptor), d.size(), out);
      };
      uint32_t flags_or = 0;
      if (node) {
        fx::FeatureExtractionNode n(0, 28800);
        n.useLaunchPreset();
        if (big_limits) n.limitsOverride = big;
        for (size_t i = 0; i < in.size(); ++i) {
          fx::PointCloud full(in[i].size() / 4), cloud, kp, kpc;
          if (!full.empty()) std::memcpy(full.data(), in[i].data(), in[i].size() * 4);
          fx::DescriptorCloud d;
          n.cloudCallback(full, cloud, kp, kpc, d);
          flags_or |= n.lastFlags();
          emit((uint32_t)i, n.lastFlags(), kp, d);
        }
      } else {
        std::vector<fx::StreamBatcher::Result> results;
        fx::StreamBatcher batcher(p, 4, 28800, 0, [&](fx::StreamBatcher::Result &&r) { results.push_back(std::move(r)); }, 0, big_limits ? &big : nullptr);
        for (size_t i = 0; i < in.size(); ++i) {
          batcher.push(0, in[i].data(), (uint32_t)(in[i].size() / 4), 16, 0.0, 0.0);
          batcher.flush();  // (every scan is a batch of its own: the context is warm, the batch before was whatever came before)
        }
        for (size_t i = 0; i < results.size(); ++i) {
          flags_or |= results[i].flags;
          emit((uint32_t)results[i].id, results[i].status == FX_OK ? results[i].flags : (0x80000000u | (uint32_t)results[i].status), results[i].keypoints,
               results[i].descriptors);
        }
        if (results.size() != in.size()) throw std::runtime_error("scans lost");
      }
      if (out) std::fclose(out);
      std::printf("fx_batcher_cli: %zu given scans one at a time through %s, flags 0x%x\n", in.size(), node ? "fx::FeatureExtractionNode" : "fx::StreamBatcher", flags_or);
      return 0;
    }
    fx_synth_cfg cfg;
    fx_synth_cfg_vlp16(&cfg, 0);
    const uint32_t N = cfg.n_rings * cfg.n_az;
    const uint32_t per_sensor = burst ? burst : (uint32_t)(hz * seconds + 0.5);
    // the scans are generated up front: the producers only push
    std::vector<std::vector<float>> scans((size_t)sensors * per_sensor, std::vector<float>((size_t)N * 4));
    for (uint32_t s = 0; s < sensors; ++s)
      for (uint32_t q = 0; q < per_sensor; ++q) {
        fx_synth_cfg_vlp16(&cfg, 1000 + 1000ull * s + q);
        if (poles) cfg.n_poles = poles;
        fx_synth_scan(&cfg, scans[(size_t)s * per_sensor + q].data(), N);
      }
    std::mutex rm;
    std::vector<fx::StreamBatcher::Result> results;
    std::vector<uint32_t> seq_of;  // id -> sequence number within its sensor
    fx::StreamBatcher batcher(p, max_batch, N, 0, [&](fx::StreamBatcher::Result &&r) {
      std::lock_guard<std::mutex> lk(rm);
      results.push_back(std::move(r));
    });
    {  // one warm-up scan (first-use costs: code upload, graph capture), not counted
      batcher.push(0, scans[0].data(), N, 16, 0.02, -0.015);
      batcher.flush();
      std::lock_guard<std::mutex> lk(rm);
      results.clear();
    }
    std::mutex im;
    std::vector<std::pair<uint64_t, std::pair<uint32_t, uint32_t>>> ids;
    std::vector<std::thread> producers;
    const auto t_start = std::chrono::steady_clock::now();
    for (uint32_t s = 0; s < sensors; ++s)
      producers.emplace_back([&, s] {
        for (uint32_t q = 0; q < per_sensor; ++q) {
          if (!burst) {  // sensor s fires at phase s / S of the period
            const double t = (q + (double)s / sensors) / hz;
            std::this_thread::sleep_until(t_start + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(t)));
          }
          const uint64_t id = batcher.push(s, scans[(size_t)s * per_sensor + q].data(), N, 16, 0.02, -0.015);
          std::lock_guard<std::mutex> lk(im);
          ids.push_back({id, {s, q}});
        }
      });
    for (auto &t : producers) t.join();
    batcher.flush();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    const fx::StreamBatcher::Stats st = batcher.stats();
    std::sort(ids.begin(), ids.end());
    std::vector<double> lat;
    uint32_t flags_or = 0;
    FILE *out = out_path ? std::fopen(out_path, "wb") : nullptr;
    for (const auto &r : results) {
      lat.push_back(r.latency_ms);
      if (r.status == FX_OK) flags_or |= r.flags;
      const auto it = std::lower_bound(ids.begin(), ids.end(), std::make_pair(r.id, std::make_pair(0u, 0u)));
      if (it == ids.end() || it->first != r.id) throw std::runtime_error("result with an unknown id");
      if (out) {
        const uint32_t hdr[4] = {it->second.first, it->second.second, r.status == FX_OK ? r.flags : (0x80000000u | (uint32_t)r.status), (uint32_t)r.keypoints.size()};
        std::fwrite(hdr, 4, 4, out);
        std::fwrite(r.keypoints.data(), sizeof(fx::Point), r.keypoints.size(), out);
        std::fwrite(r.descriptors.data(), sizeof(fx::Descriptor), r.descriptors.size(), out);
      }
    }
    if (out) std::fclose(out);
    if (results.size() != (size_t)sensors * per_sensor) throw std::runtime_error("scans lost");
    std::sort(lat.begin(), lat.end());
    auto pct = [&](double q) { return lat[std::min(lat.size() - 1, (size_t)(q * lat.size()))]; };
    std::printf("fx_batcher_cli: %u sensors x %u scans (%s), %.2f s wall: %llu scans in %llu batches (largest %u), %llu failed, flags 0x%x; latency ms "
                "p50 %.3f p90 %.3f p99 %.3f max %.3f\n", sensors, per_sensor, burst ? "burst" : "paced", wall,
                (unsigned long long)(st.scans - 1), (unsigned long long)(st.batches - 1), st.largest_batch, (unsigned long long)st.failed_batches, flags_or,
                pct(0.50), pct(0.90), pct(0.99), lat.back());
    return 0;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "fx_batcher_cli: %s\n", e.what());
    return 1;
  }
}
