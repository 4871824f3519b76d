// fx_multi_cli — the C++ multi-GPU driver (fx_multi.hpp) on synthetic VLP-16 scans: frame-shards batches over the
// visible GPUs (or --devices N of them), gathers the keypoint records over RCCL, checks the gathered table against the
// per-rank results and prints scans/s with the inputs resident in device memory (--host-input: handed over as host
// buffers).  On a 1-GPU box it runs with one rank: RCCL initialises and the collective runs.
//   fx_multi_cli [--devices N] [--batch B] [--steps K] [--inflight F] [--default] [--host-input] [--selftest G] [--bad-scan I] [--root R]
// --root R: the table is gathered on rank R only (ncclGather) instead of on every rank (ncclAllGather).
// --selftest G: G ranks on device 0 with the collective replaced by a gather through host memory (fx::MultiGpuOptions::
// host_gather: RCCL refuses the same device twice) — worker threads, tickets, the error barrier, slots in flight and
// uneven blocks with G > 1 on a one-GPU box.  --bad-scan I: scan I of one submitted batch claims more points than the
// contexts hold: that batch must fail on every rank (no rank may wait in a collective for ever) and the next one be right.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>

#include "fx_multi.hpp"

int main(int argc, char **argv) {
  // every batch in flight wants a hardware queue of its own (the HIP runtime's default is 4 per device for all streams of
  // the process: more streams than that share queues and run one after the other); read when the runtime starts
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  try {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
      std::fprintf(stderr, "fx_multi_cli: no HIP device (the hot path has no CPU fallback)\n");
      return 3;
    }
    int want = n_dev;
    uint32_t batch = 256, steps = 20, in_flight = 4;
    bool launch = true, host_input = false;
    int selftest = 0, bad_scan = -1, root = -1;
    for (int i = 1; i < argc; ++i) {
      if (!std::strcmp(argv[i], "--devices") && i + 1 < argc) want = std::atoi(argv[++i]);
      else if (!std::strcmp(argv[i], "--batch") && i + 1 < argc) batch = (uint32_t)std::atoi(argv[++i]);
      else if (!std::strcmp(argv[i], "--steps") && i + 1 < argc) steps = (uint32_t)std::atoi(argv[++i]);
      else if (!std::strcmp(argv[i], "--inflight") && i + 1 < argc) in_flight = (uint32_t)std::atoi(argv[++i]);
      else if (!std::strcmp(argv[i], "--default")) launch = false;
      else if (!std::strcmp(argv[i], "--host-input")) host_input = true;
      else if (!std::strcmp(argv[i], "--selftest") && i + 1 < argc) selftest = std::atoi(argv[++i]);
      else if (!std::strcmp(argv[i], "--bad-scan") && i + 1 < argc) bad_scan = std::atoi(argv[++i]);
      else if (!std::strcmp(argv[i], "--root") && i + 1 < argc) root = std::atoi(argv[++i]);
    }
    if (want < 1 || want > n_dev) want = n_dev;
    std::vector<int> devices;
    for (int d = 0; d < want; ++d) devices.push_back(d);
    if (selftest > 0) devices.assign((size_t)selftest, 0);
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
    fx::MultiGpu::Options opt;
    opt.in_flight = in_flight;
    opt.sparse_limits = true;  // (the stream is synthetic VLP-16 scans)
    opt.host_gather = selftest > 0;
    opt.gather_root = root;
    fx::MultiGpu multi(p, devices, batch, N, opt);
    const uint32_t G = multi.world();
    std::printf("fx_multi_cli: %d device(s) visible, %u rank(s), %u scans per batch (%u per rank), %u batches in flight per device, "
                "compact keypoint blocks of %u keypoints (%.2f MB a rank), %s, %s\n", n_dev, G, batch, multi.scans_per_rank(), multi.in_flight(),
                multi.block_keypoints(), multi.block_floats() * 4.0 / 1e6, root >= 0 ? "gathered on one rank" : "gathered on every rank",
                selftest > 0 ? "SELF-TEST: all ranks on device 0, gather through host memory" : "RCCL communicators up");
    // ---- a checked batch (host input, results back on the host): the gathered table against every rank's own results
    std::vector<float> table;
    std::vector<fx_batch_view> views;
    multi.process(scans.data(), batch, FX_OUT_HOST, &table, &views);
    uint64_t kp_total = 0;
    for (uint32_t b = 0; b < batch; ++b) {
      const uint32_t r = fx::owner_of(b, batch, G);
      const uint32_t local = (uint32_t)(b - fx::shard_range(batch, G, r).first);
      const fx_batch_view &v = views[r];
      const fx::MultiGpu::ScanKeypoints rec = multi.keypoints(table, b, batch);
      const uint32_t K = v.h_n_keypoints[local];
      if (rec.n != K || rec.flags != v.h_flags[local])
        throw std::runtime_error("gathered keypoint count / flags differ from the producing rank's");
      if (std::memcmp(rec.kp, v.h_keypoints + (size_t)local * v.max_keypoints * 4, (size_t)K * 16) != 0)
        throw std::runtime_error("gathered keypoints differ from the producing rank's");
      kp_total += K;
    }
    {  // every rank holds the same table
      fx::MultiGpu::Ticket t = multi.submit(scans.data(), batch, 0);
      std::vector<float> t0;
      const fx::MultiGpu::Batch &res = t.wait(&t0);
      std::vector<float> other(t0.size());
      for (uint32_t r = 1; r < G && root < 0; ++r) {  // (gathered on one rank: only that one holds it)
        if (hipSetDevice(devices[r]) != hipSuccess ||
            hipMemcpy(other.data(), res.tables[r], other.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
          throw std::runtime_error("hipMemcpy of a rank's table failed");
        if (std::memcmp(other.data(), t0.data(), t0.size() * sizeof(float)) != 0) throw std::runtime_error("ranks hold different gathered tables");
      }
      if (std::memcmp(t0.data(), table.data(), table.size() * sizeof(float)) != 0)
        throw std::runtime_error("the same batch gave a different table the second time");
    }
    if (bad_scan >= 0 && (uint32_t)bad_scan < batch) {
      // ---- a batch one rank cannot run: every rank must come back with the error (the barrier before the collective), the
      //      ticket throws, and the next batch is right again
      std::vector<fx_scan_desc> bad = scans;
      bad[(size_t)bad_scan].n_points = N + 1;  // more points than the contexts hold: FX_ERR_TOO_LARGE on the owning rank
      bool threw = false;
      try {
        multi.submit(bad.data(), batch, 0).wait();
      } catch (const std::exception &e) {
        threw = true;
        std::printf("fx_multi_cli: the bad batch failed on every rank as it must: %s\n", e.what());
      }
      if (!threw) throw std::runtime_error("a batch with an oversized scan did not fail");
      std::vector<float> again;
      multi.process(scans.data(), batch, 0, &again);
      if (again.size() != table.size() || std::memcmp(again.data(), table.data(), table.size() * sizeof(float)) != 0)
        throw std::runtime_error("the batch after a failed one gave a different table");
      std::printf("fx_multi_cli: the batch after the failed one equals the reference table\n");
    }
    // ---- throughput: tickets kept in flight; inputs resident on the devices unless --host-input
    std::vector<void *> d_in(G, nullptr);
    std::vector<fx_scan_desc> dscans = scans;
    uint32_t flags = 0;
    if (!host_input) {
      flags = FX_IN_DEVICE;
      for (uint32_t r = 0; r < G; ++r) {
        const auto span = fx::shard_range(batch, G, r);
        const size_t n = (size_t)(span.second - span.first);
        if (!n) continue;
        if (hipSetDevice(devices[r]) != hipSuccess || hipMalloc(&d_in[r], n * N * 16) != hipSuccess ||  // (self-test: every block on device 0)
            hipMemcpy(d_in[r], &host[(size_t)span.first * N * 4], n * N * 16, hipMemcpyHostToDevice) != hipSuccess)
          throw std::runtime_error("upload of a rank's block failed");
        for (size_t i = 0; i < n; ++i) dscans[span.first + i].points = (const char *)d_in[r] + i * N * 16;
      }
    }
    std::deque<fx::MultiGpu::Ticket> pending;
    for (uint32_t s = 0; s < in_flight; ++s) multi.submit(dscans.data(), batch, flags).wait();  // warm-up
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t s = 0; s < steps; ++s) {
      if (pending.size() >= in_flight) {
        pending.front().wait();
        pending.pop_front();
      }
      pending.push_back(multi.submit(dscans.data(), batch, flags));
    }
    while (!pending.empty()) {
      pending.front().wait();
      pending.pop_front();
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (uint32_t r = 0; r < G; ++r)
      if (d_in[r]) {
        (void)hipSetDevice(devices[r]);
        (void)hipFree(d_in[r]);
      }
    std::printf("fx_multi_cli: gathered table == per-rank results (%llu keypoints in %u scans); %s batches, %u in flight: %.0f scans/s over %u rank(s)\n",
                (unsigned long long)kp_total, batch, host_input ? "host-input" : "device-resident", in_flight, (double)batch * steps / dt, G);
    return 0;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "fx_multi_cli: %s\n", e.what());
    return 1;
  }
}
