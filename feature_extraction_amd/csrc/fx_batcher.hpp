// fx_batcher.hpp — streaming front end for nodes that fan in several sensors.
//
// The reference node handles one scan per callback, single-threaded, with subscriber queues of size 0 = unbounded
// (ref: src/feature_extraction_node.cpp:47-48, 386): a scan that arrives while another is being processed waits in
// the queue, and N sensors cost N callbacks in a row.  On the GPU a batch of a few scans costs barely more than one
// scan (0.38 ms for one, 0.63 ms for eight), so the policy here is: never wait for a batch to fill —
// whenever the GPU is free, take WHATEVER HAS ARRIVED (up to max_batch scans) as one batch.  Under light load that is
// one scan at a time at the single-scan latency; when scans arrive faster than batches finish, the batches grow by
// themselves and the throughput follows.
//
// Threads: any number of producers call push() (the scan is copied: the caller's buffer is free on return); one
// consumer thread owns the fx_ctx (a context is not thread-safe), runs the batches and hands every scan's result to
// the callback, in arrival order.  No CPU fallback: construction fails without a GPU.
//
// A batch that FAILS (fx_process_batch returns an error) does not end the stream: its scans are delivered with the status
// and the error text and no results, the context repairs its own state on the next call (fx_ctx::state_suspect, csrc/
// fx_api.cpp) and the consumer goes on with whatever has arrived meanwhile.
#ifndef FX_BATCHER_HPP_
#define FX_BATCHER_HPP_
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fx.h"
#include "fx_node.hpp"

namespace fx {

class StreamBatcher {
 public:
  struct Result {
    uint64_t id = 0;        // what push() returned
    uint32_t sensor = 0;
    uint32_t flags = 0;     // FX_FLAG_* of the scan
    uint32_t batch = 0;     // scans in the batch it rode in
    fx_status status = FX_OK;  // of the batch: anything else means the batch failed and the clouds below are empty
    std::string error;         // fx_last_error() of the failed batch
    double latency_ms = 0;  // push() to callback
    PointCloud keypoints;         // ~keypoints (ref: node.cpp:129-131)
    DescriptorCloud descriptors;  // ~features payload (ref: node.cpp:113-124)
  };
  using Callback = std::function<void(Result &&)>;
  struct Stats {
    uint64_t scans = 0, batches = 0, failed_batches = 0;
    uint32_t largest_batch = 0;
  };

  // pool_keypoints: rows of the descriptor pool (7956 B of device memory each + as much pinned host memory + the row's
  // support list, 16 KB).  0 = every scan's keypoint capacity for batches of up to 64 scans (a one- or two-scan batch with
  // more keypoints than the default's average of 64 a scan must not flag FX_FLAG_TOTAL_KP_OVERFLOW), the library's
  // default (64 a scan) beyond: max_batch 1024 is then 65 536 rows = 1.6 GB, not 6 GB.
  // limits: non-zero fields (other than max_batch / max_points) override the preset's.
  // full_pools: NB the preset is fx_limits_SPARSE (since 0.6: streaming sensors are VLP-16 class — small dense-tier pools and
  // overflow regions, 4.5 GB a 1024-scan context instead of 7): a stream of DENSE scans (64 / 128 rings, rows of thousands of
  // support points) exhausts them, and the rows they cannot hold come back as NaN descriptors with FX_FLAG_NBR_OVERFLOW where
  // the reference computes them (ref: node.cpp:343-353).  full_pools = true starts from fx_limits_default instead
  // (ADVICE r5).
  StreamBatcher(const fx_params &params, uint32_t max_batch, uint32_t max_points, int device, Callback cb, uint32_t pool_keypoints = 0,
                const fx_limits *limits = nullptr, bool full_pools = false)
      : cb_(std::move(cb)), max_batch_(max_batch), max_points_(max_points) {
    if (FX_CHECK_ABI() != FX_OK) throw std::runtime_error(std::string("fx_check_abi: ") + fx_last_error());  // (this translation unit's fx.h against the library's)
    fx_limits lim;
    if (full_pools)
      fx_limits_default(&lim, max_batch, max_points);
    else
      fx_limits_sparse(&lim, max_batch, max_points);
    if (pool_keypoints)
      lim.max_total_keypoints = pool_keypoints;
    else if (max_batch <= 64u)
      lim.max_total_keypoints = max_batch * (limits && limits->max_keypoints ? limits->max_keypoints : lim.max_keypoints);
    if (limits) {
      const uint32_t *ov = reinterpret_cast<const uint32_t *>(limits);
      uint32_t *dst = reinterpret_cast<uint32_t *>(&lim);
      for (size_t i = 2; i < sizeof(fx_limits) / 4; ++i)
        if (ov[i]) dst[i] = ov[i];
    }
    if (fx_create(&params, &lim, device, &ctx_) != FX_OK) throw std::runtime_error(std::string("fx_create: ") + fx_last_error());
    // (no HIP-graph replay of the small batch sizes any more: since the launches behind the common kernels are one small
    //  workgroup each and the dense tier's four kernels one small launch, plain launches are as fast or faster — one scan per
    //  call 0.271 ms against 0.282 with the graph, two 0.301 / 0.316, eight and sixteen level: profiles/r05_experiments.md §10)
    estimate_descriptors_ = params.estimate_descriptors != 0;
    consumer_ = std::thread([this] { run(); });
  }
  ~StreamBatcher() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    if (consumer_.joinable()) consumer_.join();
    if (ctx_) fx_destroy(ctx_);
  }
  StreamBatcher(const StreamBatcher &) = delete;
  StreamBatcher &operator=(const StreamBatcher &) = delete;

  // One scan of one sensor: n_points records of stride_bytes (16: packed x y z i; 32: pcl::PointXYZI in memory), with the
  // attitude that goes with it (ref: node.h:116).  Thread-safe; returns the scan's id.
  uint64_t push(uint32_t sensor, const void *points, uint32_t n_points, uint32_t stride_bytes, double roll, double pitch) {
    if (n_points > max_points_) throw std::invalid_argument("fx::StreamBatcher::push: scan larger than max_points");
    if (stride_bytes < 16 || stride_bytes % 16) throw std::invalid_argument("fx::StreamBatcher::push: stride_bytes must be a multiple of 16");
    Pending p;
    p.sensor = sensor;
    p.n = n_points;
    p.roll = roll, p.pitch = pitch;
    p.xyzi.resize((size_t)n_points * 4);
    const uint8_t *src = static_cast<const uint8_t *>(points);
    if (stride_bytes == 16) {
      if (n_points) std::memcpy(p.xyzi.data(), src, (size_t)n_points * 16);
    } else {
      for (uint32_t i = 0; i < n_points; ++i) std::memcpy(&p.xyzi[(size_t)i * 4], src + (size_t)i * stride_bytes, 16);
    }
    p.t0 = std::chrono::steady_clock::now();
    uint64_t id;
    {
      std::lock_guard<std::mutex> lk(m_);
      if (!error_.empty()) throw std::runtime_error(error_);
      id = p.id = next_id_++;
      queue_.push_back(std::move(p));
    }
    cv_.notify_all();
    return id;
  }
  // Returns when every scan pushed so far has been delivered (or throws what stopped the consumer).
  void flush() {
    std::unique_lock<std::mutex> lk(m_);
    const uint64_t upto = next_id_;
    cv_.wait(lk, [&] { return delivered_ >= upto || !error_.empty(); });
    if (!error_.empty()) throw std::runtime_error(error_);
  }
  Stats stats() const {
    std::lock_guard<std::mutex> lk(m_);
    return stats_;
  }

 private:
  struct Pending {
    uint64_t id = 0;
    uint32_t sensor = 0, n = 0;
    double roll = 0, pitch = 0;
    std::vector<float> xyzi;
    std::chrono::steady_clock::time_point t0;
  };
  void run() {
    std::vector<Pending> batch;
    std::vector<fx_scan_desc> descs;
    while (true) {
      batch.clear();
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
        if (queue_.empty()) return;  // (stop, and everything delivered)
        // whatever has arrived, up to the context's batch capacity; nothing is waited for
        while (!queue_.empty() && batch.size() < max_batch_) {
          batch.push_back(std::move(queue_.front()));
          queue_.pop_front();
        }
      }
      descs.resize(batch.size());
      for (size_t i = 0; i < batch.size(); ++i)
        descs[i] = fx_scan_desc{batch[i].xyzi.data(), batch[i].n, 16, batch[i].roll, batch[i].pitch};
      fx_batch_view v;
      const fx_status st = fx_process_batch(ctx_, descs.data(), (uint32_t)batch.size(), FX_OUT_HOST, &v);
      if (st != FX_OK) {
        // the batch's scans come back with the status and no results; the context starts its next batch from scratch
        // (state_suspect) and the stream goes on
        const std::string err = std::string("fx_process_batch: ") + fx_last_error();
        for (size_t i = 0; i < batch.size(); ++i) {
          Result r;
          r.id = batch[i].id;
          r.sensor = batch[i].sensor;
          r.batch = (uint32_t)batch.size();
          r.status = st;
          r.error = err;
          r.latency_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - batch[i].t0).count();
          cb_(std::move(r));
        }
        {
          std::lock_guard<std::mutex> lk(m_);
          delivered_ += batch.size();
          stats_.scans += batch.size();
          stats_.batches += 1;
          stats_.failed_batches += 1;
        }
        cv_.notify_all();
        continue;
      }
      for (size_t i = 0; i < batch.size(); ++i) {
        Result r;
        r.id = batch[i].id;
        r.sensor = batch[i].sensor;
        r.flags = v.h_flags[i];
        r.batch = (uint32_t)batch.size();
        const uint32_t K = v.h_n_keypoints[i];
        r.keypoints.resize(K);
        if (K) std::memcpy(r.keypoints.data(), v.h_keypoints + (size_t)i * v.max_keypoints * 4, (size_t)K * sizeof(Point));
        if (estimate_descriptors_ && K) {
          // rows of this scan the pool holds (all of them unless FX_FLAG_TOTAL_KP_OVERFLOW is set, which r.flags reports:
          // kp_offset and n_keypoints are not clamped to the pool, the copied rows are)
          const uint32_t off = v.h_kp_offset[i] < v.total_keypoints ? v.h_kp_offset[i] : v.total_keypoints;
          const uint32_t rows = K < v.total_keypoints - off ? K : v.total_keypoints - off;
          r.descriptors.resize(rows);
          if (rows) std::memcpy(r.descriptors.data(), v.h_descriptors + (size_t)off * FX_DESC_FLOATS, (size_t)rows * sizeof(Descriptor));
        }
        r.latency_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - batch[i].t0).count();
        cb_(std::move(r));
      }
      {
        std::lock_guard<std::mutex> lk(m_);
        delivered_ += batch.size();
        stats_.scans += batch.size();
        stats_.batches += 1;
        if (batch.size() > stats_.largest_batch) stats_.largest_batch = (uint32_t)batch.size();
      }
      cv_.notify_all();
    }
  }

  Callback cb_;
  uint32_t max_batch_, max_points_;
  bool estimate_descriptors_ = true;
  fx_ctx *ctx_ = nullptr;
  mutable std::mutex m_;
  std::condition_variable cv_;
  std::deque<Pending> queue_;
  std::thread consumer_;
  bool stop_ = false;
  uint64_t next_id_ = 0, delivered_ = 0;
  std::string error_;
  Stats stats_;
};

}  // namespace fx
#endif
