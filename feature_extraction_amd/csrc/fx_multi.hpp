// fx_multi.hpp — C++ host driver that frame-shards a batch of scans over the GPUs of one node and gathers the
// keypoint records with one RCCL collective per batch (north star: "host side stays C++ ... scans are
// batched and frame-sharded across the 8 GPUs of one node with an RCCL gather of keypoints over xGMI").
//
// One process, one host thread + one fx_ctx + one stream per device.  The reference is a single-threaded ROS node with
// one scan in flight (ref: src/feature_extraction_node.cpp:386 ros::spin, :47 queue size 0); this is what replaces its
// cloudCallback loop when a node has several GPUs and the scans of several sensors / a recorded stream to chew through.
//   * scan b of a batch of B goes to rank floor(b G / B) (contiguous blocks, fx_shard.hpp): output order is trivial;
//   * every rank runs the unchanged single-GPU pipeline (fx_process_batch) on its block;
//   * fx_pack_keypoint_records writes the block's fixed-stride records, ncclAllGather assembles the table on every GPU
//     (2 KiB per scan: latency bound, not xGMI-bandwidth bound); descriptors stay on the producing GPU.
// Links against libfx_hip.so, librccl and libamdhip64.  No CPU fallback.
#ifndef FX_MULTI_HPP_
#define FX_MULTI_HPP_
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fx.h"
#include "fx_shard.hpp"

namespace fx {

class MultiGpu {
 public:
  // devices: HIP device ids, one rank each; max_batch: scans of one call over all devices together
  MultiGpu(const fx_params &params, const std::vector<int> &devices, uint32_t max_batch, uint32_t max_points,
           uint32_t rec_kp = kRecKeypoints)
      : devices_(devices), rec_kp_(rec_kp), max_batch_(max_batch) {
    const uint32_t G = (uint32_t)devices.size();
    if (!G) throw std::invalid_argument("fx::MultiGpu: no devices");
    per_rank_ = (max_batch + G - 1) / G;
    ranks_.resize(G);
    comms_.resize(G);
    // one communicator per device of this process (ncclCommInitAll: single-process, multi-device)
    nccl(ncclCommInitAll(comms_.data(), (int)G, devices_.data()), "ncclCommInitAll");
    for (uint32_t r = 0; r < G; ++r) {
      Rank &R = ranks_[r];
      hip(hipSetDevice(devices_[r]), "hipSetDevice");
      fx_limits lim;
      fx_limits_default(&lim, per_rank_, max_points);
      const fx_status st = fx_create(&params, &lim, devices_[r], &R.ctx);
      if (st != FX_OK) throw std::runtime_error(std::string("fx_create: ") + fx_last_error());
      hip(hipStreamCreateWithFlags(&R.stream, hipStreamNonBlocking), "hipStreamCreate");
      fx_set_stream(R.ctx, R.stream);
      hip(hipMalloc((void **)&R.rec, (size_t)per_rank_ * record_floats(rec_kp_) * sizeof(float)), "hipMalloc records");
      hip(hipMalloc((void **)&R.table, (size_t)per_rank_ * G * record_floats(rec_kp_) * sizeof(float)), "hipMalloc table");
      hip(hipMemset(R.rec, 0, (size_t)per_rank_ * record_floats(rec_kp_) * sizeof(float)), "hipMemset");
    }
  }
  ~MultiGpu() {
    for (size_t r = 0; r < ranks_.size(); ++r) {
      (void)hipSetDevice(devices_[r]);
      if (ranks_[r].ctx) fx_destroy(ranks_[r].ctx);
      if (ranks_[r].rec) (void)hipFree(ranks_[r].rec);
      if (ranks_[r].table) (void)hipFree(ranks_[r].table);
      if (ranks_[r].stream) (void)hipStreamDestroy(ranks_[r].stream);
      if (comms_[r]) (void)ncclCommDestroy(comms_[r]);
    }
  }
  MultiGpu(const MultiGpu &) = delete;
  MultiGpu &operator=(const MultiGpu &) = delete;

  uint32_t world() const { return (uint32_t)devices_.size(); }
  uint32_t scans_per_rank() const { return per_rank_; }

  // Runs the hot path on `batch` scans (host or device pointers per `flags`, as fx_process_batch takes them), block r
  // of the stream on device r, then gathers the keypoint records.  table_host (optional): the gathered table as
  // world * scans_per_rank records — rank r's block starts at record r * scans_per_rank; its first
  // shard_range(batch, world, r) records are live, the rest zero.  views (optional): every rank's fx_batch_view.
  void process(const fx_scan_desc *scans, uint32_t batch, uint32_t flags, std::vector<float> *table_host,
               std::vector<fx_batch_view> *views = nullptr) {
    if (batch > max_batch_) throw std::invalid_argument("fx::MultiGpu::process: batch > max_batch");
    const uint32_t G = world();
    std::vector<std::string> errors(G);
    std::vector<fx_batch_view> local(G);
    std::vector<std::thread> threads;
    for (uint32_t r = 0; r < G; ++r)
      threads.emplace_back([&, r]() {
        try {
          Rank &R = ranks_[r];
          hip(hipSetDevice(devices_[r]), "hipSetDevice");
          const auto span = shard_range(batch, G, r);
          const uint32_t n = (uint32_t)(span.second - span.first);
          if (fx_process_batch(R.ctx, scans + span.first, n, flags, &local[r]) != FX_OK)
            throw std::runtime_error(std::string("fx_process_batch: ") + fx_last_error());
          // the whole block is rewritten every batch: ranks with fewer scans than scans_per_rank leave zero records behind
          hip(hipMemsetAsync(R.rec, 0, (size_t)per_rank_ * record_floats(rec_kp_) * sizeof(float), R.stream), "hipMemsetAsync");
          if (n && fx_pack_keypoint_records(R.ctx, R.rec, rec_kp_) != FX_OK)
            throw std::runtime_error(std::string("fx_pack_keypoint_records: ") + fx_last_error());
          // the path's one collective
          nccl(ncclAllGather(R.rec, R.table, (size_t)per_rank_ * record_floats(rec_kp_), ncclFloat, comms_[r], R.stream),
               "ncclAllGather");
          hip(hipStreamSynchronize(R.stream), "hipStreamSynchronize");
        } catch (const std::exception &e) {
          errors[r] = e.what();
        }
      });
    for (auto &t : threads) t.join();
    for (const auto &e : errors)
      if (!e.empty()) throw std::runtime_error(e);
    if (views) *views = local;
    if (table_host) {
      table_host->resize((size_t)per_rank_ * G * record_floats(rec_kp_));
      hip(hipSetDevice(devices_[0]), "hipSetDevice");
      hip(hipMemcpy(table_host->data(), ranks_[0].table, table_host->size() * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy table");
    }
  }
  // record of stream position `scan` of the last batch of `batch` scans inside a gathered table
  KeypointRecordView record(const std::vector<float> &table, uint64_t scan, uint64_t batch) const {
    const uint32_t r = owner_of(scan, batch, world());
    const uint64_t local = scan - shard_range(batch, world(), r).first;
    return record_of(table.data(), (uint64_t)r * per_rank_ + local, rec_kp_);
  }
  // the gathered table as rank `r` holds it (device memory), for a check that every rank got the same bytes
  const float *device_table(uint32_t r) const { return ranks_[r].table; }

 private:
  struct Rank {
    fx_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    float *rec = nullptr, *table = nullptr;
  };
  static void hip(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
  }
  static void nccl(ncclResult_t e, const char *what) {
    if (e != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(e));
  }
  std::vector<int> devices_;
  std::vector<Rank> ranks_;
  std::vector<ncclComm_t> comms_;
  uint32_t rec_kp_, max_batch_, per_rank_ = 0;
};

}  // namespace fx
#endif
