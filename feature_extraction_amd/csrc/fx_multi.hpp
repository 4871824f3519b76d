// fx_multi.hpp — C++ host driver that frame-shards batches of scans over the GPUs of one node and gathers the
// keypoint records with one RCCL collective per batch (north star: "host side stays C++ ... scans are
// batched and frame-sharded across the 8 GPUs of one node with an RCCL gather of keypoints over xGMI").
//
// The reference is a single-threaded ROS node with one scan in flight (ref: src/feature_extraction_node.cpp:386
// ros::spin, :47 queue size 0); this is what replaces its cloudCallback loop when a node has several GPUs and the
// scans of several sensors / a recorded stream to chew through.
//   * scan b of a batch of B goes to rank floor(b G / B) (contiguous blocks, fx_shard.hpp): output order is trivial;
//   * every rank runs the unchanged single-GPU pipeline (fx_process_batch) on its block;
//   * fx_pack_keypoint_block writes the block's keypoints as ONE compact block (offsets + flags + the keypoints packed in scan
//     order, 64 a scan of capacity by default: 1.05 MB a rank for 1024 scans where max-stride records were 4.2 MB; what does
//     not fit is flagged), ncclAllGather assembles the table on every GPU — or, Options::gather_root, ncclGather on one: a
//     publisher needs one copy —; descriptors stay on the producing GPU.
// Threads: ONE persistent worker per device, fed through a queue — no thread is created per batch.  Every device has
// `in_flight` contexts, each on its own stream: submit() returns at once with a ticket, batch t runs on context
// t % in_flight, and nothing in the worker waits for the GPU except the reuse of a context (the stage kernels of one
// batch leave most of the GPU idle; the next batch fills it).  Ticket::wait() is where the caller synchronises.
// Errors: a rank that fails before the collective would leave the others waiting inside it for ever, so the workers
// meet at a barrier after the local part of a batch and issue the collective only when every rank got that far;
// otherwise the batch fails as a whole and the error is reported by wait().
// Links against libfx_hip.so, librccl and libamdhip64.  No CPU fallback.
#ifndef FX_MULTI_HPP_
#define FX_MULTI_HPP_
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fx.h"
#include "fx_shard.hpp"

namespace fx {

struct MultiGpuOptions {
  uint32_t in_flight = 2;  // contexts (batches in flight) per device (2: 1.59 million scans/s on one MI355X, 1: 1.24, 3: 1.48 — the shared communicator orders the slots)
  uint32_t block_keypoints = 0;  // keypoint rows of a rank's gathered block; 0 = 64 a scan of the rank's share (VLP-16 scenes have 54; a batch with more is cut and flagged FX_FLAG_KP_OVERFLOW)
  int gather_root = -1;    // >= 0: the table is gathered on that rank only (ncclGather; the other ranks' tables stay untouched); -1: on every rank (ncclAllGather)
  fx_limits limits{};      // non-zero fields override fx_limits_default(scans per rank, max_points)
  // SELF-TEST: the ranks' collective is replaced by a gather through host memory (every rank downloads its block, all ranks
  // meet, every rank uploads the whole table) and no RCCL communicator is made — RCCL refuses a communicator with the same
  // device twice, so this is how G > 1 ranks (worker threads, tickets, the error barrier, slots in flight, uneven blocks)
  // run on the ONE GPU a test box has: devices = {0, 0, ..., 0}.  Never the product path.
  bool host_gather = false;
  bool sparse_limits = false;  // start from fx_limits_sparse instead (VLP-16-class streams: small dense-tier pools — 4.5 GB a 1024-scan context, not 7)
};
// What one batch produced.  Valid until `in_flight` more batches have been submitted (the contexts own the memory).
struct MultiGpuBatch {
  uint32_t batch = 0;
  std::vector<fx_batch_view> views;    // per rank: that rank's block
  std::vector<const float *> tables;   // per rank: the gathered table on that device (world blocks of block_floats() floats; with gather_root: valid on that rank only)
  std::string error;                   // empty = ok
};
class MultiGpu;
struct MultiGpuJob {
  MultiGpu *owner = nullptr;
  const fx_scan_desc *scans = nullptr;
  uint32_t flags = 0, slot = 0;
  MultiGpuBatch out;
  std::vector<std::string> local_error;  // per rank: what failed before the collective
  std::mutex m;
  std::condition_variable cv;
  uint32_t arrived = 0;    // ranks that finished the local part (the barrier before the collective)
  uint32_t enqueued = 0;   // ranks that finished enqueuing (or gave up): wait() may look at the events
  uint32_t gathered = 0;   // (host_gather) ranks whose block is in host_table
  std::vector<float> host_table;  // (host_gather) the table on its way through host memory
};


class MultiGpu {
 public:
  using Options = MultiGpuOptions;
  using Batch = MultiGpuBatch;
  class Ticket {
   public:
    Ticket() = default;
    // Waits for the batch on every device; throws what went wrong.  table_host (optional): rank 0's gathered table.
    const Batch &wait(std::vector<float> *table_host = nullptr);

   private:
    friend class MultiGpu;
    std::shared_ptr<MultiGpuJob> job_;
  };

  // devices: HIP device ids, one rank each; max_batch: scans of one batch over all devices together
  MultiGpu(const fx_params &params, const std::vector<int> &devices, uint32_t max_batch, uint32_t max_points, const Options &opt = Options())
      : devices_(devices), max_batch_(max_batch), in_flight_(opt.in_flight ? opt.in_flight : 1u), host_gather_(opt.host_gather), gather_root_(opt.gather_root) {
    const uint32_t G = (uint32_t)devices.size();
    if (!G) throw std::invalid_argument("fx::MultiGpu: no devices");
    if (FX_CHECK_ABI() != FX_OK) throw std::runtime_error(std::string("fx_check_abi: ") + fx_last_error());  // (this translation unit's fx.h against the library's)
    per_rank_ = (max_batch + G - 1) / G;
    if (gather_root_ >= (int)G) throw std::invalid_argument("fx::MultiGpu: gather_root is not a rank");
    block_kp_ = opt.block_keypoints ? opt.block_keypoints : 64u * per_rank_;
    for (uint32_t r = 0; r < G; ++r) ranks_.emplace_back(new Rank());
    comms_.assign(G, nullptr);
    try {
      // one communicator per device of this process (ncclCommInitAll: single-process, multi-device); the slots of a device
      // share it and issue their collectives in ticket order, which is RCCL's ordering contract
      if (!host_gather_) nccl(ncclCommInitAll(comms_.data(), (int)G, devices_.data()), "ncclCommInitAll");
      for (uint32_t r = 0; r < G; ++r) {
        Rank &R = *ranks_[r];
        hip(hipSetDevice(devices_[r]), "hipSetDevice");
        R.slots.resize(in_flight_);
        for (Slot &S : R.slots) {
          fx_limits lim;
          if (opt.sparse_limits)
            fx_limits_sparse(&lim, per_rank_, max_points);
          else
            fx_limits_default(&lim, per_rank_, max_points);
          const uint32_t *ov = reinterpret_cast<const uint32_t *>(&opt.limits);
          uint32_t *dst = reinterpret_cast<uint32_t *>(&lim);
          for (size_t i = 2; i < sizeof(fx_limits) / 4; ++i)  // (max_batch / max_points are the constructor's)
            if (ov[i]) dst[i] = ov[i];
          if (fx_create(&params, &lim, devices_[r], &S.ctx) != FX_OK) throw std::runtime_error(std::string("fx_create: ") + fx_last_error());
          fx_set_batches_in_flight(S.ctx, in_flight_);  // (launch-policy hint: the slots of a device share its chip)
          hip(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking), "hipStreamCreate");
          fx_set_stream(S.ctx, S.stream);
          hip(hipEventCreateWithFlags(&S.done, hipEventDisableTiming), "hipEventCreate");
          const size_t rec_bytes = block_floats() * sizeof(float);
          hip(hipMalloc((void **)&S.rec, rec_bytes), "hipMalloc records");
          hip(hipMalloc((void **)&S.table, rec_bytes * G), "hipMalloc table");
          hip(hipMemset(S.rec, 0, rec_bytes), "hipMemset");
        }
      }
      for (uint32_t r = 0; r < G; ++r) ranks_[r]->worker = std::thread([this, r] { work(r); });
    } catch (...) {
      shutdown();
      throw;
    }
  }
  ~MultiGpu() { shutdown(); }
  MultiGpu(const MultiGpu &) = delete;
  MultiGpu &operator=(const MultiGpu &) = delete;

  uint32_t world() const { return (uint32_t)devices_.size(); }
  uint32_t scans_per_rank() const { return per_rank_; }
  uint32_t in_flight() const { return in_flight_; }
  uint32_t block_keypoints() const { return block_kp_; }                                // keypoint rows of a rank's block
  size_t block_floats() const { return fx::block_floats(per_rank_, block_kp_); }         // floats of a rank's block: the collective's count
  int gather_root() const { return gather_root_; }

  // Enqueues the hot path on `batch` scans (host or device pointers per `flags`, as fx_process_batch takes them; block r
  // of the batch must be readable by device r) and the gather of the keypoint records.  Returns at once; `scans` and
  // what they point to must stay valid until the ticket has been waited for.  Call from one thread.
  Ticket submit(const fx_scan_desc *scans, uint32_t batch, uint32_t flags) {
    if (batch > max_batch_) throw std::invalid_argument("fx::MultiGpu::submit: batch > max_batch");
    auto job = std::make_shared<MultiGpuJob>();
    job->owner = this;
    job->scans = scans;
    job->flags = flags;
    job->slot = (uint32_t)(next_ticket_++ % in_flight_);
    job->out.batch = batch;
    job->out.views.resize(world());
    job->out.tables.resize(world());
    job->local_error.resize(world());
    if (host_gather_) job->host_table.assign(block_floats() * world(), 0.0f);
    for (auto &rp : ranks_) {
      Rank &R = *rp;
      {
        std::lock_guard<std::mutex> lk(R.m);
        R.queue.push_back(job);
      }
      R.cv.notify_one();
    }
    Ticket t;
    t.job_ = job;
    return t;
  }
  // submit + wait: one batch at a time (table_host: rank 0's gathered table; views: every rank's fx_batch_view)
  void process(const fx_scan_desc *scans, uint32_t batch, uint32_t flags, std::vector<float> *table_host,
               std::vector<fx_batch_view> *views = nullptr) {
    Ticket t = submit(scans, batch, flags);
    const Batch &b = t.wait(table_host);
    if (views) *views = b.views;
  }
  // the keypoints of stream position `scan` of a batch of `batch` scans inside a gathered table: its owner's block and its
  // place in it
  struct ScanKeypoints {
    uint32_t n, flags;
    const float *kp;  // n x (x, y, z, elevation)
  };
  KeypointBlockView block(const std::vector<float> &table, uint32_t rank) const { return {table.data() + (size_t)rank * block_floats(), per_rank_}; }
  ScanKeypoints keypoints(const std::vector<float> &table, uint64_t scan, uint64_t batch) const {
    const uint32_t r = owner_of(scan, batch, world());
    const uint32_t local = (uint32_t)(scan - shard_range(batch, world(), r).first);
    const KeypointBlockView v = block(table, r);
    return {v.n_keypoints(local), v.flags(local), v.keypoint(local, 0)};
  }

 private:
  struct Slot {
    fx_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    float *rec = nullptr, *table = nullptr;
    bool used = false;
  };
  struct Rank {
    std::vector<Slot> slots;
    std::thread worker;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::shared_ptr<MultiGpuJob>> queue;
    bool stop = false;
  };
  static void hip(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
  }
  static void nccl(ncclResult_t e, const char *what) {
    if (e != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(e));
  }

  // the worker of rank r: jobs in ticket order (the same order on every rank: the collectives match up)
  void work(uint32_t r);
  void shutdown() {
    for (auto &rp : ranks_) {
      Rank &R = *rp;
      {
        std::lock_guard<std::mutex> lk(R.m);
        R.stop = true;
      }
      R.cv.notify_one();
    }
    for (auto &rp : ranks_)
      if (rp->worker.joinable()) rp->worker.join();
    for (size_t r = 0; r < ranks_.size(); ++r) {
      (void)hipSetDevice(devices_[r]);
      for (Slot &S : ranks_[r]->slots) {
        if (S.stream) (void)hipStreamSynchronize(S.stream);
        if (S.ctx) fx_destroy(S.ctx);
        if (S.rec) (void)hipFree(S.rec);
        if (S.table) (void)hipFree(S.table);
        if (S.done) (void)hipEventDestroy(S.done);
        if (S.stream) (void)hipStreamDestroy(S.stream);
        S = Slot();
      }
      if (r < comms_.size() && comms_[r]) {
        (void)ncclCommDestroy(comms_[r]);
        comms_[r] = nullptr;
      }
    }
  }

  std::vector<int> devices_;
  std::vector<std::unique_ptr<Rank>> ranks_;  // (a Rank holds a mutex: it never moves)
  std::vector<ncclComm_t> comms_;
  uint32_t block_kp_ = 0, max_batch_, per_rank_ = 0, in_flight_;
  bool host_gather_ = false;
  int gather_root_ = -1;
  uint64_t next_ticket_ = 0;
};

inline const MultiGpu::Batch &MultiGpu::Ticket::wait(std::vector<float> *table_host) {
    if (!job_) throw std::logic_error("fx::MultiGpu::Ticket: empty");
    {
      std::unique_lock<std::mutex> lk(job_->m);
      job_->cv.wait(lk, [&] { return job_->enqueued == job_->owner->world(); });
    }
    if (!job_->out.error.empty()) throw std::runtime_error(job_->out.error);
    for (uint32_t r = 0; r < job_->owner->world(); ++r) {
      hip(hipSetDevice(job_->owner->devices_[r]), "hipSetDevice");
      hip(hipEventSynchronize(job_->owner->ranks_[r]->slots[job_->slot].done), "hipEventSynchronize");
    }
    if (table_host) {
      const MultiGpu &o = *job_->owner;
      const uint32_t from = o.gather_root_ >= 0 ? (uint32_t)o.gather_root_ : 0u;  // (a rank that holds the table)
      table_host->resize(o.block_floats() * o.world());
      hip(hipSetDevice(o.devices_[from]), "hipSetDevice");
      hip(hipMemcpy(table_host->data(), job_->out.tables[from], table_host->size() * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy table");
    }
    return job_->out;
  }

inline void MultiGpu::work(uint32_t r) {
  Rank &R = *ranks_[r];
  const uint32_t G = world();
  (void)hipSetDevice(devices_[r]);
  while (true) {
    std::shared_ptr<MultiGpuJob> job;
    {
      std::unique_lock<std::mutex> lk(R.m);
      R.cv.wait(lk, [&] { return R.stop || !R.queue.empty(); });
      if (R.queue.empty()) return;  // (stop, and nothing left to run)
      job = R.queue.front();
      R.queue.pop_front();
    }
    Slot &S = R.slots[job->slot];
    // ---- the local part: this rank's block through the unchanged single-GPU pipeline, then its records
    try {
      if (S.used) hip(hipEventSynchronize(S.done), "hipEventSynchronize");  // the context's previous batch (in_flight tickets ago)
      const auto span = shard_range(job->out.batch, G, r);
      const uint32_t n = (uint32_t)(span.second - span.first);
      if (fx_process_batch(S.ctx, job->scans + span.first, n, job->flags, &job->out.views[r]) != FX_OK)
        throw std::runtime_error(std::string("fx_process_batch: ") + fx_last_error());
      // (the whole block is rewritten every batch — header, offsets, flags, keypoints, the zero tail — whatever the rank's share)
      if (fx_pack_keypoint_block(S.ctx, S.rec, per_rank_, block_kp_) != FX_OK)
        throw std::runtime_error(std::string("fx_pack_keypoint_block: ") + fx_last_error());
    } catch (const std::exception &e) {
      job->local_error[r] = std::string("rank ") + std::to_string(r) + ": " + e.what();
    }
    // ---- every rank got this far?  (a rank missing from the collective would hang the others inside it)
    bool all_ok = true;
    {
      std::unique_lock<std::mutex> lk(job->m);
      ++job->arrived;
      job->cv.notify_all();
      job->cv.wait(lk, [&] { return job->arrived == G; });
      for (const std::string &e : job->local_error)
        if (!e.empty()) {
          all_ok = false;
          if (job->out.error.empty()) job->out.error = e;
        }
    }
    std::string late;
    if (all_ok && host_gather_) {
      // (self-test: the all-gather through host memory — every rank has passed the error barrier, so every rank gets here)
      const size_t block = block_floats();
      hipError_t ge = hipMemcpyAsync(job->host_table.data() + (size_t)r * block, S.rec, block * sizeof(float), hipMemcpyDeviceToHost, S.stream);
      if (ge == hipSuccess) ge = hipStreamSynchronize(S.stream);
      {
        std::unique_lock<std::mutex> lk(job->m);
        ++job->gathered;
        job->cv.notify_all();
        job->cv.wait(lk, [&] { return job->gathered == G; });
      }
      if (ge == hipSuccess) ge = hipMemcpyAsync(S.table, job->host_table.data(), block * G * sizeof(float), hipMemcpyHostToDevice, S.stream);
      if (ge == hipSuccess) ge = hipStreamSynchronize(S.stream);  // (the job's host table may go away with the last ticket)
      if (ge != hipSuccess) late = std::string("rank ") + std::to_string(r) + ": host gather: " + hipGetErrorString(ge);
    } else if (all_ok) {
      // the path's one collective, an ordinary kernel of the slot's stream
      const ncclResult_t ce = gather_root_ >= 0 ? ncclGather(S.rec, S.table, block_floats(), ncclFloat, gather_root_, comms_[r], S.stream)
                                                : ncclAllGather(S.rec, S.table, block_floats(), ncclFloat, comms_[r], S.stream);
      if (ce != ncclSuccess) late = std::string("rank ") + std::to_string(r) + ": RCCL gather: " + ncclGetErrorString(ce);
    }
    const hipError_t he = hipEventRecord(S.done, S.stream);
    if (he != hipSuccess && late.empty()) late = std::string("rank ") + std::to_string(r) + ": hipEventRecord: " + hipGetErrorString(he);
    S.used = true;
    job->out.tables[r] = S.table;
    {
      std::lock_guard<std::mutex> lk(job->m);
      if (!late.empty() && job->out.error.empty()) job->out.error = late;
      ++job->enqueued;
    }
    job->cv.notify_all();
  }
}

}  // namespace fx
#endif
