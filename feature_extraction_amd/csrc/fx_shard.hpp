// fx_shard.hpp — frame sharding of a scan stream over the GPUs of one node, and the fixed-stride keypoint records
// that cross GPUs.  Plain C++, no HIP: the same plan and record layout as
// feature_extraction_amd/sharding.py (bench.py, the gloo test) and fx_pack_keypoint_records (the device writer).
//
// Scans are independent — the reference keeps no state across scans except roll/pitch, which are per-scan inputs
// (ref: include/feature_extraction/feature_extraction_node.h:116) — so rank r owns a contiguous block of the stream
// and runs the unchanged single-GPU pipeline on it; the only exchange is the keypoint table.
#ifndef FX_SHARD_HPP_
#define FX_SHARD_HPP_
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

namespace fx {

constexpr uint32_t kRecKeypoints = 127;  // 1 header + 127 keypoints = 2 KiB per scan

// contiguous block [first, second) of a stream of `total` scans owned by `rank` of `world`
inline std::pair<uint64_t, uint64_t> shard_range(uint64_t total, uint32_t world, uint32_t rank) {
  return {total * rank / world, total * (rank + 1) / world};
}
// rank that owns stream position `scan`
inline uint32_t owner_of(uint64_t scan, uint64_t total, uint32_t world) {
  uint32_t r = total ? (uint32_t)((scan * world) / total) : 0u;
  while (r + 1 < world && scan >= shard_range(total, world, r).second) ++r;
  while (r > 0 && scan < shard_range(total, world, r).first) --r;
  return r;
}

// Records every rank contributes to the all-gather of a stream of `total` scans: the plan's largest block (a collective
// takes equal contributions; a rank whose block is a scan shorter pads it with an empty record), and the row of stream
// position `scan` in the gathered table (rank blocks of block_size records one after the other).
inline uint64_t block_size(uint64_t total, uint32_t world) {
  uint64_t b = 0;
  for (uint32_t r = 0; r < world; ++r) {
    const auto s = shard_range(total, world, r);
    b = s.second - s.first > b ? s.second - s.first : b;
  }
  return b;
}
inline uint64_t table_row(uint64_t scan, uint64_t total, uint32_t world) {
  const uint32_t o = owner_of(scan, total, world);
  return (uint64_t)o * block_size(total, world) + (scan - shard_range(total, world, o).first);
}

// One scan's record: (1 + rec_kp) float4 = header {n_kp, flags, 0, 0} as uint32, then rec_kp (x, y, z, elevation)
// entries, zero padded.
struct KeypointRecordView {
  const float *rec;  // (1 + rec_kp) * 4 floats
  uint32_t rec_kp;
  uint32_t n_keypoints() const {
    uint32_t v;
    std::memcpy(&v, rec, 4);
    return v;
  }
  uint32_t flags() const {
    uint32_t v;
    std::memcpy(&v, rec + 1, 4);
    return v;
  }
  const float *keypoint(uint32_t k) const { return rec + 4 * (1 + k); }
};
inline size_t record_floats(uint32_t rec_kp) { return (size_t)(1 + rec_kp) * 4; }
inline KeypointRecordView record_of(const float *table, uint64_t scan, uint32_t rec_kp = kRecKeypoints) {
  return {table + scan * record_floats(rec_kp), rec_kp};
}
// host statement of fx_pack_keypoint_records for one scan (tests; a CPU producer)
inline void pack_record(float *dst, const float *keypoints_xyzi, uint32_t n_kp, uint32_t flags, uint32_t rec_kp = kRecKeypoints) {
  std::memset(dst, 0, record_floats(rec_kp) * sizeof(float));
  const uint32_t k = n_kp < rec_kp ? n_kp : rec_kp;
  const uint32_t f = flags | (n_kp > rec_kp ? 0x4u /* FX_FLAG_KP_OVERFLOW */ : 0u);
  std::memcpy(dst, &k, 4);
  std::memcpy(dst + 1, &f, 4);
  if (k) std::memcpy(dst + 4, keypoints_xyzi, (size_t)k * 16);
}

}  // namespace fx
#endif
