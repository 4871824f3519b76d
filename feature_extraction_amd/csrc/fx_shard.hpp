// fx_shard.hpp — frame sharding of a scan stream over the GPUs of one node, and the keypoint tables that cross GPUs
// (SURVEY.md 8e): the compact keypoint block (since 0.7: what fx::MultiGpu and bench.py gather) and the fixed-stride records.  Plain C++, no HIP: the same plan and record layout as
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


// ---- the compact keypoint block (fx_pack_keypoint_block, include/fx.h): what crosses GPUs since 0.7.  One block per rank and
// batch, rows of four floats: row 0 {scans, keypoints stored, OR of the flags, max_total} (u32), kp_offset[max_scans + 1] (u32,
// four a row), flags[max_scans] (u32, four a row), then max_total keypoint rows (x, y, z, elevation) packed in scan order.
inline size_t block_off_rows(uint32_t max_scans) { return ((size_t)max_scans + 1 + 3) / 4; }
inline size_t block_flag_rows(uint32_t max_scans) { return ((size_t)max_scans + 3) / 4; }
inline size_t block_floats(uint32_t max_scans, uint32_t max_total) {
  return (1 + block_off_rows(max_scans) + block_flag_rows(max_scans) + (size_t)max_total) * 4;
}
struct KeypointBlockView {
  const float *blk;
  uint32_t max_scans;
  uint32_t word(size_t i) const {
    uint32_t v;
    std::memcpy(&v, blk + i, 4);
    return v;
  }
  uint32_t scans() const { return word(0); }
  uint32_t keypoints_stored() const { return word(1); }
  uint32_t flags_or() const { return word(2); }
  uint32_t max_total() const { return word(3); }
  uint32_t offset(uint32_t b) const { return word(4 + b); }
  uint32_t n_keypoints(uint32_t b) const { return offset(b + 1) - offset(b); }
  uint32_t flags(uint32_t b) const { return word(4 + 4 * block_off_rows(max_scans) + b); }
  const float *keypoint(uint32_t b, uint32_t k) const {
    return blk + 4 * (1 + block_off_rows(max_scans) + block_flag_rows(max_scans) + (size_t)offset(b) + k);
  }
};
// host statement of fx_pack_keypoint_block (tests; a CPU producer): scan b has n_kp[b] keypoints at kp[b]
inline void pack_block(float *dst, const std::vector<const float *> &kp, const std::vector<uint32_t> &n_kp, const std::vector<uint32_t> &flags,
                       uint32_t max_scans, uint32_t max_total) {
  std::memset(dst, 0, block_floats(max_scans, max_total) * sizeof(float));
  const uint32_t nb = (uint32_t)(n_kp.size() < max_scans ? n_kp.size() : max_scans);
  std::vector<uint64_t> off(nb + 1, 0);
  for (uint32_t b = 0; b < nb; ++b) off[b + 1] = off[b] + n_kp[b];
  auto put = [&](size_t i, uint32_t v) { std::memcpy(dst + i, &v, 4); };
  auto clip = [&](uint64_t v) { return (uint32_t)(v < max_total ? v : max_total); };
  for (size_t i = 0; i < 4 * block_off_rows(max_scans); ++i) put(4 + i, clip(off[i < nb ? i : nb]));
  uint32_t flags_or = n_kp.size() > max_scans ? 0x4u : 0u;
  const size_t f0 = 4 + 4 * block_off_rows(max_scans), k0 = f0 + 4 * block_flag_rows(max_scans);
  for (uint32_t b = 0; b < nb; ++b) {
    const uint32_t o0 = clip(off[b]), o1 = clip(off[b + 1]);
    const uint32_t f = flags[b] | (o1 - o0 < n_kp[b] ? 0x4u /* FX_FLAG_KP_OVERFLOW: the scan keeps fewer keypoints than it has */ : 0u);
    put(f0 + b, f);
    flags_or |= f;
    if (o1 > o0) std::memcpy(dst + k0 + 4 * (size_t)o0, kp[b], (size_t)(o1 - o0) * 16);
  }
  put(0, nb), put(1, clip(off[nb])), put(2, flags_or), put(3, max_total);
}

}  // namespace fx
#endif
