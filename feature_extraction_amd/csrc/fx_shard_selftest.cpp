// fx_shard_selftest — CPU check of the C++ sharding plan and record layout (fx_shard.hpp), run as one process per rank
// by tests/test_cpp_sharding.py:  fx_shard_selftest TOTAL WORLD RANK IN.bin OUT.bin [REC_KP]
// IN.bin: TOTAL records of keypoints as the test wrote them ({u32 n, n x float4} per scan); the rank packs the records of
// its block exactly as fx_pack_keypoint_records lays them out and writes them to OUT.bin; stdout: "first last".
//   fx_shard_selftest TOTAL WORLD RANK IN.bin OUT.bin block MAX_TOTAL
// the same with the rank's scans as ONE compact keypoint block (fx_pack_keypoint_block's layout, max_scans = the plan's block
// size): what fx::MultiGpu hands the collective since 0.7.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fx_shard.hpp"

int main(int argc, char **argv) {
  if (argc != 6 && argc != 7 && argc != 8) return 2;
  const bool as_block = argc == 8;
  const uint32_t rec_kp = argc == 7 ? (uint32_t)std::atoi(argv[6]) : fx::kRecKeypoints;  // record stride (fx::MultiGpu: the contexts' max_keypoints)
  const uint64_t total = std::strtoull(argv[1], nullptr, 10);
  const uint32_t world = (uint32_t)std::atoi(argv[2]), rank = (uint32_t)std::atoi(argv[3]);
  const auto span = fx::shard_range(total, world, rank);
  for (uint64_t s = 0; s < total; ++s) {  // the plan is a partition: every scan has exactly one owner, consistent with the ranges
    const uint32_t o = fx::owner_of(s, total, world);
    const auto os = fx::shard_range(total, world, o);
    if (s < os.first || s >= os.second) return 3;
    // its row in the padded gathered table lies in its owner's block, behind the rows of the block's earlier scans
    const uint64_t row = fx::table_row(s, total, world), bs = fx::block_size(total, world);
    if (row / bs != o || row % bs != s - os.first || os.second - os.first > bs || bs > os.second - os.first + 1) return 7;
  }
  FILE *in = std::fopen(argv[4], "rb"), *out = std::fopen(argv[5], "wb");
  if (!in || !out) return 4;
  std::vector<float> rec(fx::record_floats(rec_kp));
  std::vector<std::vector<float>> mine;  // (block mode: this rank's scans)
  for (uint64_t s = 0; s < total; ++s) {
    uint32_t n = 0;
    if (std::fread(&n, 4, 1, in) != 1) return 5;
    std::vector<float> kp((size_t)n * 4);
    if (n && std::fread(kp.data(), 16, n, in) != n) return 5;
    if (s < span.first || s >= span.second) continue;
    if (as_block) {
      mine.push_back(kp);
      continue;
    }
    fx::pack_record(rec.data(), kp.data(), n, 0u, rec_kp);
    const fx::KeypointRecordView v = fx::record_of(rec.data(), 0, rec_kp);
    if (v.n_keypoints() != (n < rec_kp ? n : rec_kp)) return 6;
    std::fwrite(rec.data(), sizeof(float), rec.size(), out);
  }
  if (as_block) {
    const uint32_t max_scans = (uint32_t)fx::block_size(total, world), max_total = (uint32_t)std::atoi(argv[7]);
    std::vector<const float *> kp;
    std::vector<uint32_t> n_kp, flags;
    for (const auto &m : mine) kp.push_back(m.data()), n_kp.push_back((uint32_t)(m.size() / 4)), flags.push_back(0u);
    std::vector<float> blk(fx::block_floats(max_scans, max_total));
    fx::pack_block(blk.data(), kp, n_kp, flags, max_scans, max_total);
    const fx::KeypointBlockView v{blk.data(), max_scans};
    if (v.scans() != mine.size() || v.max_total() != max_total) return 8;
    for (uint32_t b = 0; b < v.scans(); ++b)
      if (v.n_keypoints(b) > n_kp[b] || ((v.n_keypoints(b) < n_kp[b]) != ((v.flags(b) & 0x4u) != 0u))) return 9;  // (cut exactly when flagged)
    std::fwrite(blk.data(), sizeof(float), blk.size(), out);
  }
  // the padding of a short block: empty records up to the plan's block size (what the rank hands the collective)
  for (uint64_t s = span.second - span.first; !as_block && s < fx::block_size(total, world); ++s) {
    fx::pack_record(rec.data(), nullptr, 0u, 0u, rec_kp);
    std::fwrite(rec.data(), sizeof(float), rec.size(), out);
  }
  std::fclose(in);
  std::fclose(out);
  std::printf("%llu %llu\n", (unsigned long long)span.first, (unsigned long long)span.second);
  return 0;
}
