// fx_api.cpp — context management and the batch driver behind include/fx.h.
//
// A context owns every device buffer (sized once from fx_limits: no allocation in the
// steady state), a stream, and lazily created pinned host mirrors.  fx_process_batch
// enqueues the stage kernels of fx_kernels.hip on the context's stream; with FX_OUT_HOST
// it also copies the results back and synchronises.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fx.h"
#include "fx_device.h"

extern "C" {
size_t fxk_ring_large_lds_bytes(uint32_t cap, uint32_t ccap);
void fxk_rings_large(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t cap, uint32_t ccap, uint32_t grid,
                     uint32_t after_runs2);
void fxk_rings_runs2(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t max_pts, uint32_t grid);
size_t fxk_merge_lds_bytes(uint32_t cap, uint32_t n_rings);
size_t fxk_desc_lds_bytes(uint32_t cap);
size_t fxk_gather_lds_bytes(uint32_t max_keypoints);
uint32_t fxk_dense_cells(void);
uint32_t fxk_group_cap(void);
uint32_t fxk_dfin_k(void);
void fxk_dense(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t n_cu, uint32_t rows, uint32_t items, uint32_t skip);
size_t fxk_dense_slow_words(uint32_t max_points);
size_t fxk_merge_huge_lds_bytes(uint32_t cap, uint32_t ccap, uint32_t n_rings);
hipError_t fxk_configure(size_t ring_big, size_t merge_big, size_t merge_huge, size_t desc_big, size_t gather);
uint32_t fxk_near_words(uint32_t max_points);
void fxk_prep(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, float near_margin, float el0, float inv_step,
              uint32_t clk_slot);
void fxk_bucket(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, float el0, float inv_step, uint32_t clk_next);
uint32_t fxk_prep_slices_max(void);
void fxk_prep_sliced(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, uint32_t slices, float near_margin, float el0, float inv_step,
                     uint32_t clk_slot);
void fxk_bucket_sliced(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, uint32_t slices, float el0, float inv_step,
                       uint32_t clk_next);
size_t fxk_ring_runs_lds_bytes(void);
void fxk_rings_runs(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, uint32_t max_pts, uint32_t grid);
void fxk_merge_small(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, uint32_t cap);
void fxk_merge_big(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t cap, uint32_t grid, uint32_t last);
void fxk_merge_huge(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t cap, uint32_t ccap, uint32_t grid);
size_t fxk_merge_hp_words(uint32_t cap);
uint32_t fxk_merge_slices_max(void);
void fxk_merge_huge_split(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t cap, uint32_t ccap, uint32_t grid, uint32_t slices);
uint32_t fxk_front_max_rings(void);
uint32_t fxk_front_merge_cap(void);
void fxk_front(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, float near_margin, float el0, float inv_step,
               uint32_t clk_slot, uint32_t merge_cap, uint32_t force_redo);
void fxk_front_ab(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, float near_margin, float el0, float inv_step,
                  uint32_t clk_slot, uint32_t force_redo);
void fxk_front_cd(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, uint32_t clk_slot, uint32_t merge_cap, uint32_t lean,
                  uint32_t self_n, uint32_t force_redo);
void fxk_front_redo(hipStream_t s, const FxDevParams &P, const FxBuffers &B, float el0, float inv_step, uint32_t huge_ccap, uint32_t force_slow,
                    uint32_t grid);
size_t fxk_slow_words(uint32_t max_ring_points, uint32_t max_candidates, uint32_t huge_ccap);
void fxk_slow(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t huge_ccap, uint32_t grid, uint32_t batch, uint32_t clk_next);
hipError_t fxk_configure_front(void);
uint32_t fxk_gather_slices(uint32_t batch);
uint32_t fxk_gather(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, float box_margin, uint32_t counted);
uint32_t fxk_gather_counted_max(void);
void fxk_desc_group(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, uint32_t grid);
void fxk_desc_mid(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, uint32_t cap, uint32_t n_wg, uint32_t n_wave,
                  uint32_t n_dslow);
void fxk_pack_kp_records(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, void *dst,
                         uint32_t rec_kp);
size_t fxk_kp_block_bytes(uint32_t max_scans, uint32_t max_total);
void fxk_pack_kp_block(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, void *dst, uint32_t max_scans, uint32_t max_total,
                       uint32_t grid);
void fxk_rng_ord(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch);
#ifdef FX_TEST_HOOKS
void fxk_test_sort_replay(hipStream_t s, const uint32_t *sizes, uint32_t n_seq, uint32_t n, uint32_t *perm);
void fxk_test_elevation(hipStream_t s, const float *xyz, uint32_t n, const double *tab, float *fast, uint8_t *ok, float *exact);
void fxk_test_within(hipStream_t s, const float4 *sp, uint32_t n, const float4 *queries, uint32_t nq, float r2, uint32_t *packed, uint32_t *plain);
#endif
void fxk_unpack_pc2(hipStream_t s, const void *src, uint32_t n, uint32_t point_step, uint32_t ox, uint32_t oy, uint32_t oz,
                    uint32_t oi, uint32_t big_endian, void *dst, uint32_t grid);
void fxk_pack_xyzi32(hipStream_t s, const void *src, uint32_t n, void *dst, uint32_t grid);
void fxk_pack_features(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, void *dst,
                       uint32_t capacity, uint32_t grid);
}

namespace {
thread_local std::string g_last_error;

fx_status fail(fx_status s, const std::string &msg) {
  g_last_error = msg;
  return s;
}
#define FX_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return fail(e_ == hipErrorOutOfMemory ? FX_ERR_OOM : FX_ERR_HIP,                            \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                             \
  } while (0)

constexpr int kMetaSlots = 8;
// Test and experiment hooks read the environment only in the TEST build of the library (lib/libfx_hip_test.so, compiled
// with -DFX_TEST_HOOKS: feature_extraction_amd/build.py); the product library never changes tiers, grids or kernels because
// of the caller's environment.
inline const char *test_hook(const char *name) {
#ifdef FX_TEST_HOOKS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}
constexpr uint32_t kMergeCapSmall = 512, kListCap = 4096, kDenseMin = 1024;
}  // namespace

struct fx_ctx {
  fx_params params;
  fx_limits lim;
  int device = 0;
  int n_cu = 256;
  FxDevParams dp;
  FxBuffers buf;
  std::vector<void *> dev_allocs;
  std::vector<void *> host_allocs;
  hipStream_t own_stream = nullptr, stream = nullptr;
  // per-batch scan table: ring of pinned slots so back-to-back batches never overwrite one in flight
  FxScanMeta *h_meta[kMetaSlots] = {};
  hipEvent_t meta_ev[kMetaSlots] = {};
  bool meta_used[kMetaSlots] = {};
  int meta_next = 0;
  FxScanMeta *d_meta = nullptr;
  float box_margin = 0.f;
  uint32_t desc_wgs_per_cu = 0;  // k_desc_group's workgroups a CU (0: by batches_in_flight; FX_DESC_WGS_PER_CU in the test build)
  uint32_t desc_grid_abs = 0;      // test hook (FX_DESC_GRID): k_desc_group's grid, absolute
  uint32_t batches_in_flight = 1;  // fx_set_batches_in_flight: the caller's contexts busy on this device at a time
  uint32_t ring_lds_cap = 0;     // points a ring may have in the workgroup ring tier's LDS (<= max_ring_points; beyond: k_slow)
  uint32_t merge_big_cap = 0;    // candidates the LDS merge tier holds as points (<= max_candidates)
  uint32_t merge_huge_cap = 0;   // candidates the large merge tier (coordinates in HBM, tables in LDS) holds (<= max_candidates; beyond: k_slow)
  uint32_t merge_huge_ccap = 0;  // clusters the large merge tier can order (>= max_keypoints)
  // host-input staging
  float *d_stage = nullptr;
  uint32_t stage_stride = 0;  // record stride the staging buffer is sized for
  // pinned host mirrors (lazy)
  uint32_t *h_hdr = nullptr;  // the per-scan words' block (hdr_stride words an array, the device block's layout): one copy a batch
  size_t hdr_stride = 0;
  uint32_t *h_n_kp = nullptr, *h_kp_offset = nullptr, *h_flags = nullptr, *h_n_filt = nullptr, *h_n_kpc = nullptr,
           *h_n_cand = nullptr, *h_cand_size = nullptr, *h_kpc_cand = nullptr, *h_kp_size = nullptr,
           *h_kp_nbrs = nullptr;
  int32_t *h_cand_kp = nullptr;
  float *h_keypoints = nullptr, *h_desc = nullptr, *h_filtered = nullptr, *h_kpc = nullptr, *h_cand = nullptr;
  // profiling
  // profiling: a ring of event sets so a whole timed region can be read back afterwards
  bool profiling = false;
  std::vector<hipEvent_t> ev_ring;  // depth * (FX_N_STAGES + 1)
  uint32_t ev_depth = 0, ev_count = 0;
  uint32_t prof_mask = ~0u;        // stages that get events (bit i = stage i); fx_set_profiling_stages
  std::vector<uint32_t> ev_mask;   // the mask each ring slot was recorded with
  std::vector<uint64_t> ev_seq;    // the batch sequence number each ring slot belongs to (k_prep's clock slot)
  uint64_t batch_seq = 0;          // batches enqueued so far
  double clk_khz = 100000.0;       // rate of the device's constant clock
  hipEvent_t *ev = nullptr;  // set of the batch being enqueued
  uint32_t last_batch = 0;
  // HIP graphs of the stage sequence, one per batch size (batches up to graph_max_batch)
  std::vector<std::pair<uint32_t, hipGraphExec_t>> graphs;
  uint32_t graph_max_batch = 0;
  bool debug_sync = false;
  // the previous batch's work for the rarely used tiers (pinned; written by the device: FxBuffers::tier_hint)
  volatile uint32_t *tier_hint = nullptr;
  uint32_t tier_min_grid = 8;  // FX_TIER_MIN_GRID: workgroups those tiers get at least (0: always the full grids)
  // what the tiers were handed lately: the largest count of the last hint_hold_batches batches (a workload that alternates
  // between batches with and without work for a tier keeps the tier's grid)
  uint32_t hint_hold[FX_N_HINTS] = {}, hint_age[FX_N_HINTS] = {};
  static constexpr uint32_t hint_hold_batches = 64;
  // The fused front kernel (k_front: filter to keypoints in one launch, for scans whose filtered cloud fits LDS) with
  // k_front_redo behind it.  A batch that hands more than an eighth of its scans to k_front_redo sends the next front_retry
  // batches through the separate kernels (k_prep ... k_merge_*), which are the fast way for large scans.
  bool front_ok = false;       // sensor within k_front's ring capacity
  bool front_last = false;     // what the last batch ran
  uint32_t front_force = 0;    // test hook (FX_FRONT_FORCE)
  // k_front as TWO launches (k_front_ab: streaming pass + ring split, the ring-major records through HBM; k_front_cd:
  // clustering + merge).  Built, parity-green, and NOT the default: alone the two take what the one takes
  // (0.134 + 0.156 against 0.286 ms), with four batches in flight the headline is 3 % lower (profiles/r06_experiments.md §1).
  // The test build's FX_FRONT_SPLIT=1 runs it (tests/test_gpu_front_split.py, the fuzz's front-split path).
  int front_stream = -1;       // test hook (FX_FRONT_STREAM): 1 always the sliced streaming pass + k_front_cd, 0 never; -1: batches of up to front_stream_max_batch scans
  static constexpr uint32_t front_stream_max_batch = 8;
  int front_split = 0;         // 1: the two launches; 2: the same with k_front_cd's lean image (the points left in HBM); 0: the one fused launch
  // A batch that failed after its kernels were enqueued leaves state the next batch would build on: descriptor rows are
  // cleared by un-writing what the last batch recorded for them (desc_nbins / desc_bins), the work-list counters are
  // cleared by the batch's first kernel, the tier hints size the next grids.  The next batch then starts from scratch:
  // every row is cleared whole, counters and hints are reset (fx_process_batch).
  bool state_suspect = false;
  uint32_t fail_after = 0;     // test hook (FX_FAIL_AFTER_ENQUEUE = n: the n-th batch returns an error after its kernels were enqueued)
  uint32_t front_pause = 0;    // batches left on the separate kernels
  // The dense descriptor tier: its own kernels (k_dense_sort, k_dense_density, k_dense_finish) when a recent batch had rows for it (or nothing is known), else a handful of
  // workgroups in k_desc_mid's launch (dense_slow_loop) that compute whatever does turn up, slower — the same results either way.
  uint32_t dense_fast_left = 0;            // batches that still get the tier's kernels after the last one that needed them
  static constexpr uint32_t dense_linger = 64;
  int gather_slices = -1;                  // test hook (FX_GATHER_COUNTED): workgroups a scan in the support gather's counted slices (1: one workgroup a scan)
  int merge_slices = -1;                   // test hook (FX_MERGE_SLICES): workgroups a scan in the large merge tier's pair loop (1: the one launch)
  int prep_slices = -1;                    // test hook (FX_PREP_SLICES): workgroups a scan in the separate kernels' streaming pass and ring split
  int dense_force = -1;                    // test hook (FX_DENSE_SLOW): 1 always k_desc_mid's workgroups, 0 always the tier's own kernels
  uint32_t skip_mask = 0;      // experiment hook (FX_SKIP_EMPTY, test build): bit 0 no k_front_redo, bit 1 no dense tier — only for workloads that need neither; bits 2 / 3: no k_dense_finish / k_dense_density (measurement: results wrong)
  static constexpr uint32_t front_retry = 64;
};

namespace {

template <typename T>
fx_status dev_alloc(fx_ctx *c, T **p, size_t count) {
  void *q = nullptr;
  const size_t bytes = (count ? count : 1) * sizeof(T);
  hipError_t e = hipMalloc(&q, bytes);
  if (e != hipSuccess) return fail(FX_ERR_OOM, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
  c->dev_allocs.push_back(q);
  *p = (T *)q;
  return FX_OK;
}
template <typename T>
fx_status host_alloc(fx_ctx *c, T **p, size_t count) {
  if (*p) return FX_OK;
  void *q = nullptr;
  const size_t bytes = (count ? count : 1) * sizeof(T);
  hipError_t e = hipHostMalloc(&q, bytes, hipHostMallocDefault);
  if (e != hipSuccess) return fail(FX_ERR_OOM, std::string("hipHostMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
  c->host_allocs.push_back(q);
  *p = (T *)q;
  return FX_OK;
}
#define FX_TRY(expr)              \
  do {                            \
    fx_status s_ = (expr);        \
    if (s_ != FX_OK) return s_;   \
  } while (0)

}  // namespace

namespace {
// k_prep's arctangent (elevation_fast): atan(c + d) = sum_k a_k d^k about c = i / FX_ATAN_N, a_0 = atan(c),
// a_k = (-1)^(k-1) sin^k(th) sin(k th) / k with th = pi/2 - atan(c) (the k-th derivative of the arctangent in closed form);
// evaluated in long double: the entries are good to the last bit or two of double.
std::vector<double> atan_table() {
  std::vector<double> at((FX_ATAN_N + 1) * (FX_ATAN_DEG + 1));
  for (int i = 0; i <= FX_ATAN_N; ++i) {
    const long double cpt = (long double)i / FX_ATAN_N, a0 = atanl(cpt), th = atan2l(1.0L, cpt), st = sinl(th);
    long double pw = 1.0L;
    at[(size_t)i * (FX_ATAN_DEG + 1)] = (double)a0;
    for (int kk = 1; kk <= FX_ATAN_DEG; ++kk) {
      pw *= st;
      at[(size_t)i * (FX_ATAN_DEG + 1) + kk] = (double)(((kk & 1) ? 1.0L : -1.0L) * pw * sinl(kk * th) / kk);
    }
  }
  return at;
}

// Enqueues the stage kernels of one batch on stream s (the whole device-side pipeline between the
// scan-table upload and the result copies).  Also what a HIP graph of the batch is captured from.
fx_status enqueue_stages(fx_ctx *c, hipStream_t s, uint32_t batch, bool prof, bool front, bool capture = false) {
  const fx_limits &L = c->lim;
  const FxDevParams &P = c->dp;
  const FxBuffers &B = c->buf;
  const uint32_t big_grid = (uint32_t)c->n_cu;
  // The tiers behind the common ones (workgroup-per-ring, LDS-sized merges, the dense descriptor tier) take their work from
  // lists by stride or ticket, so any grid computes the same; a launch that finds its list empty still has to place every
  // workgroup, and these want most of a CU each.  Their grids follow what the context's previous batch handed them (twice
  // that, at least tier_min_grid, at most the full grid; the full grid while nothing is known): a sparse batch costs
  // eight workgroups a tier instead of 256, and a batch that is suddenly dense runs its tiers narrow once.
  uint32_t hint[FX_N_HINTS];
  for (int i = 0; i < FX_N_HINTS; ++i) {
    hint[i] = (capture || !c->tier_min_grid) ? 0xffffffffu : c->tier_hint[i];
    if (capture || !c->tier_min_grid) continue;
    // (held: the largest count of the last hint_hold_batches batches; "nothing known yet" stays that only until something
    //  is known — held like a count it kept every rare tier at its full grid for the first 64 batches of a context, and
    //  after every reset)
    if (hint[i] == 0xffffffffu || c->hint_hold[i] == 0xffffffffu || hint[i] >= c->hint_hold[i] || ++c->hint_age[i] > fx_ctx::hint_hold_batches)
      c->hint_hold[i] = hint[i], c->hint_age[i] = 0;
    hint[i] = c->hint_hold[i];
  }
  // (min_wg: the scans k_front hands on — k_front_redo, k_slow — get ONE workgroup when the last 64 batches handed on
  //  nothing: an empty launch of eight k_front-shaped workgroups instead costs the headline 1 %, profiles/r05_experiments.md §9)
  auto tier_grid = [&](uint32_t work, uint32_t full, uint32_t bound, uint32_t per = 1, uint32_t min_wg = 0) -> uint32_t {
    uint32_t g = work > full / 2 ? full : 2 * work;  // (also: unknown)
    const uint32_t floor_wg = min_wg ? std::min(min_wg, c->tier_min_grid) : c->tier_min_grid;
    if (g < (floor_wg + per - 1) / per) g = (floor_wg + per - 1) / per;
    if (g > bound) g = bound ? bound : 1;            // never more workgroups than the batch could have items for
    return g < full ? g : full;
  };
  if (prof) {
    c->ev = &c->ev_ring[(size_t)(c->ev_count % c->ev_depth) * (FX_N_STAGES + 1)];
    c->ev_mask[c->ev_count % c->ev_depth] = c->prof_mask;
    c->ev_seq[c->ev_count % c->ev_depth] = c->batch_seq;
  }
  const uint32_t pmask = c->prof_mask;
  auto mark = [&](int i) -> hipError_t {
    if (c->debug_sync) {  // FX_DEBUG_SYNC=1: name the stage a device fault belongs to
      fprintf(stderr, "[fx] stage %d enqueued\n", i);
      hipError_t e = hipStreamSynchronize(s);
      if (e != hipSuccess) return e;
    }
    // event i closes stage i - 1 and opens stage i; the first and last always bracket the batch
    const bool wanted = i == 0 || i == FX_N_STAGES || ((pmask >> i) & 1u) || ((pmask >> (i - 1)) & 1u);
    return (prof && wanted) ? hipEventRecord(c->ev[i], s) : hipSuccess;
  };
  FX_HIP(mark(0));
  if (!batch) FX_HIP(hipMemsetAsync(B.counters, 0, FX_N_COUNTERS * sizeof(uint32_t), s));  // (k_prep clears them otherwise)
  if (batch) {
    const uint32_t merge_small = L.max_candidates < kMergeCapSmall ? L.max_candidates : kMergeCapSmall;
    // k_desc_group strides over the rows: ten workgroups a CU (resident all at once, 32 KB of LDS each) when the batch has the
    // chip to itself; with batches in flight fewer, longer-lived ones (measured, 1 / 2 / 3 / 4 batches in flight, scans/s with
    // 10 against 1 a CU: 1.64 / 1.47, 2.11 / 2.04, 2.25 / 2.27 — 2.29 with 3 —, 2.32 / 2.38e6: profiles/r05_experiments.md)
    const uint32_t desc_per_cu = c->desc_wgs_per_cu ? c->desc_wgs_per_cu : (c->batches_in_flight >= 4u ? 1u : (c->batches_in_flight == 3u ? 3u : 10u));
    // (never more workgroups than the batch can have rows for: sixteen rows a workgroup and trip — one scan per call launched
    //  2560 workgroups for its fifty rows)
    const uint32_t rows_bound = std::min<uint64_t>((uint64_t)batch * L.max_keypoints, L.max_total_keypoints);
    uint32_t desc_grid = std::max(1u, std::min((uint32_t)c->n_cu * desc_per_cu, (rows_bound + 15u) / 16u));
    if (c->desc_grid_abs) desc_grid = c->desc_grid_abs;  // (test hook)
    const uint32_t clk_slot = (uint32_t)(c->batch_seq % FX_CLK_SLOTS), clk_next = (uint32_t)((c->batch_seq + 1) % FX_CLK_SLOTS);
    const float el0 = (float)c->params.el0_deg, inv_step = (float)(1.0 / c->params.el_step_deg);
    if (front) {
      // stages 0-4 in three launches: k_front; k_front_redo — workgroups of k_front's shape, a few of them unless the
      // previous batch had work for it — for the scans that do not fit k_front's tables; and k_slow (256 threads, 9 KB of
      // LDS: an empty launch of it places anywhere) for what exceeds that shape's LDS too, on scratch in HBM.  All three go
      // with EVERY batch: the hints size grids, they never decide whether a scan gets what it needs.
      const uint32_t mcap = std::min(fxk_front_merge_cap(), L.max_candidates);
      const bool split = c->front_split != 0;
      // A handful of scans per call (the reference's own mode: one scan per callback, ref: node.cpp:72, :386): k_front's streaming
      // pass would be ONE workgroup a scan on a chip of 256 CUs — 0.036 of a scan's 0.093 ms in k_front.  So the streaming pass
      // and the ring split go out SLICED, sixteen workgroups a scan (the counting pass, k_prep_sliced, k_bucket_sliced: the
      // separate kernels' own, whose ring-major records are what k_front_ab writes), and k_front_cd clusters and merges in one.
      const bool stream = c->front_stream >= 0 ? c->front_stream != 0 : batch <= fx_ctx::front_stream_max_batch;
      if (stream && !split) {  // (stage 0: the streaming pass, stage 1: ring split + k_front_cd)
        fxk_prep_sliced(s, P, B, batch, fxk_prep_slices_max(), c->box_margin, el0, inv_step, clk_slot);
        FX_HIP(mark(1));
        fxk_bucket_sliced(s, P, B, batch, fxk_prep_slices_max(), el0, inv_step, clk_next);
        fxk_front_cd(s, P, B, batch, clk_slot, mcap, 0u, 1u, c->front_force >= 1u ? 1u : 0u);
      } else if (split) {  // (stage 0: k_front_ab, stage 1: k_front_cd)
        fxk_front_ab(s, P, B, batch, c->box_margin, el0, inv_step, clk_slot, c->front_force >= 1u ? 1u : 0u);
        FX_HIP(mark(1));
        fxk_front_cd(s, P, B, batch, clk_slot, mcap, c->front_split >= 2 ? 1u : 0u, 0u, 0u);
      } else {
        fxk_front(s, P, B, batch, c->box_margin, el0, inv_step, clk_slot, mcap, c->front_force >= 1u ? 1u : 0u);
        FX_HIP(mark(1));
      }
      for (int i = 2; i <= 4; ++i) FX_HIP(mark(i));
      // (the grid k_front_redo gets when the last 64 batches handed it nothing: ONE workgroup when the caller keeps four
      //  batches in flight — an empty launch of k_front-shaped workgroups waits for whole free CUs in the other batches' way:
      //  four of them cost the headline 0.6 %, eight 1.1 % —, four otherwise: a stream with the chip to itself pays nothing for
      //  them, and a batch that suddenly hands on many scans is not dealt through a single workgroup — ADVICE r5)
      const uint32_t redo_floor = c->batches_in_flight >= 4u ? 1u : 4u;
      if (!(c->skip_mask & 1u))
      fxk_front_redo(s, P, B, el0, inv_step, c->merge_huge_ccap, c->front_force >= 2u ? 1u : 0u, tier_grid(hint[6], 2 * big_grid, batch, 1, redo_floor));
      fxk_slow(s, P, B, c->merge_huge_ccap, tier_grid(hint[7], big_grid, batch), batch, clk_next);  // (its last workgroup does the batch's keypoint offsets too)
      FX_HIP(mark(5));
    } else {
    // one workgroup a scan when the batch fills the chip with that, several (a counting pass first) when it does not and the
    // scans are big enough to be worth a second read: 64 scans of 262 144 points are 64 workgroups on 256 CUs
    uint32_t slices = 1;
    if (c->prep_slices >= 0) {
      slices = (uint32_t)c->prep_slices;  // (test hook)
    } else if (4u * batch <= 2u * big_grid && L.max_points >= 65536u) {
      // (from four workgroups a scan on: with two — 256 scans of 131 072 points — the second read of the scan costs more
      //  than the wider grid gains: k_prep 0.15 -> 0.23 ms; with eight — 64 scans of 262 144 points — 0.23 -> 0.125 ms.
      //  Three workgroups a CU in all: what k_bucket_sliced's scalar registers (106: six wavefronts a SIMD) let be resident at once — 64 scans: the ring split
      //  0.092 / 0.065 / 0.109 ms with 8 / 12 / 16 a scan; 128 scans: 0.16 / 0.109 / 0.206 with 4 / 6 / 8,
      //  profiles/r06_experiments.md §12)
      slices = 3u * big_grid / batch;
    }
    slices = std::max(1u, std::min(slices, fxk_prep_slices_max()));
    if (slices > 1) {
      fxk_prep_sliced(s, P, B, batch, slices, c->box_margin, el0, inv_step, clk_slot);
      FX_HIP(mark(1));
      fxk_bucket_sliced(s, P, B, batch, slices, el0, inv_step, clk_next);
    } else {
      fxk_prep(s, P, B, batch, c->box_margin, el0, inv_step, clk_slot);
      FX_HIP(mark(1));
      fxk_bucket(s, P, B, batch, el0, inv_step, clk_next);
    }
    FX_HIP(mark(2));
    // one wavefront per (scan, ring): the hardware dispatcher balances the rings, whose costs differ a lot
    // (persistent wavefronts striding over the items: 0.18 ms instead of 0.14)
    fxk_rings_runs(s, P, B, batch, L.max_ring_points, (batch * (uint32_t)c->params.n_rings + 7) / 8 * 8);
    FX_HIP(mark(3));
    // sensors of more than the reference's 16 rings: dense rings, many of which need longer run tables than the first tier's
    const bool runs2 = c->params.n_rings > 16;
    // (these two take the list of XCD class blockIdx % 8: grids are multiples of 8, the hints the longest class list)
    const uint32_t n_items = batch * (uint32_t)c->params.n_rings;
    if (runs2) fxk_rings_runs2(s, P, B, L.max_ring_points, 8 * tier_grid(hint[0], (big_grid * 7 + 7) / 8, (n_items + 7) / 8, 8));  // (seven a CU: 21 KB of LDS each)
    fxk_rings_large(s, P, B, c->ring_lds_cap, c->ring_lds_cap, 8 * tier_grid(hint[runs2 ? 1 : 0], (big_grid + 7) / 8, (n_items + 7) / 8, 8),
                    runs2 ? 1u : 0u);
    FX_HIP(mark(4));
    fxk_merge_small(s, P, B, batch, merge_small);
    fxk_merge_big(s, P, B, c->merge_big_cap, tier_grid(hint[2], big_grid, batch), c->merge_big_cap >= L.max_candidates);
    if (c->merge_big_cap < L.max_candidates) {
      // one workgroup a scan in one launch — or, when the batches before had so few scans for this tier that those leave most
      // of the chip idle (64 scans of config 5 on 256 CUs), three launches with several workgroups a scan in the pair loop,
      // which is 97 % of the tier (merge_body's PHASE).  The hint decides speed, never a result.
      const uint32_t g = tier_grid(hint[3], big_grid, batch);
      uint32_t slices = 1;
      if (c->merge_slices >= 0)
        slices = (uint32_t)c->merge_slices;  // (test hook)
      else if (!capture && hint[3] != 0xffffffffu && hint[3] > 0 && 2u * hint[3] * c->batches_in_flight <= big_grid)
        slices = big_grid / (hint[3] * c->batches_in_flight);  // (the caller's other batches fill the chip as well: config 5 with four in flight 8.6e4 scans/s in one launch, 8.0e4 in three)
      slices = std::min(slices, fxk_merge_slices_max());
      if (slices > 1 && B.merge_hp)
        fxk_merge_huge_split(s, P, B, c->merge_huge_cap, c->merge_huge_ccap, g, slices);
      else
        fxk_merge_huge(s, P, B, c->merge_huge_cap, c->merge_huge_ccap, g);
    }
    fxk_slow(s, P, B, c->merge_huge_ccap, tier_grid(hint[7], big_grid, batch), batch, clk_next);  // (rings / merges beyond the LDS tiers, and the batch's keypoint offsets: see the front path)
    FX_HIP(mark(5));
    }
    if (P.estimate_descriptors) {
      // one workgroup a scan — or, for batches of few big scans with the chip to themselves (64 scans of 262 144 points: 64
      // workgroups on 256 CUs), several a scan in two launches: a counting pass, then the scatter with every slice's list
      // positions known (from four a scan on: the near sectors are read and tested twice)
      uint32_t counted = 1;
      if (c->gather_slices >= 0)
        counted = (uint32_t)c->gather_slices;  // (test hook)
      else if (!capture && 10u * batch * c->batches_in_flight <= 3u * big_grid && L.max_points >= 65536u)
        // (256-thread workgroups, three a CU: all resident at once.  From ten a scan on: config 5 — 64 scans — 0.255 ms with one
        //  wide workgroup a scan, 0.405 / 0.246 / 0.197 / 0.273 with 4 / 8 / 12 / 16 counted slices, profiles/r06_experiments.md §4)
        counted = 3u * big_grid / (batch * c->batches_in_flight);
      if (fxk_gather(s, P, B, batch, c->box_margin, counted) > 1) fxk_rng_ord(s, P, B, batch);  // (one workgroup per scan settles the RNG ordinals itself)
      FX_HIP(mark(6));
      fxk_desc_group(s, P, B, batch, desc_grid);
      FX_HIP(mark(7));
      // The dense tier (larger support sets and overflowed lists; nothing on sparse scans): its own three kernels, or a handful of
      // workgroups of k_desc_mid's launch — the previous batches decide speed, never results.
      // (beside k_desc_mid on a second stream of the context: measured and dropped, profiles/r04_front_experiments.md)
      bool fast = capture || hint[4] != 0u;  // (0xffffffff: nothing known yet)
      if (fast && !capture && hint[4] != 0xffffffffu) c->dense_fast_left = fx_ctx::dense_linger;
      if (!fast && c->dense_fast_left) --c->dense_fast_left, fast = true;
      if (c->dense_force >= 0) fast = c->dense_force == 0;
      const bool skip_dense = (c->skip_mask & 2u) != 0u;
      // wave rows and list rows (lists of up to dense_min entries, four keypoints per CU in flight) share a launch
      fxk_desc_mid(s, P, B, batch, P.list_cap < P.dense_min ? P.list_cap : P.dense_min, std::max(1u, std::min(big_grid * 4, rows_bound)),
                   std::max(1u, std::min(big_grid * 4, (rows_bound + 3u) / 4u)), fast || skip_dense ? 0u : (c->tier_min_grid ? c->tier_min_grid : 8u));
      FX_HIP(mark(8));
      if (fast && !skip_dense) {
        const uint32_t max_rows = batch * L.max_keypoints < L.max_total_keypoints ? batch * L.max_keypoints : L.max_total_keypoints;
        const uint32_t rows = tier_grid(hint[4], 3 * big_grid, max_rows);
        // density items: 1024 queries each, at most one a row more than the support points fill
        const uint32_t items = hint[5] == 0xffffffffu ? 3 * big_grid : tier_grid(hint[4] + hint[5] / 1024u, 3 * big_grid, 0xffffffffu);
        fxk_dense(s, P, B, big_grid, rows, items, (c->skip_mask >> 2) & 3u);
      }
    } else {
      for (int i = 6; i <= 8; ++i) FX_HIP(mark(i));
    }
    FX_HIP(mark(9));
    FX_HIP(hipGetLastError());
  } else {
    for (int i = 1; i <= FX_N_STAGES; ++i) FX_HIP(mark(i));
  }
  return FX_OK;
}

// Small batches are launch-bound (about 30 launches for a few hundred microseconds of work): replay the
// sequence as one HIP graph per batch size.  Kernel arguments depend on the batch size only; the scan
// table lives in the fixed d_meta buffer and is uploaded before the graph runs.
fx_status launch_graph(fx_ctx *c, hipStream_t s, uint32_t batch, bool front) {
  hipGraphExec_t exec = nullptr;
  const uint32_t key = batch | (front ? 0x80000000u : 0u);
  for (auto &g : c->graphs)
    if (g.first == key) exec = g.second;
  if (!exec) {
    hipGraph_t graph = nullptr;
    FX_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    fx_status st = enqueue_stages(c, s, batch, false, front, true);  // (a graph's grids are fixed: the full ones, bounded by the batch)
    hipError_t e = hipStreamEndCapture(s, &graph);
    if (st != FX_OK) {
      if (graph) (void)hipGraphDestroy(graph);
      return st;
    }
    FX_HIP(e);
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    FX_HIP(e);
    if (c->graphs.size() >= 8) {  // bounded cache: drop the oldest batch size
      (void)hipGraphExecDestroy(c->graphs.front().second);
      c->graphs.erase(c->graphs.begin());
    }
    c->graphs.emplace_back(key, exec);
  }
  FX_HIP(hipGraphLaunch(exec, s));
  return FX_OK;
}
}  // namespace

extern "C" {

const char *fx_last_error(void) { return g_last_error.c_str(); }

fx_status fx_check_abi(uint32_t header_version, size_t sizeof_params, size_t sizeof_limits, size_t sizeof_scan_desc, size_t sizeof_batch_view) {
  const uint32_t lib_version = ((uint32_t)FX_VERSION_MAJOR << 16) | FX_VERSION_MINOR;
  if (header_version == lib_version && sizeof_params == sizeof(fx_params) && sizeof_limits == sizeof(fx_limits) &&
      sizeof_scan_desc == sizeof(fx_scan_desc) && sizeof_batch_view == sizeof(fx_batch_view))
    return FX_OK;
  char msg[256];
  snprintf(msg, sizeof msg, "caller compiled against fx.h %u.%u (fx_params %zu, fx_limits %zu, fx_scan_desc %zu, fx_batch_view %zu bytes); "
           "this library is %u.%u (%zu, %zu, %zu, %zu)", header_version >> 16, header_version & 0xffffu, sizeof_params, sizeof_limits,
           sizeof_scan_desc, sizeof_batch_view, lib_version >> 16, lib_version & 0xffffu, sizeof(fx_params), sizeof(fx_limits),
           sizeof(fx_scan_desc), sizeof(fx_batch_view));
  return fail(FX_ERR_INVALID_ARG, msg);
}

fx_status fx_create(const fx_params *params, const fx_limits *limits, int device_id, fx_ctx **out) {
  if (!params || !limits || !out) return fail(FX_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  int n_dev = 0;
  hipError_t e = hipGetDeviceCount(&n_dev);
  if (e != hipSuccess || n_dev <= 0)
    return fail(FX_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + (e != hipSuccess ? hipGetErrorString(e) : "0 devices"));
  if (device_id < 0 || device_id >= n_dev) return fail(FX_ERR_INVALID_ARG, "device_id out of range");

  fx_limits L = *limits;
  fx_limits D;
  fx_limits_default(&D, L.max_batch, L.max_points);
  if (!L.max_ring_points) L.max_ring_points = D.max_ring_points;
  if (!L.max_ring_candidates) L.max_ring_candidates = D.max_ring_candidates;
  if (!L.max_candidates) L.max_candidates = D.max_candidates;
  if (!L.max_keypoints) L.max_keypoints = D.max_keypoints;
  if (!L.max_neighbors) L.max_neighbors = D.max_neighbors;
  if (!L.max_total_keypoints) L.max_total_keypoints = D.max_total_keypoints;
  if (!L.max_kpc_points) L.max_kpc_points = D.max_kpc_points;
  if (!L.max_dense_points) L.max_dense_points = D.max_dense_points;
  if (!L.max_overflow_points) L.max_overflow_points = D.max_overflow_points;
  if (L.max_batch == 0 || L.max_points == 0) return fail(FX_ERR_INVALID_ARG, "max_batch and max_points must be > 0");
  if (L.max_batch > 65535) return fail(FX_ERR_INVALID_ARG, "max_batch > 65535 (one grid row per scan in the support gather)");
  if (L.max_points > (1u << 20)) return fail(FX_ERR_INVALID_ARG, "max_points > 2^20 (descriptor sort key packs the point index in 20 bits)");
  if (params->n_rings < 1 || params->n_rings > 1024) return fail(FX_ERR_INVALID_ARG, "n_rings must be in [1, 1024]");
  if (!(params->descriptor_radius > 0.0) || !(params->el_step_deg > 0.0))
    return fail(FX_ERR_INVALID_ARG, "descriptor_radius and el_step_deg must be > 0");
  if (params->descriptor_radius < 1e-12 || params->descriptor_radius > 1e12)  // (the squared radii stay normal floats with room to scale)
    return fail(FX_ERR_INVALID_ARG, "descriptor_radius out of range");
  if (L.max_ring_candidates > L.max_ring_points) L.max_ring_candidates = L.max_ring_points;
  // LDS budget of the large tiers (160 KiB per workgroup on gfx950); what the limits allow beyond it takes the slow tier
  // (k_slow: the same bodies on scratch in HBM)
  const size_t kLds = 160 * 1024;
  if (L.max_ring_points > 32768 || L.max_candidates > 32768 || L.max_keypoints > 65535)
    return fail(FX_ERR_INVALID_ARG, "limit exceeds the 16-bit packing of the order replay (max_ring_points, max_candidates <= 32768)");
  uint32_t ring_lds_cap = L.max_ring_points;
  while (fxk_ring_large_lds_bytes(ring_lds_cap, ring_lds_cap) > kLds) ring_lds_cap -= ring_lds_cap > 64 ? 16 : 1;
  // merge tiers: up to merge_big_cap candidates a scan live in LDS as points; beyond that (dense many-ring scans)
  // the large tier keeps only parents and a cell sort in LDS
  uint32_t merge_big_cap = L.max_candidates;
  while (fxk_merge_lds_bytes(merge_big_cap, params->n_rings) > kLds) merge_big_cap -= merge_big_cap > 64 ? 64 : 1;
  if (const char *e = test_hook("FX_MERGE_BIG_CAP")) {  // test hook: push scans on to the large merge tier
    const uint32_t v = (uint32_t)atoi(e);
    if (v >= 16 && v < merge_big_cap) merge_big_cap = v;
  }
  const uint32_t merge_huge_ccap = L.max_keypoints > 64 ? L.max_keypoints : 64;
  uint32_t merge_huge_cap = L.max_candidates;
  while (merge_huge_cap > merge_big_cap && fxk_merge_huge_lds_bytes(merge_huge_cap, merge_huge_ccap, params->n_rings) > kLds) merge_huge_cap -= 64;
  if (merge_huge_cap < merge_big_cap) merge_huge_cap = merge_big_cap;
  if (const char *e = test_hook("FX_MERGE_HUGE_CAP")) {  // test hook: push scans on from the large merge tier to the slow one
    const uint32_t v = (uint32_t)atoi(e);
    if (v >= merge_big_cap && v < merge_huge_cap) merge_huge_cap = v;
  }
  if (const char *e = test_hook("FX_RING_LDS_CAP")) {  // test hook: push rings on from the workgroup ring tier to the slow one
    const uint32_t v = (uint32_t)atoi(e);
    if (v >= 16 && v < ring_lds_cap) ring_lds_cap = v & ~3u;
  }
  if (fxk_gather_lds_bytes(L.max_keypoints) > kLds) return fail(FX_ERR_INVALID_ARG, "support gather: LDS tables beyond the budget (build parameter FX_GATHER_KCAP)");

  FX_HIP(hipSetDevice(device_id));
  fx_ctx *c = new fx_ctx();
  c->params = *params;
  c->lim = L;
  c->device = device_id;
  c->ring_lds_cap = ring_lds_cap;
  c->merge_big_cap = merge_big_cap;
  c->merge_huge_cap = merge_huge_cap;
  c->merge_huge_ccap = merge_huge_ccap;
  if (const char *e = test_hook("FX_DESC_WGS_PER_CU")) {  // experiment hook: ignored outside 1..32
    const int v = atoi(e);
    if (v >= 1 && v <= 32) c->desc_wgs_per_cu = (uint32_t)v;
  }
  if (const char *e = test_hook("FX_DESC_GRID")) c->desc_grid_abs = (uint32_t)std::max(0, atoi(e));
  if (const char *e = test_hook("FX_DEBUG_SYNC")) c->debug_sync = atoi(e) != 0;
  if (const char *e = test_hook("FX_GRAPH_MAX_BATCH")) {
    const int v = atoi(e);
    if (v >= 0) c->graph_max_batch = (uint32_t)v;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->n_cu = prop.multiProcessorCount;

  // ---- constants, narrowed where PCL narrows them
  FxDevParams &P = c->dp;
  std::memset(&P, 0, sizeof(P));
  P.x_min = (float)params->x_min;  // PassThrough::setFilterLimits(const float&, const float&)
  P.x_max = (float)params->x_max;
  P.y_min = (float)params->y_min;
  P.y_max = (float)params->y_max;
  P.z_min = (float)params->z_min;
  P.z_max = (float)params->z_max;
  P.n_rings = params->n_rings;
  {
    // EuclideanClusterExtraction narrows the tolerance to float; KdTreeFLANN squares it in double
    const double tol = (double)(float)params->cluster_tolerance;
    P.r2_cluster = (float)(tol * tol);
    const double crt = (double)(float)params->cluster_radius_threshold;
    P.r2_merge = (float)(crt * crt);
  }
  P.min_count = (uint32_t)params->cluster_min_count;
  P.max_count = (uint32_t)params->cluster_max_count;
  P.gate_diameter = 2 * params->cluster_radius_threshold;
  P.crt = params->cluster_radius_threshold;
  P.ndc = (uint32_t)params->number_detection_channels;
  P.secondary_max = (uint32_t)params->secondary_max;
  {
    const double R = params->descriptor_radius, rd = params->descriptor_radius / 5.0;
    P.r2_search = (float)(R * R);  // 3DSC hands its radii to the tree as doubles
    P.r2_density = (float)(rd * rd);
    const double rs = (R + rd) * 1.0001 + 1e-4;  // conservative superset radius of the support set
    P.r2_support = (float)(rs * rs);
    c->box_margin = (float)(rs * 1.001 + 1e-3);  // k_gather's bounding-box reject, conservative
  }
  P.estimate_descriptors = params->estimate_descriptors;
  P.max_points = L.max_points;
  P.max_ring_cands = L.max_ring_candidates;
  P.max_candidates = L.max_candidates;
  P.max_keypoints = L.max_keypoints;
  P.max_total_kp = L.max_total_keypoints;
  P.max_kpc = L.max_kpc_points;
  P.max_neighbors = L.max_neighbors;
  P.max_ring_points = L.max_ring_points;
  P.list_cap = L.max_neighbors < kListCap ? L.max_neighbors : kListCap;
  P.near_words = fxk_near_words(L.max_points);
  // dense tier pools (fx_limits.max_dense_points; default: as many entries as the batch has points, at least 32 scans' worth)
  P.dense_min = kDenseMin;
  P.ovf_cap = L.max_overflow_points > 0x7ff00000u ? 0x7ff00000u : L.max_overflow_points;  // a scan's overflow region (fx_limits.max_overflow_points)
  {
    const unsigned long long want = L.max_dense_points ? L.max_dense_points : (unsigned long long)(L.max_batch > 32u ? L.max_batch : 32u) * L.max_points;
    P.dense_cap = (uint32_t)(want > 0xfff00000ull ? 0xfff00000ull : (want < 4096ull ? 4096ull : want));  // (a row takes its support points + 700 entries of it)
    const uint32_t per_row = P.list_cap < P.dense_min ? P.list_cap : P.dense_min;  // a dense row has more support points than this
    const unsigned long long rows = (unsigned long long)P.dense_cap / (per_row ? per_row : 1u) + 1ull;
    P.max_dense_rows = (uint32_t)(rows < L.max_total_keypoints ? rows : L.max_total_keypoints);
  }
  P.dense_qcap = P.dense_cap > 0x7ff00000u ? 0xfff00000u : 2u * P.dense_cap;  // (cells' queries padded to four: typically 1.2 entries per query)
  P.dense_lds_keys = fxk_dfin_k();
  if (const char *e = test_hook("FX_DENSE_LDS_KEYS")) {  // test hook: push rows on to the global-memory key sort (can only lower the cap)
    const int v = atoi(e);
    if (v >= 1 && (uint32_t)v < P.dense_lds_keys) P.dense_lds_keys = (uint32_t)v;
  }
  P.dense_won_points = 0xffffffffu;  // (the kernel's own bit map bounds it)
  if (const char *e = test_hook("FX_DENSE_WON_POINTS")) {  // test hook: rows of more points mark their queries in the sorted region itself
    const int v = atoi(e);
    if (v >= 0) P.dense_won_points = (uint32_t)v;
  }
  P.ring_slot_cap = 2 * L.max_points;  // worst case: every point on a window boundary, i.e. in two rings
  P.ring_list_cap = ((L.max_batch + 7) / 8) * (uint32_t)params->n_rings;  // rings of the scans of one XCD class

  fx_status st = FX_OK;
  auto bail = [&](fx_status s) {
    fx_destroy(c);
    return s;
  };
#define FX_A(expr)                     \
  st = (expr);                         \
  if (st != FX_OK) return bail(st)

  const size_t B = L.max_batch, R = (size_t)params->n_rings;
  FxBuffers &b = c->buf;
  std::memset(&b, 0, sizeof(b));
  FX_A(dev_alloc(c, &c->d_meta, B));
  b.meta = c->d_meta;
  float2 *d_win = nullptr;
  FX_A(dev_alloc(c, &d_win, R));
  b.ring_win = d_win;
  double *d_atan = nullptr;
  FX_A(dev_alloc(c, &d_atan, (size_t)(FX_ATAN_N + 1) * (FX_ATAN_DEG + 1)));
  b.atan_tab = d_atan;
  FxScTables *d_tab = nullptr;
  FX_A(dev_alloc(c, &d_tab, 1));
  b.tables = d_tab;
  float2 *d_xa = nullptr;
  FX_A(dev_alloc(c, &d_xa, L.max_keypoints));
  b.xaxis = d_xa;
  FX_A(dev_alloc(c, &b.filt, B * L.max_points));
  {
    // the five words a scan the host reads back after every batch — counts, the keypoint offsets, flags — in ONE block, so
    // that FX_OUT_HOST is one copy instead of five
    c->hdr_stride = (B + 1 + 3) & ~(size_t)3;
    uint32_t *hdr = nullptr;
    FX_A(dev_alloc(c, &hdr, 5 * c->hdr_stride));
    if (hipMemset(hdr, 0, 5 * c->hdr_stride * sizeof(uint32_t)) != hipSuccess) return bail(fail(FX_ERR_HIP, "hipMemset"));
    b.n_kp = hdr, b.kp_offset = hdr + c->hdr_stride, b.flags = hdr + 2 * c->hdr_stride, b.n_filt = hdr + 3 * c->hdr_stride,
    b.n_kpc = hdr + 4 * c->hdr_stride;
  }
  FX_A(dev_alloc(c, &b.near_bits, B * P.near_words));
  FX_A(dev_alloc(c, &b.prep_cnt, B * fxk_prep_slices_max()));
  FX_A(dev_alloc(c, &b.prep_ring_cnt, B * fxk_prep_slices_max() * R));
  FX_A(dev_alloc(c, &b.ring_cand, B * R * L.max_ring_candidates));
  FX_A(dev_alloc(c, &b.ring_cand_size, B * R * L.max_ring_candidates));
  FX_A(dev_alloc(c, &b.ring_cand_cnt, B * R));
  FX_A(dev_alloc(c, &b.ring_pts, B * P.ring_slot_cap));
  FX_A(dev_alloc(c, &b.ring_off, B * R));
  FX_A(dev_alloc(c, &b.ring_cnt, B * R));
  FX_A(dev_alloc(c, &b.kpc_pool, B * P.ring_slot_cap));
  FX_A(dev_alloc(c, &b.kpc_pool_cand, B * P.ring_slot_cap));
  FX_A(dev_alloc(c, &b.kpc_ring_cnt, B * R));
  FX_A(dev_alloc(c, &b.cand, B * L.max_candidates));
  FX_A(dev_alloc(c, &b.cand_size, B * L.max_candidates));
  FX_A(dev_alloc(c, &b.cand_kp, B * L.max_candidates));
  FX_A(dev_alloc(c, &b.n_cand, B));
  FX_A(dev_alloc(c, &b.keypoints, B * L.max_keypoints));
  FX_A(dev_alloc(c, &b.kp_size, B * L.max_keypoints));
  FX_A(dev_alloc(c, &b.kp_nbrs, B * L.max_keypoints));
  FX_A(dev_alloc(c, &b.kpc, B * L.max_kpc_points));
  FX_A(dev_alloc(c, &b.kpc_cand, B * L.max_kpc_points));
  FX_A(dev_alloc(c, &b.desc, (size_t)L.max_total_keypoints * FX_DESC_FLOATS + 4));
  FX_A(dev_alloc(c, &b.desc_nbins, (size_t)L.max_total_keypoints));
  FX_A(dev_alloc(c, &b.desc_bins, (size_t)L.max_total_keypoints * fxk_group_cap()));
  // rows start out zero with nothing recorded in them: k_desc_group then clears a row by un-writing what it wrote last time
  if (hipMemset(b.desc, 0, ((size_t)L.max_total_keypoints * FX_DESC_FLOATS + 4) * sizeof(float)) != hipSuccess ||
      hipMemset(b.desc_nbins, 0, (size_t)L.max_total_keypoints * sizeof(uint32_t)) != hipSuccess)
    return bail(fail(FX_ERR_HIP, "hipMemset"));
  FX_A(dev_alloc(c, &b.huge_rings, (size_t)8 * P.ring_list_cap));
  FX_A(dev_alloc(c, &b.huge_rings2, (size_t)8 * P.ring_list_cap));
  FX_A(dev_alloc(c, &b.big_merge, B));
  FX_A(dev_alloc(c, &b.huge_merge, B));
  FX_A(dev_alloc(c, &b.redo, B));
  FX_A(dev_alloc(c, &b.front_n, B));
  FX_A(dev_alloc(c, &b.slow, B));
  FX_A(dev_alloc(c, &b.slow_state, B));
  FX_A(dev_alloc(c, &b.ring_pending, B * ((R + 31) / 32)));
  if (hipMemset(b.slow_state, 0, B * sizeof(uint32_t)) != hipSuccess || hipMemset(b.ring_pending, 0, B * ((R + 31) / 32) * sizeof(uint32_t)) != hipSuccess)
    return bail(fail(FX_ERR_HIP, "hipMemset"));
  {
    // the slow tier's scratch: a region per workgroup of its largest grid — a CU's worth of workgroups, fewer for small
    // contexts or huge limits (256 MB at most)
    const size_t words = fxk_slow_words(L.max_ring_points, L.max_candidates, c->merge_huge_ccap);
    size_t slots = std::min<size_t>((size_t)c->n_cu, B);
    while (slots > 1 && slots * words * 4 > ((size_t)256 << 20)) --slots;
    P.gs_words = (uint32_t)words;
    P.gs_slots = (uint32_t)slots;
    FX_A(dev_alloc(c, &b.gs_pool, slots * words));
  }
  {
    // dense_slow_loop's scratch: a region per workgroup (a handful of them), 128 MB at most
    const size_t words = fxk_dense_slow_words(L.max_points);
    size_t slots = 32;
    while (slots > 1 && slots * words * 4 > ((size_t)128 << 20)) --slots;
    P.gsd_words = (uint32_t)words;
    P.gsd_slots = (uint32_t)slots;
    FX_A(dev_alloc(c, &b.gsd_pool, slots * words));
  }
  // (the large merge tier's bin-ordered copy: k_merge_huge, and k_front_redo — whose LDS image holds fewer candidates as points
  //  than k_merge_big's — whenever a scan has more candidates than that)
  FX_A(dev_alloc(c, &b.merge_sorted, (size_t)B * L.max_candidates));
  // (the large merge tier as three launches: a region per scan, when the tier exists and the regions stay within 256 MB)
  if (c->merge_big_cap < L.max_candidates && (size_t)B * fxk_merge_hp_words(c->merge_huge_cap) * 4 <= ((size_t)256 << 20))
    FX_A(dev_alloc(c, &b.merge_hp, (size_t)B * fxk_merge_hp_words(c->merge_huge_cap)));
  FX_A(dev_alloc(c, &b.list_desc, L.max_total_keypoints));
  FX_A(dev_alloc(c, &b.s_pts, (size_t)L.max_total_keypoints * P.list_cap));
  FX_A(dev_alloc(c, &b.s_cnt, L.max_total_keypoints));
  FX_A(dev_alloc(c, &b.gather_cnt, (size_t)B * fxk_gather_counted_max() * L.max_keypoints));
  FX_A(dev_alloc(c, &b.row_map, L.max_total_keypoints));
  FX_A(dev_alloc(c, &b.row_kp, L.max_total_keypoints));
  FX_A(dev_alloc(c, &b.row_xa, L.max_total_keypoints));
  FX_A(dev_alloc(c, &b.wave_desc, L.max_total_keypoints));
  // dense tier
  FX_A(dev_alloc(c, &b.dense_rows, P.max_dense_rows));
  FX_A(dev_alloc(c, &b.dense_order, (size_t)4 * P.max_dense_rows));
  FX_A(dev_alloc(c, &b.dense_off, P.max_dense_rows));
  FX_A(dev_alloc(c, &b.dense_koff, P.max_dense_rows));
  FX_A(dev_alloc(c, &b.dense_nq, P.max_dense_rows));
  FX_A(dev_alloc(c, &b.dense_nm, P.max_dense_rows));
  FX_A(dev_alloc(c, &b.dense_cells, (size_t)P.max_dense_rows * fxk_dense_cells()));
  FX_A(dev_alloc(c, &b.dense_items, (size_t)P.dense_cap / 256 + P.max_dense_rows + 1));
  FX_A(dev_alloc(c, &b.dense_pts, P.dense_cap));
  FX_A(dev_alloc(c, &b.dense_q, P.dense_qcap));
  FX_A(dev_alloc(c, &b.dense_qoff, P.max_dense_rows));
  FX_A(dev_alloc(c, &b.dense_key, P.dense_cap));
  FX_A(dev_alloc(c, &b.dens_cache, B * L.max_points));
  FX_A(dev_alloc(c, &b.seq, 1));
  FX_A(dev_alloc(c, &b.ovf_pts, (size_t)B * P.ovf_cap));
  FX_A(dev_alloc(c, &b.ovf_kp, (size_t)B * P.ovf_cap));
  FX_A(dev_alloc(c, &b.ovf_cnt, B));
  if (hipMemset(b.dens_cache, 0, B * L.max_points * sizeof(unsigned long long)) != hipSuccess) return bail(fail(FX_ERR_HIP, "hipMemset"));
  if (hipMemset(b.seq, 0, sizeof(unsigned long long)) != hipSuccess) return bail(fail(FX_ERR_HIP, "hipMemset"));
  if (hipMemset(b.ovf_cnt, 0, B * sizeof(uint32_t)) != hipSuccess) return bail(fail(FX_ERR_HIP, "hipMemset"));
  FX_A(dev_alloc(c, &b.counters, FX_N_COUNTER_WORDS));
  {
    uint32_t *h = nullptr;
    FX_A(host_alloc(c, &h, FX_N_HINTS));
    for (int i = 0; i < FX_N_HINTS; ++i) h[i] = 0xffffffffu;  // nothing known yet: full grids
    void *dv = nullptr;
    if (hipHostGetDevicePointer(&dv, h, 0) != hipSuccess) return bail(fail(FX_ERR_HIP, "hipHostGetDevicePointer"));
    b.tier_hint = (uint32_t *)dv;
    c->tier_hint = h;
    if (const char *e = test_hook("FX_TIER_MIN_GRID")) c->tier_min_grid = (uint32_t)std::max(0, atoi(e));
  }
  FX_A(dev_alloc(c, &b.clk, 2 * FX_CLK_SLOTS));
  {
    std::vector<unsigned long long> init(2 * FX_CLK_SLOTS);
    for (int i = 0; i < FX_CLK_SLOTS; ++i) init[2 * i] = ~0ull, init[2 * i + 1] = 0ull;
    if (hipMemcpy(b.clk, init.data(), init.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return bail(fail(FX_ERR_HIP, "hipMemcpy"));
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device_id) == hipSuccess && khz > 0) c->clk_khz = khz;
  }
  FX_A(dev_alloc(c, &b.stamps, 64 * 64));
  if (hipMemset(b.stamps, 0, 64 * 64 * 8) != hipSuccess) return bail(fail(FX_ERR_HIP, "hipMemset"));

  // ---- tables
  {
    std::vector<float2> win(R);
    const double half = params->el_step_deg / 2.0;
    for (size_t i = 0; i < R; ++i) {
      // ref: node.cpp:200-201: centre (i-7)*2-1, limits centre -+ 1.0, narrowed to float by PassThrough
      const double centre = params->el0_deg + (double)i * params->el_step_deg;
      win[i].x = (float)(centre - half);
      win[i].y = (float)(centre + half);
    }
    if (hipMemcpy(d_win, win.data(), R * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess)
      return bail(fail(FX_ERR_HIP, "upload ring windows"));
    {
      const std::vector<double> at = atan_table();
      if (hipMemcpy(d_atan, at.data(), at.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(FX_ERR_HIP, "upload arctangent table"));
    }
    FxScTables T;
    std::vector<float> lut(FX_DESC_BINS);
    fx_sc3d_tables(params->descriptor_radius, T.radii, T.theta, T.phi, lut.data());
    for (int k = 0; k < 11; ++k)
      for (int j = 0; j < 15; ++j) T.lut[k * 15 + j] = lut[k * 15 + j];  // azimuth bin 0; all azimuth bins are equal
    if (hipMemcpy(d_tab, &T, sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
      return bail(fail(FX_ERR_HIP, "upload 3DSC tables"));
    std::vector<float2> xa(L.max_keypoints);
    for (uint32_t k = 0; k < L.max_keypoints; ++k) {
      float xy[2];
      fx_sc3d_xaxis(k, xy);
      xa[k].x = xy[0];
      xa[k].y = xy[1];
    }
    if (hipMemcpy(d_xa, xa.data(), xa.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess)
      return bail(fail(FX_ERR_HIP, "upload 3DSC x-axes"));
  }
  for (int i = 0; i < kMetaSlots; ++i) {
    FX_A(host_alloc(c, &c->h_meta[i], B));
    if (hipEventCreateWithFlags(&c->meta_ev[i], hipEventDisableTiming) != hipSuccess)
      return bail(fail(FX_ERR_HIP, "hipEventCreate"));
  }
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail(FX_ERR_HIP, "hipStreamCreate"));
  c->stream = c->own_stream;
  {
    hipError_t ce = fxk_configure(fxk_ring_large_lds_bytes(c->ring_lds_cap, c->ring_lds_cap), fxk_merge_lds_bytes(c->merge_big_cap, params->n_rings),
                                  c->merge_big_cap < L.max_candidates ? fxk_merge_huge_lds_bytes(c->merge_huge_cap, c->merge_huge_ccap, params->n_rings) : 0,
                                  fxk_desc_lds_bytes(P.list_cap < P.dense_min ? P.list_cap : P.dense_min), fxk_gather_lds_bytes(L.max_keypoints));
    if (ce != hipSuccess) return bail(fail(FX_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(ce)));
  }
  if (hipMemset(b.counters, 0, FX_N_COUNTER_WORDS * sizeof(uint32_t)) != hipSuccess) return bail(fail(FX_ERR_HIP, "hipMemset"));
  c->front_ok = (uint32_t)params->n_rings <= fxk_front_max_rings();
  if (const char *e = test_hook("FX_FRONT")) c->front_ok = c->front_ok && atoi(e) != 0;  // 0 = the separate kernels (measurements; tests of those kernels)
  // 1: k_front hands every scan to k_front_redo; 2: and that one every ring and merge to the slow tier, k_slow (tests of those two)
  if (const char *e = test_hook("FX_FRONT_FORCE")) c->front_force = (uint32_t)std::max(0, atoi(e));
  if (const char *e = test_hook("FX_FRONT_STREAM")) c->front_stream = atoi(e) != 0 ? 1 : 0;
  if (const char *e = test_hook("FX_FRONT_SPLIT")) c->front_split = std::max(0, std::min(2, atoi(e)));  // (2: k_front_cd with the lean image)
  if (const char *e = test_hook("FX_FAIL_AFTER_ENQUEUE")) c->fail_after = (uint32_t)std::max(0, atoi(e));
  if (const char *e = test_hook("FX_SKIP_EMPTY")) c->skip_mask = (uint32_t)std::max(0, atoi(e));
  if (const char *e = test_hook("FX_DENSE_SLOW")) c->dense_force = atoi(e) != 0 ? 1 : 0;
  if (const char *e = test_hook("FX_PREP_SLICES")) c->prep_slices = std::max(1, atoi(e));
  if (const char *e = test_hook("FX_MERGE_SLICES")) c->merge_slices = std::max(1, atoi(e));
  if (const char *e = test_hook("FX_GATHER_COUNTED")) c->gather_slices = std::max(1, atoi(e));
  if (c->front_ok) {
    hipError_t ce = fxk_configure_front();
    if (ce != hipSuccess) return bail(fail(FX_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(ce)));
  }
  if (hipMemset(b.kp_offset, 0, (B + 1) * sizeof(uint32_t)) != hipSuccess) return bail(fail(FX_ERR_HIP, "hipMemset"));
  if (hipMemset(b.n_kp, 0, B * sizeof(uint32_t)) != hipSuccess) return bail(fail(FX_ERR_HIP, "hipMemset"));
#undef FX_A
  *out = c;
  return FX_OK;
}

void fx_destroy(fx_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
  (void)hipDeviceSynchronize();
  for (void *p : c->dev_allocs) (void)hipFree(p);
  for (void *p : c->host_allocs) (void)hipHostFree(p);
  if (c->d_stage) (void)hipFree(c->d_stage);
  for (int i = 0; i < kMetaSlots; ++i)
    if (c->meta_ev[i]) (void)hipEventDestroy(c->meta_ev[i]);
  for (hipEvent_t e : c->ev_ring) (void)hipEventDestroy(e);
  for (auto &g : c->graphs) (void)hipGraphExecDestroy(g.second);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

fx_status fx_set_stream(fx_ctx *c, void *hip_stream) {
  if (!c) return fail(FX_ERR_INVALID_ARG, "null ctx");
  c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
  return FX_OK;
}
fx_status fx_get_stream(fx_ctx *c, void **hip_stream) {
  if (!c || !hip_stream) return fail(FX_ERR_INVALID_ARG, "null argument");
  *hip_stream = (void *)c->stream;
  return FX_OK;
}
fx_status fx_set_batches_in_flight(fx_ctx *c, uint32_t n) {
  if (!c) return fail(FX_ERR_INVALID_ARG, "null ctx");
  c->batches_in_flight = n ? n : 1u;
  if (!c->graphs.empty()) {  // (captured grids are fixed: graphs are rebuilt with the new ones — once no replay of an old one is still running)
    FX_HIP(hipSetDevice(c->device));
    FX_HIP(hipStreamSynchronize(c->stream));
    for (auto &g : c->graphs) (void)hipGraphExecDestroy(g.second);
    c->graphs.clear();
  }
  return FX_OK;
}
fx_status fx_set_graph_batch(fx_ctx *c, uint32_t max_batch) {
  if (!c) return fail(FX_ERR_INVALID_ARG, "null ctx");
  c->graph_max_batch = max_batch;
  return FX_OK;
}

fx_status fx_set_profiling(fx_ctx *c, int depth) {
  if (!c || depth < 0) return fail(FX_ERR_INVALID_ARG, "bad argument");
  FX_HIP(hipSetDevice(c->device));
  FX_HIP(hipStreamSynchronize(c->stream));
  for (hipEvent_t e : c->ev_ring) (void)hipEventDestroy(e);
  c->ev_ring.clear();
  c->ev_depth = (uint32_t)depth;
  c->ev_count = 0;
  c->profiling = depth > 0;
  c->ev_ring.resize((size_t)depth * (FX_N_STAGES + 1), nullptr);
  c->ev_mask.assign((size_t)depth, ~0u);
  c->ev_seq.assign((size_t)depth, 0);
  for (hipEvent_t &e : c->ev_ring) FX_HIP(hipEventCreate(&e));
  return FX_OK;
}
fx_status fx_get_limits(const fx_ctx *c, fx_limits *l) {
  if (!c || !l) return fail(FX_ERR_INVALID_ARG, "null argument");
  *l = c->lim;
  return FX_OK;
}
fx_status fx_set_profiling_stages(fx_ctx *c, uint32_t stage_mask) {
  if (!c) return fail(FX_ERR_INVALID_ARG, "null ctx");
  c->prof_mask = stage_mask;
  return FX_OK;
}
fx_status fx_get_timings(fx_ctx *c, uint32_t back, fx_timings *t) {
  if (!c || !t) return fail(FX_ERR_INVALID_ARG, "null argument");
  std::memset(t, 0, sizeof(*t));
  if (!c->profiling || back >= c->ev_count || back >= c->ev_depth)
    return fail(FX_ERR_INVALID_ARG, "no profiled batch that far back (fx_set_profiling(depth) first)");
  hipEvent_t *ev = &c->ev_ring[(size_t)((c->ev_count - 1 - back) % c->ev_depth) * (FX_N_STAGES + 1)];
  FX_HIP(hipEventSynchronize(ev[FX_N_STAGES]));
  const uint32_t mask = c->ev_mask[(c->ev_count - 1 - back) % c->ev_depth];
  for (int i = 0; i < FX_N_STAGES; ++i)
    if ((mask >> i) & 1u) FX_HIP(hipEventElapsedTime(&t->ms[i], ev[i], ev[i + 1]));  // (stages without events stay 0)
  FX_HIP(hipEventElapsedTime(&t->total_ms, ev[0], ev[FX_N_STAGES]));
  {  // k_prep's own execution span (the batch has completed: its clock slot is final, and not yet reused)
    const uint64_t seq = c->ev_seq[(c->ev_count - 1 - back) % c->ev_depth];
    unsigned long long span[2] = {0, 0};
    if (c->batch_seq - seq < FX_CLK_SLOTS - 1) {
      FX_HIP(hipMemcpy(span, c->buf.clk + 2 * (seq % FX_CLK_SLOTS), sizeof(span), hipMemcpyDeviceToHost));
      if (span[1] > span[0]) t->k_prep_exec_ms = (float)((double)(span[1] - span[0]) / c->clk_khz);
    }
  }
  return FX_OK;
}
// Algorithmic bytes per stage of the last batch, from the batch's own counts (diagnostic: small device-to-host copies).
fx_status fx_get_stage_bytes(fx_ctx *c, fx_stage_bytes *out) {
  if (!c || !out) return fail(FX_ERR_INVALID_ARG, "null argument");
  std::memset(out, 0, sizeof(*out));
  const uint32_t B = c->last_batch;
  if (!B) return FX_OK;
  FX_HIP(hipSetDevice(c->device));
  FX_HIP(hipStreamSynchronize(c->stream));
  const fx_limits &L = c->lim;
  const FxBuffers &b = c->buf;
  const size_t R = (size_t)c->params.n_rings;
  auto fetch = [&](const uint32_t *src, size_t n, std::vector<uint32_t> &v) -> hipError_t {
    v.resize(n);
    return n ? hipMemcpy(v.data(), src, n * 4, hipMemcpyDeviceToHost) : hipSuccess;
  };
  std::vector<uint32_t> n_filt, ring_cnt, ring_cand, kpc_ring, n_cand, n_kp, n_kpc, kp_off, s_cnt, near;
  std::vector<FxScanMeta> meta(B);
  FX_HIP(hipMemcpy(meta.data(), c->d_meta, (size_t)B * sizeof(FxScanMeta), hipMemcpyDeviceToHost));
  FX_HIP(fetch(b.n_filt, B, n_filt));
  FX_HIP(fetch(b.ring_cnt, B * R, ring_cnt));
  FX_HIP(fetch(b.ring_cand_cnt, B * R, ring_cand));
  FX_HIP(fetch(b.kpc_ring_cnt, B * R, kpc_ring));
  FX_HIP(fetch(b.n_cand, B, n_cand));
  FX_HIP(fetch(b.n_kp, B, n_kp));
  FX_HIP(fetch(b.n_kpc, B, n_kpc));
  FX_HIP(fetch(b.kp_offset, B + 1, kp_off));
  uint32_t total_kp = kp_off[B] < L.max_total_keypoints ? kp_off[B] : L.max_total_keypoints;
  FX_HIP(fetch(b.s_cnt, total_kp, s_cnt));
  FX_HIP(fetch(b.near_bits, (size_t)B * c->dp.near_words, near));
  double n_pts = 0, nf = 0, n_ring = 0, n_rc = 0, n_mem = 0, nc = 0, nk = 0, nkpc = 0, near_pts = 0;
  for (uint32_t i = 0; i < B; ++i) {
    n_pts += meta[i].n, nf += n_filt[i], nc += n_cand[i], nk += n_kp[i], nkpc += n_kpc[i];
    for (size_t r = 0; r < R; ++r) n_ring += ring_cnt[i * R + r], n_rc += ring_cand[i * R + r], n_mem += kpc_ring[i * R + r];
    if (n_kp[i]) {  // (k_gather does not touch a scan without keypoints)
      const uint32_t words = (meta[i].n + 127u) / 128u;  // one bit per four points
      for (uint32_t w = 0; w < words && w < c->dp.near_words; ++w) near_pts += 4.0 * __builtin_popcount(near[(size_t)i * c->dp.near_words + w]);
    }
  }
  double s_small = 0, s_mid = 0, s_dense = 0, rows_mid = 0, rows_dense = 0;
  for (uint32_t r = 0; r < total_kp; ++r) {
    const double n = s_cnt[r];
    if (s_cnt[r] > c->dp.dense_min || s_cnt[r] > c->dp.list_cap)
      s_dense += n, rows_dense += 1;
    else if (s_cnt[r] > 64u)
      s_mid += n, rows_mid += 1;
    else
      s_small += n;
  }
  const bool desc = c->params.estimate_descriptors != 0;
  double *rd = out->read, *wr = out->written;
  rd[0] = 16.0 * n_pts, wr[0] = 16.0 * nf + n_pts / 32.0 + 4.0 * R * B;      // k_prep: the scan; ~cloud, near-sector bits, ring counts
  rd[1] = 16.0 * nf, wr[1] = 16.0 * n_ring;                                     // k_bucket: ~cloud; ring-major copy
  rd[2] = 16.0 * n_ring, wr[2] = 20.0 * n_rc + 20.0 * n_mem;                   // ring tiers: ring points; candidates + sizes, members + their candidate
  rd[3] = 0, wr[3] = 0;                                                         // (the larger ring tiers' share is counted with the first)
  rd[4] = 20.0 * n_rc + 20.0 * n_mem, wr[4] = 24.0 * nc + 20.0 * nk + 20.0 * nkpc;  // merge: candidates, members; keypoints_full + maps, keypoints, keypoint_cloud
  if (c->front_last) {  // k_front: the scan in; ~cloud, near bits, ring counts, keypoints_full + maps, keypoints, keypoint_cloud out — no intermediates
    wr[0] += wr[4];
    for (int i = 1; i <= 4; ++i) rd[i] = wr[i] = 0;
  }
  if (desc) {
    rd[5] = 16.0 * near_pts + 16.0 * nk, wr[5] = 16.0 * (s_small + s_mid + s_dense) + 40.0 * nk;  // k_gather: near sectors; support lists + row tables
    rd[6] = 16.0 * s_small + 44.0 * total_kp, wr[6] = 7956.0 * total_kp;       // k_desc_group: short lists; every row (cleared here)
    rd[7] = 16.0 * s_mid, wr[7] = 0.0 * rows_mid;                               // k_desc_mid: lists; (non-empty bins of rows already counted)
    rd[8] = 16.0 * s_dense, wr[8] = 0.0 * rows_dense;                           // dense tier: lists
  }
  return FX_OK;
}
fx_status fx_synchronize(fx_ctx *c) {
  if (!c) return fail(FX_ERR_INVALID_ARG, "null ctx");
  FX_HIP(hipStreamSynchronize(c->stream));
  return FX_OK;
}

fx_status fx_process_batch(fx_ctx *c, const fx_scan_desc *scans, uint32_t batch, uint32_t flags, fx_batch_view *out) {
  if (!c || (!scans && batch) || !out) return fail(FX_ERR_INVALID_ARG, "null argument");
  if (batch > c->lim.max_batch) return fail(FX_ERR_TOO_LARGE, "batch > max_batch");
  FX_HIP(hipSetDevice(c->device));
  const fx_limits &L = c->lim;
  hipStream_t s = c->stream;

  // ---- scan table
  const int slot = c->meta_next;
  c->meta_next = (c->meta_next + 1) % kMetaSlots;
  if (c->meta_used[slot]) FX_HIP(hipEventSynchronize(c->meta_ev[slot]));
  FxScanMeta *hm = c->h_meta[slot];
  const bool in_dev = (flags & FX_IN_DEVICE) != 0;
  for (uint32_t i = 0; i < batch; ++i) {
    const fx_scan_desc &d = scans[i];
    if (d.n_points > L.max_points) return fail(FX_ERR_TOO_LARGE, "scan has more points than max_points");
    if (d.n_points && !d.points) return fail(FX_ERR_INVALID_ARG, "scan with null points");
    if (d.stride_bytes < 16 || (d.stride_bytes % 16) != 0 || d.stride_bytes > 256)
      return fail(FX_ERR_INVALID_ARG, "stride_bytes must be a multiple of 16 in [16, 256]");
    if (((uintptr_t)d.points % 16) != 0) return fail(FX_ERR_INVALID_ARG, "points must be 16-byte aligned");
    hm[i].n = d.n_points;
    hm[i].pad_ = 0;
    // (scans of one sensor and instant share their attitude: the matrix — four trigonometric calls — is built once per run of
    //  equal angles; a 1024-scan batch spent 0.25 ms of host time here, most of the enqueue)
    if (i && d.roll == scans[i - 1].roll && d.pitch == scans[i - 1].pitch)
      std::memcpy(hm[i].R, hm[i - 1].R, sizeof(hm[i].R));
    else
      fx_rotation_from_roll_pitch(d.roll, d.pitch, hm[i].R);
    if (in_dev) {
      hm[i].pts = (const float *)d.points;
      hm[i].stride_f = d.stride_bytes / 4;
    }
  }
  if (!in_dev && batch) {
    // Host scans go over as they are, whatever their record stride (16: packed x y z i; 32: pcl::PointXYZI in memory, what
    // the reference-side binding has — ref: node.cpp:81): the kernels stride over the records (FxScanMeta::stride_f), so
    // nothing is repacked on the host and nothing is waited for.  The staging buffer holds max_batch scans of the widest
    // stride seen so far.
    uint32_t stride = 16;
    for (uint32_t i = 0; i < batch; ++i) stride = std::max(stride, scans[i].stride_bytes);
    if (!c->d_stage || stride > c->stage_stride) {
      if (c->d_stage) {  // (an earlier batch issued without FX_OUT_HOST may still be reading it)
        FX_HIP(hipStreamSynchronize(s));
        FX_HIP(hipFree(c->d_stage));
        c->d_stage = nullptr;
      }
      void *q = nullptr;
      hipError_t e = hipMalloc(&q, (size_t)L.max_batch * L.max_points * stride);
      if (e != hipSuccess) return fail(FX_ERR_OOM, std::string("staging hipMalloc: ") + hipGetErrorString(e));
      c->d_stage = (float *)q;
      c->stage_stride = stride;
    }
    const size_t slot_bytes = (size_t)L.max_points * c->stage_stride;
    for (uint32_t i = 0; i < batch; ++i) {
      const fx_scan_desc &d = scans[i];
      uint8_t *dst = (uint8_t *)c->d_stage + (size_t)i * slot_bytes;
      hm[i].pts = (const float *)dst;
      hm[i].stride_f = d.stride_bytes / 4;
      if (!d.n_points) continue;
      // full-size scans that follow one another in host memory (a stacked batch) go over in one copy
      // (up to the last record's x y z i: nothing behind it is read, on the host or on the device)
      uint32_t j = i;
      size_t bytes = (size_t)(d.n_points - 1u) * d.stride_bytes + 16u;
      while (d.n_points == L.max_points && d.stride_bytes == c->stage_stride && j + 1 < batch && scans[j + 1].stride_bytes == d.stride_bytes &&
             scans[j + 1].n_points == L.max_points && (const uint8_t *)scans[j + 1].points == (const uint8_t *)scans[j].points + slot_bytes) {
        ++j;
        bytes += slot_bytes;
        hm[j].pts = (const float *)((uint8_t *)c->d_stage + (size_t)j * slot_bytes);
        hm[j].stride_f = d.stride_bytes / 4;
      }
      FX_HIP(hipMemcpyAsync(dst, d.points, bytes, hipMemcpyHostToDevice, s));
      i = j;
    }
  }
  if (batch) FX_HIP(hipMemcpyAsync(c->d_meta, hm, (size_t)batch * sizeof(FxScanMeta), hipMemcpyHostToDevice, s));
  FX_HIP(hipEventRecord(c->meta_ev[slot], s));
  c->meta_used[slot] = true;

  // ---- state a failed batch may have left behind (see fx_ctx::state_suspect)
  if (c->state_suspect) {
    FX_HIP(hipMemsetAsync(c->buf.desc_nbins, 0xff, (size_t)L.max_total_keypoints * sizeof(uint32_t), s));  // FX_ROW_DIRTY: clear whole
    FX_HIP(hipMemsetAsync(c->buf.counters, 0, FX_N_COUNTER_WORDS * sizeof(uint32_t), s));
    FX_HIP(hipMemsetAsync(c->buf.slow_state, 0, (size_t)L.max_batch * sizeof(uint32_t), s));  // (the slow tier's list markers: k_slow clears what it has done)
    FX_HIP(hipMemsetAsync(c->buf.ring_pending, 0, (size_t)L.max_batch * (((size_t)c->params.n_rings + 31) / 32) * sizeof(uint32_t), s));
    FX_HIP(hipStreamSynchronize(s));  // (the hints are host memory the device writes: nothing of the failed batch may land after the reset)
    for (int i = 0; i < FX_N_HINTS; ++i) c->tier_hint[i] = 0xffffffffu, c->hint_hold[i] = 0xffffffffu, c->hint_age[i] = 0;
    c->front_pause = 0;
    c->state_suspect = false;
  }
  struct Suspect {  // armed from the first launch to the successful return
    fx_ctx *c;
    bool armed = false;
    ~Suspect() {
      if (armed) c->state_suspect = true;
    }
  } suspect{c};
  suspect.armed = true;
  // ---- kernels
  const FxDevParams &P = c->dp;
  const FxBuffers &B = c->buf;
  const bool prof = c->profiling;
  // front kernel or separate kernels (see fx_ctx::front_pause): tier_hint[6] is what the last COMPLETED batch of this context
  // handed to k_front_redo (with batches in flight: an older batch's count, possibly of another size — this decides speed
  // only, never a result)
  bool front = c->front_ok;
  if (front && c->front_pause) {
    --c->front_pause;
    front = false;
  } else if (front && !c->front_force && c->front_last && c->tier_hint[6] != 0xffffffffu && c->tier_hint[6] > (c->last_batch + 7u) / 8u) {
    c->front_pause = fx_ctx::front_retry;
    front = false;
  }
  if (!prof && !c->debug_sync && batch && batch <= c->graph_max_batch && s != nullptr)
    FX_TRY(launch_graph(c, s, batch, front));
  else
    FX_TRY(enqueue_stages(c, s, batch, prof, front));
  c->front_last = front;
#ifdef FX_TEST_HOOKS
  if (c->fail_after && c->batch_seq + 1 == c->fail_after) {  // a failure after the kernels went out
    ++c->batch_seq;
    return fail(FX_ERR_HIP, "FX_FAIL_AFTER_ENQUEUE (test hook)");
  }
#endif
  if (prof) ++c->ev_count;
  ++c->batch_seq;
  c->last_batch = batch;

  // ---- view
  std::memset(out, 0, sizeof(*out));
  out->batch = batch;
  out->max_points = L.max_points;
  out->max_keypoints = L.max_keypoints;
  out->max_candidates = L.max_candidates;
  out->max_kpc_points = L.max_kpc_points;
  out->d_n_keypoints = B.n_kp;
  out->d_kp_offset = B.kp_offset;
  out->d_keypoints = (const float *)B.keypoints;
  out->d_descriptors = B.desc;
  out->d_flags = B.flags;
  out->d_n_filtered = B.n_filt;
  out->d_filtered = (const float *)B.filt;
  out->d_n_kpc = B.n_kpc;
  out->d_kpc = (const float *)B.kpc;

  if (!(flags & FX_OUT_HOST)) {
    suspect.armed = false;
    return FX_OK;
  }
  const size_t Bm = L.max_batch;
  const size_t S = c->hdr_stride;
  FX_TRY(host_alloc(c, &c->h_hdr, 5 * S));
  c->h_n_kp = c->h_hdr, c->h_kp_offset = c->h_hdr + S, c->h_flags = c->h_hdr + 2 * S, c->h_n_filt = c->h_hdr + 3 * S, c->h_n_kpc = c->h_hdr + 4 * S;
  FX_TRY(host_alloc(c, &c->h_keypoints, Bm * L.max_keypoints * 4));
  FX_TRY(host_alloc(c, &c->h_desc, (size_t)L.max_total_keypoints * FX_DESC_FLOATS));
  if (batch) {
    // (n_kp | kp_offset | flags | n_filt | n_kpc, S words each: everything up to the last array's last scan of this batch)
    FX_HIP(hipMemcpyAsync(c->h_hdr, B.n_kp, (4 * S + batch) * 4, hipMemcpyDeviceToHost, s));
    FX_HIP(hipMemcpyAsync(c->h_keypoints, B.keypoints, (size_t)batch * L.max_keypoints * 16, hipMemcpyDeviceToHost, s));
  } else {
    c->h_kp_offset[0] = 0;
  }
  FX_HIP(hipStreamSynchronize(s));
  uint32_t total = batch ? c->h_kp_offset[batch] : 0;
  if (total > L.max_total_keypoints) total = L.max_total_keypoints;
  out->total_keypoints = total;
  // (about as many rows as the last call had, copied BEFORE the wait, and the per-scan words in one block instead of five
  //  copies: measured for a scan per call, 0.244 ms either way — the copies are not what the call waits for; the one block stayed)
  if (total && P.estimate_descriptors)
    FX_HIP(hipMemcpyAsync(c->h_desc, B.desc, (size_t)total * FX_DESC_FLOATS * 4, hipMemcpyDeviceToHost, s));
  if (flags & FX_OUT_CLOUDS) {
    FX_TRY(host_alloc(c, &c->h_filtered, Bm * L.max_points * 4));
    FX_TRY(host_alloc(c, &c->h_kpc, Bm * L.max_kpc_points * 4));
    for (uint32_t i = 0; i < batch; ++i) {
      if (c->h_n_filt[i])
        FX_HIP(hipMemcpyAsync(c->h_filtered + (size_t)i * L.max_points * 4, B.filt + (size_t)i * L.max_points,
                              (size_t)c->h_n_filt[i] * 16, hipMemcpyDeviceToHost, s));
      if (c->h_n_kpc[i])
        FX_HIP(hipMemcpyAsync(c->h_kpc + (size_t)i * L.max_kpc_points * 4, B.kpc + (size_t)i * L.max_kpc_points,
                              (size_t)c->h_n_kpc[i] * 16, hipMemcpyDeviceToHost, s));
    }
    out->h_filtered = c->h_filtered;
    out->h_kpc = c->h_kpc;
  }
  if (flags & FX_OUT_DEBUG) {
    FX_TRY(host_alloc(c, &c->h_n_cand, Bm));
    FX_TRY(host_alloc(c, &c->h_cand, Bm * L.max_candidates * 4));
    FX_TRY(host_alloc(c, &c->h_cand_size, Bm * L.max_candidates));
    FX_TRY(host_alloc(c, &c->h_cand_kp, Bm * L.max_candidates));
    FX_TRY(host_alloc(c, &c->h_kpc_cand, Bm * L.max_kpc_points));
    FX_TRY(host_alloc(c, &c->h_kp_size, Bm * L.max_keypoints));
    FX_TRY(host_alloc(c, &c->h_kp_nbrs, Bm * L.max_keypoints));
    if (batch) {
      FX_HIP(hipMemcpyAsync(c->h_n_cand, B.n_cand, batch * 4, hipMemcpyDeviceToHost, s));
      FX_HIP(hipMemcpyAsync(c->h_cand, B.cand, (size_t)batch * L.max_candidates * 16, hipMemcpyDeviceToHost, s));
      FX_HIP(hipMemcpyAsync(c->h_cand_size, B.cand_size, (size_t)batch * L.max_candidates * 4, hipMemcpyDeviceToHost, s));
      FX_HIP(hipMemcpyAsync(c->h_cand_kp, B.cand_kp, (size_t)batch * L.max_candidates * 4, hipMemcpyDeviceToHost, s));
      FX_HIP(hipMemcpyAsync(c->h_kpc_cand, B.kpc_cand, (size_t)batch * L.max_kpc_points * 4, hipMemcpyDeviceToHost, s));
      FX_HIP(hipMemcpyAsync(c->h_kp_size, B.kp_size, (size_t)batch * L.max_keypoints * 4, hipMemcpyDeviceToHost, s));
      FX_HIP(hipMemcpyAsync(c->h_kp_nbrs, B.kp_nbrs, (size_t)batch * L.max_keypoints * 4, hipMemcpyDeviceToHost, s));
    }
    out->h_n_candidates = c->h_n_cand;
    out->h_candidates = c->h_cand;
    out->h_cand_size = c->h_cand_size;
    out->h_cand_keypoint = c->h_cand_kp;
    out->h_kpc_cand = c->h_kpc_cand;
    out->h_kp_size = c->h_kp_size;
    out->h_kp_neighbors = c->h_kp_nbrs;
  }
  FX_HIP(hipStreamSynchronize(s));
  out->h_n_keypoints = c->h_n_kp;
  out->h_kp_offset = c->h_kp_offset;
  out->h_keypoints = c->h_keypoints;
  out->h_descriptors = c->h_desc;
  out->h_flags = c->h_flags;
  out->h_n_filtered = c->h_n_filt;
  out->h_n_kpc = c->h_n_kpc;
  suspect.armed = false;
  return FX_OK;
}

#ifdef FX_TEST_HOOKS
// Test hook: k_prep's two elevation paths on caller-supplied points.
fx_status fx_test_elevation_device(int device, const float *xyz, uint32_t n, float *fast_out, uint8_t *fast_ok_out, float *exact_out) {
  if (!xyz || !fast_out || !fast_ok_out || !exact_out) return fail(FX_ERR_INVALID_ARG, "null argument");
  if (!n) return FX_OK;
  FX_HIP(hipSetDevice(device));
  const std::vector<double> at = atan_table();
  char *d = nullptr;
  const size_t o_tab = ((size_t)n * 12 + 7) & ~(size_t)7, o_fast = o_tab + at.size() * 8, o_exact = o_fast + (size_t)n * 4, o_ok = o_exact + (size_t)n * 4;
  FX_HIP(hipMalloc((void **)&d, o_ok + n));
  hipError_t e = hipMemcpy(d, xyz, (size_t)n * 12, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + o_tab, at.data(), at.size() * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    fxk_test_elevation(nullptr, (const float *)d, n, (const double *)(d + o_tab), (float *)(d + o_fast), (uint8_t *)(d + o_ok), (float *)(d + o_exact));
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(fast_out, d + o_fast, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(exact_out, d + o_exact, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(fast_ok_out, d + o_ok, n, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  FX_HIP(e);
  return FX_OK;
}

// Test hook: the descriptor tiers' packed within-radius count and the plain compare it replaces.
fx_status fx_test_within_device(int device, const float *support_xyzw, uint32_t n, const float *query_xyzw, uint32_t nq, float r2,
                                uint32_t *packed_out, uint32_t *plain_out) {
  if (!support_xyzw || !query_xyzw || !packed_out || !plain_out) return fail(FX_ERR_INVALID_ARG, "null argument");
  if (!nq) return FX_OK;
  FX_HIP(hipSetDevice(device));
  char *d = nullptr;
  const size_t o_q = (size_t)(n ? n : 1) * 16, o_a = o_q + (size_t)nq * 16, o_c = o_a + (size_t)nq * 4;
  FX_HIP(hipMalloc((void **)&d, o_c + (size_t)nq * 4));
  hipError_t e = n ? hipMemcpy(d, support_xyzw, (size_t)n * 16, hipMemcpyHostToDevice) : hipSuccess;
  if (e == hipSuccess) e = hipMemcpy(d + o_q, query_xyzw, (size_t)nq * 16, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    fxk_test_within(nullptr, (const float4 *)d, n, (const float4 *)(d + o_q), nq, r2, (uint32_t *)(d + o_a), (uint32_t *)(d + o_c));
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(packed_out, d + o_a, (size_t)nq * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(plain_out, d + o_c, (size_t)nq * 4, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  FX_HIP(e);
  return FX_OK;
}

// Test hook: the device build of the cluster-order replay on caller-supplied size sequences.
fx_status fx_test_sort_replay_device(int device, const uint32_t *sizes, uint32_t n_seq, uint32_t n, uint32_t *perm_out) {
  if (!sizes || !perm_out) return fail(FX_ERR_INVALID_ARG, "null argument");
  if (n > 192) return fail(FX_ERR_TOO_LARGE, "n > 192");
  if (!n_seq || !n) return FX_OK;
  FX_HIP(hipSetDevice(device));
  const size_t bytes = (size_t)n_seq * n * sizeof(uint32_t);
  uint32_t *d_in = nullptr, *d_out = nullptr;
  FX_HIP(hipMalloc((void **)&d_in, bytes));
  hipError_t e = hipMalloc((void **)&d_out, bytes);
  if (e == hipSuccess) e = hipMemcpy(d_in, sizes, bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    fxk_test_sort_replay(nullptr, d_in, n_seq, n, d_out);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(perm_out, d_out, bytes, hipMemcpyDeviceToHost);
  (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  FX_HIP(e);
  return FX_OK;
}

#endif  // FX_TEST_HOOKS

// Diagnostic: what the last completed batch left for the next one's tier grids (FxBuffers::tier_hint).
fx_status fx_debug_tier_hints(fx_ctx *c, uint32_t *out /* FX_N_HINTS = 8 words */) {
  if (!c || !out) return fail(FX_ERR_INVALID_ARG, "null argument");
  FX_HIP(hipSetDevice(c->device));
  FX_HIP(hipStreamSynchronize(c->stream));
  for (int i = 0; i < FX_N_HINTS; ++i) out[i] = c->tier_hint[i];
  return FX_OK;
}
// Diagnostic: 1 when the last batch went through the fused front kernel (k_front), 0 when through the separate kernels.
int fx_debug_front(fx_ctx *c) { return c && c->front_last ? 1 : 0; }
// Diagnostic: the work-list counters of the last batch (rings / scans / keypoint rows deferred to larger tiers).
fx_status fx_debug_counters(fx_ctx *c, uint32_t *out8 /* 16 words */) {
  if (!c || !out8) return fail(FX_ERR_INVALID_ARG, "null argument");
  FX_HIP(hipSetDevice(c->device));
  FX_HIP(hipStreamSynchronize(c->stream));
  uint32_t all[FX_N_COUNTERS];
  FX_HIP(hipMemcpy(all, c->buf.counters, sizeof(all), hipMemcpyDeviceToHost));
  std::memcpy(out8, all, 16 * 4);
  out8[0] = out8[5] = 0;  // the rings handed on are counted per XCD class: 0 to the second run tier, 5 to the workgroup tier
  const bool runs2 = c->params.n_rings > 16;
  for (int k = 0; k < 8; ++k) {
    if (runs2) out8[0] += all[FX_CNT_LARGE + k];
    out8[5] += all[(runs2 ? FX_CNT_LARGE2 : FX_CNT_LARGE) + k];
  }
  return FX_OK;
}

// Diagnostic (-DFX_STAMPS builds): cumulative per-phase cycle counters of the ring kernel.
fx_status fx_debug_stamps(fx_ctx *c, unsigned long long *out32 /* 64 words */) {
  if (!c || !out32) return fail(FX_ERR_INVALID_ARG, "null argument");
  FX_HIP(hipSetDevice(c->device));
  FX_HIP(hipStreamSynchronize(c->stream));
  std::vector<unsigned long long> all(64 * 64);
  FX_HIP(hipMemcpy(all.data(), c->buf.stamps, all.size() * 8, hipMemcpyDeviceToHost));
  for (int k = 0; k < 64; ++k) {
    out32[k] = 0;
    for (int w = 0; w < 64; ++w) out32[k] += all[(size_t)w * 64 + k];
  }
  return FX_OK;
}

// Diagnostic (-DFX_STAMPS builds): the raw stamp words (64 x 64).
fx_status fx_debug_stamps_raw(fx_ctx *c, unsigned long long *out4096) {
  if (!c || !out4096) return fail(FX_ERR_INVALID_ARG, "null argument");
  FX_HIP(hipSetDevice(c->device));
  FX_HIP(hipStreamSynchronize(c->stream));
  FX_HIP(hipMemcpy(out4096, c->buf.stamps, 64 * 64 * 8, hipMemcpyDeviceToHost));
  return FX_OK;
}

fx_status fx_pack_keypoint_records(fx_ctx *c, void *dst_device, uint32_t rec_keypoints) {
  if (!c || !dst_device || !rec_keypoints) return fail(FX_ERR_INVALID_ARG, "null argument");
  FX_HIP(hipSetDevice(c->device));
  if (c->last_batch) fxk_pack_kp_records(c->stream, c->dp, c->buf, c->last_batch, dst_device, rec_keypoints);
  FX_HIP(hipGetLastError());
  return FX_OK;
}

size_t fx_keypoint_block_bytes(uint32_t max_scans, uint32_t max_total_keypoints) { return fxk_kp_block_bytes(max_scans, max_total_keypoints); }

fx_status fx_pack_keypoint_block(fx_ctx *c, void *dst_device, uint32_t max_scans, uint32_t max_total_keypoints) {
  if (!c || !dst_device || !max_scans) return fail(FX_ERR_INVALID_ARG, "null argument");
  FX_HIP(hipSetDevice(c->device));
  // a workgroup a scan up to four a CU (the block's zero tail is dealt by stride; an empty batch — kp_offset[0] is 0 — gives
  // the empty block)
  const uint32_t grid = std::max(1u, std::min(c->last_batch, 4u * (uint32_t)c->n_cu));
  fxk_pack_kp_block(c->stream, c->dp, c->buf, c->last_batch, dst_device, max_scans, max_total_keypoints, grid);
  FX_HIP(hipGetLastError());
  return FX_OK;
}

fx_status fx_unpack_pointcloud2(fx_ctx *c, const void *data_device, uint32_t n_points, const fx_pc2_layout *lay,
                                void *dst_device_xyzi) {
  if (!c || !lay || (n_points && (!data_device || !dst_device_xyzi))) return fail(FX_ERR_INVALID_ARG, "null argument");
  const uint32_t offs[3] = {lay->offset_x, lay->offset_y, lay->offset_z};
  for (uint32_t o : offs)
    if (o + 4 > lay->point_step) return fail(FX_ERR_INVALID_ARG, "field offset outside the point record");
  if (lay->offset_intensity != 0xffffffffu && lay->offset_intensity + 4 > lay->point_step)
    return fail(FX_ERR_INVALID_ARG, "intensity offset outside the point record");
  if (((uintptr_t)dst_device_xyzi % 16) != 0) return fail(FX_ERR_INVALID_ARG, "dst must be 16-byte aligned");
  FX_HIP(hipSetDevice(c->device));
  if (n_points)
    fxk_unpack_pc2(c->stream, data_device, n_points, lay->point_step, lay->offset_x, lay->offset_y, lay->offset_z,
                   lay->offset_intensity, lay->is_bigendian, dst_device_xyzi, (uint32_t)c->n_cu * 8u);
  FX_HIP(hipGetLastError());
  return FX_OK;
}

fx_status fx_pack_pointxyzi(fx_ctx *c, uint32_t which, uint32_t scan, void *dst_device, uint32_t capacity_points,
                            uint32_t *n_points_out) {
  if (!c || !dst_device || !n_points_out) return fail(FX_ERR_INVALID_ARG, "null argument");
  if (scan >= c->last_batch) return fail(FX_ERR_INVALID_ARG, "scan index outside the last batch");
  if (((uintptr_t)dst_device % 16) != 0) return fail(FX_ERR_INVALID_ARG, "dst must be 16-byte aligned");
  FX_HIP(hipSetDevice(c->device));
  const FxBuffers &B = c->buf;
  const fx_limits &L = c->lim;
  const float4 *src = nullptr;
  const uint32_t *cnt = nullptr;
  switch (which) {
    case FX_CLOUD_KEYPOINTS: src = B.keypoints + (size_t)scan * L.max_keypoints, cnt = B.n_kp + scan; break;
    case FX_CLOUD_FILTERED: src = B.filt + (size_t)scan * L.max_points, cnt = B.n_filt + scan; break;
    case FX_CLOUD_KEYPOINT_CLOUD: src = B.kpc + (size_t)scan * L.max_kpc_points, cnt = B.n_kpc + scan; break;
    default: return fail(FX_ERR_INVALID_ARG, "unknown cloud selector");
  }
  uint32_t n = 0;
  FX_HIP(hipMemcpyAsync(&n, cnt, 4, hipMemcpyDeviceToHost, c->stream));
  FX_HIP(hipStreamSynchronize(c->stream));
  if (n > capacity_points) return fail(FX_ERR_TOO_LARGE, "destination too small for the cloud");
  if (n) fxk_pack_xyzi32(c->stream, src, n, dst_device, (uint32_t)c->n_cu * 4u);
  FX_HIP(hipGetLastError());
  *n_points_out = n;
  return FX_OK;
}

fx_status fx_pack_features(fx_ctx *c, void *dst_device, uint32_t capacity_records) {
  if (!c || !dst_device) return fail(FX_ERR_INVALID_ARG, "null argument");
  FX_HIP(hipSetDevice(c->device));
  fxk_pack_features(c->stream, c->dp, c->buf, c->last_batch, dst_device, capacity_records, (uint32_t)c->n_cu * 8u);
  FX_HIP(hipGetLastError());
  return FX_OK;
}

}  // extern "C"
