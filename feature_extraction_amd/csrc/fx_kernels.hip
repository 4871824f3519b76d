// fx_kernels.hip — gfx950 kernels of the per-scan detector/descriptor hot path.
//
// Stage map (reference: src/feature_extraction_node.cpp):
//   k_prep       rotateCloud :159-167 + filterCloud :169-183 + getElevationAngles :147-156
//                (elevation only for points that survive the filter: it is dead otherwise)
//   k_bucket     estimateKeypoints ring loop :195-207 (the 16 PassThrough filters as one stable split)
//   k_rings_*    getCylinderSegments :261-327 (one wavefront per ring; workgroup tiers for big rings)
//   k_merge_*    secondary merge :209-257 (+ ring-order assembly of keypoints_full / keypoint_cloud)
//   k_slow       what exceeds every LDS-sized tier, on scratch in HBM; its last workgroup: the batch-wide keypoint offsets
//   k_gather, k_desc_*   estimateDescriptors :329-355 == pcl::ShapeContext3DEstimation
// Every stage has a fast tier sized for the common case and larger tiers fed through device-side
// work lists, so no input is ever truncated silently (flags) and the common case stays small in LDS.
//
// Numerics contract: every result-bearing float expression is evaluated in the operation
// order PCL / FLANN / Eigen use and is never contracted into an FMA (this TU is built with
// -ffp-contract=off and carries the pragma below); fp32 divide / sqrt are the correctly
// rounded hipcc defaults.  Integer results (membership, sizes, orders) are exact.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <type_traits>
#include <algorithm>

#include "fx_device.h"
#include "fx_sort_replay.h"
#include "../../include/fx.h"

#pragma clang fp contract(off)

#define FX_WG 256
#define FX_NWAVE (FX_WG / 64)

namespace {

// two floats an instruction (v_pk_mul_f32 / v_pk_add_f32: IEEE per component, the same bits as the scalar forms)
typedef float fx_f2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------ wave / block helpers
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Stable rank of the threads whose pred is true (order = thread index); total = number of them.
// s_w: FX_NWAVE words of LDS scratch.  Contains two barriers.
// Inclusive prefix sum across a wavefront whose 64 lanes are all active: six DPP adds (row shifts, then the row broadcasts
// of gfx9) — a __shfl_up is a ds_bpermute, an LDS round trip a step, and these scans sit between barriers with the other
// wavefronts waiting.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31
  return x;
}
template <int NT>
__device__ __forceinline__ uint32_t block_rank(bool pred, uint32_t *s_w, uint32_t &total) {
  const unsigned long long m = __ballot(pred);
  const uint32_t wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s_w[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < (NT / 64); ++w) {
    const uint32_t c = s_w[w];
    base += (w < (int)wave) ? c : 0u;
    tot += c;
  }
  __syncthreads();
  total = tot;
  return base + lanes_below(m);
}

// Exclusive prefix sum of v over the block in thread order; total = block sum.
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_w, uint32_t &total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
  inc = wave_incl_scan(inc);
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < (NT / 64); ++w) {
    const uint32_t c = s_w[w];
    base += (w < (int)wave) ? c : 0u;
    tot += c;
  }
  __syncthreads();
  total = tot;
  return base + inc - v;
}

// FLANN L2_Simple<float> over (x,y,z): ((dx*dx) + dy*dy) + dz*dz with d = query - point.
__device__ __forceinline__ float dist2(float qx, float qy, float qz, float px, float py, float pz) {
  const float dx = qx - px, dy = qy - py, dz = qz - pz;
  float r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}

// How many of the points sp[q0 .. q1) (LDS, x y z first in a float4) are closer to (bx, by, bz) than sqrt(r2): dist2's
// arithmetic — FLANN's L2_Simple order, ((dx dx) + dy dy) + dz dz — two points per packed instruction, and counted without
// a compare and an add-with-carry per test: with S a power of two, fma(d2, -S, r2 S) is the exactly scaled difference
// rounded once — positive, zero or negative as r2 - d2 is (-inf when d2 S overflows; never a NaN for finite inputs) — at
// least 2^76 in magnitude unless zero, so the instruction's clamp to [0, 1] gives 1.0f or 0.0f, and the counts add up
// exactly in fp32 (below 2^24).  3DSC's local point density in every descriptor tier.  (r2 >= 1e-30: fx_create.)
struct WithinR2 {
  fx_f2 neg_s, r2s;
  float r2;
  __device__ __forceinline__ explicit WithinR2(float r2_) : r2(r2_) {
    const float kS = __uint_as_float(min(354u - ((__float_as_uint(r2_) >> 23) & 0xffu), 254u) << 23);  // r2 S in [2^100, 2^101)
    neg_s = fx_f2{-kS, -kS};
    r2s = fx_f2{r2_ * kS, r2_ * kS};
  }
  // the two points at f and f + 4 (floats) against b
  __device__ __forceinline__ fx_f2 pair(const float *f, float bx, float by, float bz) const {
    const fx_f2 X = {f[0], f[4]}, Y = {f[1], f[5]}, Z = {f[2], f[6]};
    const fx_f2 dx = bx - X, dy = by - Y, dz = bz - Z;
    fx_f2 r = dx * dx;
    r = r + dy * dy;
    r = r + dz * dz;
    fx_f2 in;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(in) : "v"(r), "v"(neg_s), "v"(r2s));
    return in;
  }
  __device__ __forceinline__ uint32_t count(const float4 *sp, uint32_t q0, uint32_t q1, float bx, float by, float bz) const {
    const float *f = reinterpret_cast<const float *>(sp);
    fx_f2 acc0 = {0.0f, 0.0f}, acc1 = {0.0f, 0.0f};
    uint32_t q = q0;
    for (; q + 3u < q1; q += 4u) {  // (unrolled by hand: the pragma gives up on a loop with inline assembly)
      acc0 = acc0 + pair(f + 4u * q, bx, by, bz);
      acc1 = acc1 + pair(f + 4u * q + 8u, bx, by, bz);
    }
    if (q + 1u < q1) {
      acc0 = acc0 + pair(f + 4u * q, bx, by, bz);
      q += 2u;
    }
    uint32_t n = (uint32_t)(acc0.x + acc1.x) + (uint32_t)(acc0.y + acc1.y);
    if (q < q1) n += dist2(bx, by, bz, f[4u * q], f[4u * q + 1u], f[4u * q + 2u]) < r2 ? 1u : 0u;
    return n;
  }
};

// Global memory written by one wavefront of a workgroup and re-read by another: __syncthreads() orders the stores (they
// are written through to the L2 and acknowledged before the barrier), and an ACQUIRE at agent scope drops the lines this
// CU's L1 may still hold from an earlier read.  (__threadfence() would also RELEASE at agent scope: on gfx950 that writes
// the whole L2 back — paid by every wavefront that executes it; the dense tier spent half its sorting time there.)
__device__ __forceinline__ void wg_global_sync() {
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// ------------------------------------------------------------------ union-find in LDS
// Links always go from the larger root to the smaller one, so a component's root is its
// smallest member index — PCL's "indices[0]" and its discovery order.
// The parent array is always LDS; addressing it through an explicit LDS pointer keeps these
// loops on ds_read / ds_min instead of flat instructions.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
// GS ("global scratch"): the slow tier behind every LDS-sized one (k_slow) runs the same bodies with their per-point and
// per-cluster arrays in a scratch region of HBM instead of LDS — any ring / candidate count the limits allow, at L2 latency.
// Results do not depend on which tier ran.  Barriers there also drop this CU's L1 lines (wg_global_sync).
template <bool GS>
struct UfWord { typedef lds_u32 type; };
template <>
struct UfWord<true> { typedef uint32_t type; };
template <bool GS>
__device__ __forceinline__ void wg_sync() {
  if (GS)
    wg_global_sync();
  else
    __syncthreads();
}
// a wavefront's own hand-over (one lane writes, another reads): LDS is in order within a wavefront; HBM scratch waits for the
// stores and drops the L1's lines
template <bool GS>
__device__ __forceinline__ void wave_sync() {
  if (GS) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}
template <bool GS = false>
__device__ __forceinline__ uint32_t uf_load(const uint32_t *parent, uint32_t i) {
  return ((const volatile typename UfWord<GS>::type *)parent)[i];
}
template <bool GS = false>
__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t i) {
  volatile typename UfWord<GS>::type *p = (volatile typename UfWord<GS>::type *)parent;
  uint32_t q = p[i];
  while (q != i) {
    const uint32_t g = p[q];
    if (g != q) p[i] = g;  // path halving: still an ancestor, so concurrent finds stay valid
    i = q;
    q = g;
  }
  return i;
}
// Read-only variant for the final root pass: there each owner overwrites parent[i] with its root,
// and a concurrent path-halving write from another lane could put a non-root ancestor back.
template <bool GS = false>
__device__ __forceinline__ uint32_t uf_find_ro(uint32_t *parent, uint32_t i) {
  volatile typename UfWord<GS>::type *p = (volatile typename UfWord<GS>::type *)parent;
  uint32_t q = p[i];
  while (q != i) {
    i = q;
    q = p[i];
  }
  return i;
}
template <bool GS = false>
__device__ __forceinline__ void uf_union(uint32_t *parent, uint32_t a, uint32_t b) {
  typename UfWord<GS>::type *p = (typename UfWord<GS>::type *)parent;
  while (true) {
    a = uf_find<GS>(parent, a);
    b = uf_find<GS>(parent, b);
    if (a == b) return;
    if (a < b) {
      const uint32_t t = a;
      a = b;
      b = t;
    }
    const uint32_t old = __hip_atomic_fetch_min(p + a, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (old == a) return;
    a = old;  // a stopped being a root meanwhile: its former parent must join b's set too
  }
}

// The same, from any members (or ancestors) of the two sets; returns the root of the united set as of the call's end.
template <bool GS = false>
__device__ __forceinline__ uint32_t uf_union_root(uint32_t *parent, uint32_t a, uint32_t b) {
  typename UfWord<GS>::type *p = (typename UfWord<GS>::type *)parent;
  while (true) {
    a = uf_find<GS>(parent, a);
    b = uf_find<GS>(parent, b);
    if (a == b) return a;
    if (a < b) {
      const uint32_t t = a;
      a = b;
      b = t;
    }
    const uint32_t old = __hip_atomic_fetch_min(p + a, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (old == a) return b;
    a = old;  // a stopped being a root meanwhile: its former parent must join b's set too
  }
}

// The slow tier's work list (k_slow): a scan is listed once however many of its rings are pending (slow_state), a ring
// handed over is marked in the scan's bit map and counts for nothing until k_slow has done it.  One thread calls these.
__device__ __forceinline__ void slow_push(const FxBuffers &B, uint32_t scan) {
  if (atomicExch(&B.slow_state[scan], 1u) == 0u) B.slow[atomicAdd(&B.counters[FX_CNT_REDO + 1], 1u)] = scan;
}
__device__ __forceinline__ void slow_ring(const FxDevParams &P, const FxBuffers &B, uint32_t scan, uint32_t ring) {
  const uint32_t R = (uint32_t)P.n_rings;
  atomicOr(&B.ring_pending[(size_t)scan * ((R + 31u) / 32u) + (ring >> 5)], 1u << (ring & 31u));
  B.ring_cand_cnt[(size_t)scan * R + ring] = 0u;
  B.kpc_ring_cnt[(size_t)scan * R + ring] = 0u;
  slow_push(B, scan);
}

// Diagnostic build only (-DFX_STAMPS): per-phase cycle shares of the ring kernel, summed by
// thread 0 of every workgroup into B.counters-adjacent debug words.  Never in the product build.
#ifdef FX_STAMPS
#define FX_STAMP(slot)                                                                      \
  do {                                                                                      \
    if (threadIdx.x == 0 && stamps_) {                                                      \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();                         \
      atomicAdd(&stamps_[((blockIdx.x & 63u) << 6) + (slot)], now_ - stamp_prev_);          \
      stamp_prev_ = __builtin_amdgcn_s_memtime();                                           \
    }                                                                                       \
  } while (0)
#define FX_STAMP_INIT(ptr)                 \
  unsigned long long *stamps_ = (ptr);    \
  unsigned long long stamp_prev_ = __builtin_amdgcn_s_memtime()
#define FX_COUNT(slot, v)                                                             \
  do {                                                                                \
    if (stamps_) atomicAdd(&stamps_[((blockIdx.x & 63u) << 6) + (slot)], (unsigned long long)(v)); \
  } while (0)
#else
#define FX_COUNT(slot, v)
#define FX_STAMP(slot)
#define FX_STAMP_INIT(ptr)
#endif
// Diagnostic build only (-DFX_SSTAMPS): the streaming pass's steps on SCALAR registers — s_memtime differences summed in
// uniform variables and written out once at the pass's end.  (The vector-register stamps above make the compiler spill
// the pass's load offsets, and every reload waits for ALL outstanding loads: their split of the pass is an artefact.)
#ifdef FX_SSTAMPS
#define FX_SS_INIT                                                                   \
  unsigned long long ss0_ = 0, ss1_ = 0, ss2_ = 0, ss3_ = 0, ss4_ = 0, ss5_ = 0, ss6_ = 0; \
  unsigned long long ssp_ = __builtin_amdgcn_s_memtime()
#define FX_SS(k)                                                       \
  do {                                                                 \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
    ss##k##_ += now_ - ssp_;                                           \
    ssp_ = now_;                                                       \
  } while (0)
#define FX_SS_FLUSH(ptr)                                                                       \
  do {                                                                                         \
    if (threadIdx.x == 0 && (ptr)) {                                                           \
      unsigned long long *q_ = (ptr) + ((blockIdx.x & 63u) << 6);                              \
      atomicAdd(&q_[24], ss0_), atomicAdd(&q_[30], ss1_), atomicAdd(&q_[25], ss2_), atomicAdd(&q_[26], ss3_); \
      atomicAdd(&q_[27], ss4_), atomicAdd(&q_[28], ss5_), atomicAdd(&q_[29], ss6_);            \
    }                                                                                          \
  } while (0)
#else
#define FX_SS_INIT
#define FX_SS(k)
#define FX_SS_FLUSH(ptr)
#endif

// Scratch words in front of the per-point arrays (NT = workgroup size of the tier):
// [0..15] block helpers, [16..31] broadcast slots, [32..151] sort stack, [160..] segment table
#define FX_SEG_TABLE 160
#define FX_WAVE_QUEUE 128  // work items one wavefront can park before it drains them
template <int NT>
struct SegCfg {
  static constexpr uint32_t kMax = NT == 64 ? 64u : 128u;                        // segments the table holds
  static constexpr uint32_t kQueue = ((FX_SEG_TABLE + 11 * kMax + 2 + 3) / 4) * 4;    // per-wave work queues start here
  static constexpr uint32_t kWords = kQueue + (NT / 64) * FX_WAVE_QUEUE;               // multiple of 4: the carve stays 16-byte aligned
};
#define FX_NONE 0xffffffffu

// order-preserving map float -> uint32 (for LDS atomicMin/Max on coordinates)
__device__ __forceinline__ uint32_t f2ord(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// Segment = up to `seg_len` consecutive points of one run; run = maximal chain of consecutive
// points closer than the tolerance.  The table lives in the scratch words (plain uint32 words,
// floats stored as bits) and is valid when the segment count fits SegCfg<NT>::kMax.
template <int NT>
struct SegTable {
  static constexpr uint32_t M = SegCfg<NT>::kMax;
  uint32_t *w;
  __device__ __forceinline__ explicit SegTable(uint32_t *s_w) : w(s_w + FX_SEG_TABLE) {}
  // segments: first point (start[n_segs] = n), run id, xy bounding box
  __device__ __forceinline__ uint32_t &start(uint32_t i) const { return w[i]; }
  __device__ __forceinline__ uint32_t &run(uint32_t i) const { return w[M + 1 + i]; }
  __device__ __forceinline__ float box(uint32_t which, uint32_t i) const { return __uint_as_float(w[(2 + which) * M + 1 + i]); }
  __device__ __forceinline__ void set_box(uint32_t which, uint32_t i, float v) const { w[(2 + which) * M + 1 + i] = __float_as_uint(v); }
  // runs: first segment (rseg[n_runs] = n_segs), xy bounding box
  __device__ __forceinline__ uint32_t &rseg(uint32_t r) const { return w[6 * M + 1 + r]; }
  __device__ __forceinline__ float rbox(uint32_t which, uint32_t r) const { return __uint_as_float(w[(7 + which) * M + 2 + r]); }
  __device__ __forceinline__ void set_rbox(uint32_t which, uint32_t r, float v) const { w[(7 + which) * M + 2 + r] = __float_as_uint(v); }
};
enum { FX_MINX = 0, FX_MAXX = 1, FX_MINY = 2, FX_MAXY = 3 };

// Stable numbering of the flagged lanes across the block, chunk by chunk: returns the number
// of flags at or before this thread in the current chunk (inclusive) and the chunk total.
template <int NT>
__device__ __forceinline__ uint32_t block_count_incl(bool flag, uint32_t *s_w, uint32_t &total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(flag);
  if (lane == 0) s_w[8 + wave] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t before = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    const uint32_t c = s_w[8 + w];
    before += (w < (int)wave) ? c : 0u;
    tot += c;
  }
  __syncthreads();
  total = tot;
  const unsigned long long incl = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
  return before + (uint32_t)__popcll(m & incl);
}

// Connected components of {d2(i,j) < r2} over n points held in LDS as float4 (x, y, z, *).
// On return parent[i] is the smallest index of i's component, csize[root] the component size,
// rid[i] the run of point i.
//  1. run labelling: consecutive points i-1, i closer than the tolerance form runs; a wave ballot
//     + highest-set-bit gives every point its run head.  Sensor rings arrive azimuth ordered, so
//     this one scan already finds almost every cluster, and only run heads ever get linked.
//     The same pass numbers runs and segments (<= seg_len points of one run) and folds every
//     point into its segment's xy bounding box with LDS atomics.
//  2. cross-run edges: every point against every later run — run box, then the boxes of the run's
//     segments, then the segment's points — skipping runs already in the point's component.  If
//     the segments do not fit the table (unordered input) every point pair of different runs is
//     tested instead.  Either way every pair that could be an edge is examined: exact for any
//     input order.
//  3. roots per run head, then per point; sizes per segment.
// Returns the number of segments (the table is valid iff it is <= SegCfg<NT>::kMax).
template <int NT, bool GS = false>
__device__ __forceinline__ uint32_t cc_label(const float4 *pt, uint32_t n, float r2, uint32_t *parent, uint32_t *csize, uint32_t *rid,
                             uint32_t *s_w, unsigned long long *stamps = nullptr) {
  constexpr uint32_t kSegMax = SegCfg<NT>::kMax;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  FX_STAMP_INIT(stamps);
  const SegTable<NT> ST(s_w);
  const uint32_t seg_len = max(8u, (n + 95u) / 96u);
  for (uint32_t t = threadIdx.x; t < kSegMax; t += NT) {
    ST.set_box(FX_MINX, t, __uint_as_float(f2ord(INFINITY)));
    ST.set_box(FX_MAXX, t, __uint_as_float(f2ord(-INFINITY)));
    ST.set_box(FX_MINY, t, __uint_as_float(f2ord(INFINITY)));
    ST.set_box(FX_MAXY, t, __uint_as_float(f2ord(-INFINITY)));
    ST.set_rbox(FX_MINX, t, __uint_as_float(f2ord(INFINITY)));
    ST.set_rbox(FX_MAXX, t, __uint_as_float(f2ord(-INFINITY)));
    ST.set_rbox(FX_MINY, t, __uint_as_float(f2ord(INFINITY)));
    ST.set_rbox(FX_MAXY, t, __uint_as_float(f2ord(-INFINITY)));
  }
  wg_sync<GS>();
  uint32_t carry = 0, n_runs = 0, n_segs = 0;
  for (uint32_t b0 = 0; b0 < n; b0 += NT) {
    const uint32_t i = b0 + threadIdx.x;
    const bool in = i < n;
    bool start = true;
    float4 q = make_float4(0, 0, 0, 0);
    if (in) {
      q = pt[i];
      csize[i] = 0;
      if (i > 0) {
        const float4 p = pt[i - 1];
        start = !(dist2(q.x, q.y, q.z, p.x, p.y, p.z) < r2);
      }
    }
    const unsigned long long m = __ballot(start);
    if (lane == 0) s_w[wave] = m ? (b0 + wave * 64 + (63u - (uint32_t)__clzll((long long)m))) : FX_NONE;
    wg_sync<GS>();
    const unsigned long long below = m & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    uint32_t head = carry, last = carry;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
      const uint32_t v = s_w[w];
      if (v != FX_NONE) {
        if (w < (int)wave) head = v;
        last = v;
      }
    }
    if (below) head = b0 + wave * 64 + (63u - (uint32_t)__clzll((long long)below));
    wg_sync<GS>();
    carry = last;
    // number the runs and the segments (a segment starts at a run head and every seg_len points)
    uint32_t tot_r, tot_s;
    const uint32_t r_incl = block_count_incl<NT>(in && start, s_w, tot_r);
    const bool seg_start = in && (start || ((i - head) % seg_len) == 0u);
    const uint32_t s_incl = block_count_incl<NT>(seg_start, s_w, tot_s);
    if (in) {
      const uint32_t r = n_runs + r_incl - 1u, sg = n_segs + s_incl - 1u;
      parent[i] = head;
      rid[i] = r;
      if (sg < kSegMax) {
        if (seg_start) {
          ST.start(sg) = i;
          ST.run(sg) = r;
          if (start) ST.rseg(r) = sg;
        }
        const uint32_t ox = f2ord(q.x), oy = f2ord(q.y);
        atomicMin(&ST.w[(2 + FX_MINX) * kSegMax + 1 + sg], ox);
        atomicMax(&ST.w[(2 + FX_MAXX) * kSegMax + 1 + sg], ox);
        atomicMin(&ST.w[(2 + FX_MINY) * kSegMax + 1 + sg], oy);
        atomicMax(&ST.w[(2 + FX_MAXY) * kSegMax + 1 + sg], oy);
      }
    }
    n_runs += tot_r;
    n_segs += tot_s;
  }
  const bool table = n_segs <= kSegMax;
  wg_sync<GS>();
  FX_STAMP(2);
  if (threadIdx.x == 0) {
    FX_COUNT(12, 1);
    FX_COUNT(13, n_runs);
    FX_COUNT(14, n_segs);
  }
  if (table) {
    // segment boxes -> floats, folded into the run boxes
    for (uint32_t sg = threadIdx.x; sg < n_segs; sg += NT) {
      const uint32_t r = ST.run(sg);
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t o = ST.w[(2 + k) * kSegMax + 1 + sg];
        ST.w[(2 + k) * kSegMax + 1 + sg] = __float_as_uint(ord2f(o));
        if (k & 1)
          atomicMax(&ST.w[(7 + k) * kSegMax + 2 + r], o);
        else
          atomicMin(&ST.w[(7 + k) * kSegMax + 2 + r], o);
      }
    }
    if (threadIdx.x == 0) {
      ST.start(n_segs) = n;
      ST.rseg(n_runs) = n_segs;
    }
    wg_sync<GS>();
    for (uint32_t r = threadIdx.x; r < n_runs; r += NT) {
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) ST.w[(7 + k) * kSegMax + 2 + r] = __float_as_uint(ord2f(ST.w[(7 + k) * kSegMax + 2 + r]));
    }
    wg_sync<GS>();
  }
  if (n_runs > 1) {
    // Candidate generation is wave-uniform and cheap; the expensive part (finds, segment and
    // point scans, unions) would serialise the wavefront if done in place, one lane at a time.
    // So each wavefront parks its (point, run) / (point, point) work items in a small LDS queue
    // and drains it with all lanes busy on different items.
    uint32_t *wq = s_w + SegCfg<NT>::kQueue + wave * FX_WAVE_QUEUE;
    uint32_t wq_n = 0;
    const float r2_pad = r2 * 1.001f;  // box distances are lower bounds; pad them against fp32 rounding
    auto drain = [&]() {
      wave_sync<GS>();
      for (uint32_t t = lane; t < wq_n; t += 64) {
        const uint32_t item = wq[t];
        const uint32_t i = item >> 16, x = item & 0xffffu;
        if (!table) {
          uf_union<GS>(parent, x, i);  // (i, j) is an edge
          continue;
        }
        // (point i, other run x): its segments' boxes, then their points
        const uint32_t s0 = ST.rseg(x), s1 = ST.rseg(x + 1);
        if (uf_find<GS>(parent, i) == uf_find<GS>(parent, ST.start(s0))) continue;  // already one component
        const float4 q = pt[i];
        bool linked = false;
        for (uint32_t sg = s0; sg < s1 && !linked; ++sg) {
          const float dx = fmaxf(fmaxf(ST.box(FX_MINX, sg) - q.x, q.x - ST.box(FX_MAXX, sg)), 0.0f);
          const float dy = fmaxf(fmaxf(ST.box(FX_MINY, sg) - q.y, q.y - ST.box(FX_MAXY, sg)), 0.0f);
          if (dx * dx + dy * dy > r2_pad) continue;
          for (uint32_t j = ST.start(sg); j < ST.start(sg + 1); ++j) {
            const float4 p = pt[j];
            if (dist2(q.x, q.y, q.z, p.x, p.y, p.z) < r2) {
              uf_union<GS>(parent, j, i);
              linked = true;  // the two runs are one component now; more edges add nothing
              break;
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      wq_n = 0;
    };
    auto park = [&](bool has, uint32_t item) {
      const unsigned long long m = __ballot(has);
      if (m) {  // wave-uniform: nothing to do in the common case
        if (has) wq[wq_n + lanes_below(m)] = item;
        wq_n += (uint32_t)__popcll(m);
        if (wq_n > FX_WAVE_QUEUE - 64) drain();
      }
    };
    if (table) {
      // Run pairs whose boxes come within the tolerance: azimuth-ordered rings have few or none, so
      // listing them first (one box-box test per pair, pairs spread over all lanes) replaces
      // the points x runs sweep.  The list borrows the sort stack, which is idle until cc_order.
      uint32_t *rp = s_w + 32;
      constexpr uint32_t kPairCap = FX_SORT_STACK_WORDS;
      if (threadIdx.x == 0) s_w[16] = 0;
      wg_sync<GS>();
      const float inv_runs = 1.0f / (float)n_runs;
      for (uint32_t p = threadIdx.x; p < n_runs * n_runs; p += NT) {
        // p = a * n_runs + b; (p + 0.5) / n_runs is never within rounding distance of an integer (p < 2^14)
        const uint32_t a = (uint32_t)(((float)p + 0.5f) * inv_runs), b = p - a * n_runs;
        if (b <= a) continue;
        const float dx = fmaxf(fmaxf(ST.rbox(FX_MINX, b) - ST.rbox(FX_MAXX, a), ST.rbox(FX_MINX, a) - ST.rbox(FX_MAXX, b)), 0.0f);
        const float dy = fmaxf(fmaxf(ST.rbox(FX_MINY, b) - ST.rbox(FX_MAXY, a), ST.rbox(FX_MINY, a) - ST.rbox(FX_MAXY, b)), 0.0f);
        if (dx * dx + dy * dy > r2_pad) continue;
        const uint32_t slot = atomicAdd(&s_w[16], 1u);
        if (slot < kPairCap) rp[slot] = (a << 16) | b;
      }
      wg_sync<GS>();
      const uint32_t n_rp = s_w[16];
      if (threadIdx.x == 0) {
        FX_COUNT(15, n_rp);
      }
      if (n_rp <= kPairCap) {
        // the points of the earlier run against the later run's box, one near pair per wavefront at a time
        for (uint32_t t = wave; t < n_rp; t += NT / 64) {
          const uint32_t a = rp[t] >> 16, b = rp[t] & 0xffffu;
          const uint32_t i_end = ST.start(ST.rseg(a + 1));
          for (uint32_t i0 = ST.start(ST.rseg(a)); i0 < i_end; i0 += 64) {
            const uint32_t i = i0 + lane;
            bool ok = false;
            if (i < i_end) {
              const float4 q = pt[i];
              const float dx = fmaxf(fmaxf(ST.rbox(FX_MINX, b) - q.x, q.x - ST.rbox(FX_MAXX, b)), 0.0f);
              const float dy = fmaxf(fmaxf(ST.rbox(FX_MINY, b) - q.y, q.y - ST.rbox(FX_MAXY, b)), 0.0f);
              ok = !(dx * dx + dy * dy > r2_pad);
            }
            park(ok, (i << 16) | b);
          }
        }
      } else {
        // every point against the box of every later run; the run loop is wave-uniform (LDS
        // broadcast reads, four boxes in flight per trip)
        for (uint32_t i0 = 0; i0 < n; i0 += NT) {
          const uint32_t i = i0 + threadIdx.x;
          const bool live = i < n;
          const float4 q = live ? pt[i] : make_float4(0, 0, 0, 0);
          const uint32_t my_run = live ? rid[i] : FX_NONE;
          // runs before the smallest run id of this wave cannot be "later" for any lane
          uint32_t r_lo = my_run;
#pragma unroll
          for (int d = 32; d > 0; d >>= 1) r_lo = min(r_lo, (uint32_t)__shfl_xor((int)r_lo, d, 64));
          if (r_lo == FX_NONE) continue;
          for (uint32_t r0 = r_lo + 1; r0 < n_runs; r0 += 4) {
            uint32_t near = 0;
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
              const uint32_t r = min(r0 + u, n_runs - 1);
              const float dx = fmaxf(fmaxf(ST.rbox(FX_MINX, r) - q.x, q.x - ST.rbox(FX_MAXX, r)), 0.0f);
              const float dy = fmaxf(fmaxf(ST.rbox(FX_MINY, r) - q.y, q.y - ST.rbox(FX_MAXY, r)), 0.0f);
              const bool ok = (r0 + u < n_runs) && (r0 + u > my_run) && !(dx * dx + dy * dy > r2_pad);
              near |= ok ? (1u << u) : 0u;
            }
            if (__ballot(near != 0u)) {  // wave-uniform and rare
#pragma unroll
              for (uint32_t u = 0; u < 4; ++u) park((near >> u) & 1u, (i << 16) | (r0 + u));
            }
          }
        }
      }
    } else {
      // every pair (i, j > i) of different runs: each lane keeps its point i in registers and all
      // lanes walk j together (LDS broadcast reads), a dozen instructions per 64 pairs
      for (uint32_t i0 = 0; i0 < n; i0 += NT) {
        const uint32_t i = i0 + threadIdx.x;
        const bool live = i < n;
        const float4 q = live ? pt[i] : make_float4(0, 0, 0, 0);
        const uint32_t my_run = live ? rid[i] : FX_NONE;
        // (eight points per trip, loaded before any of them is used: a trip costs one LDS round trip, not eight)
        uint32_t j = i0 + wave * 64 + 1;
        for (; j + 8 <= n; j += 8) {
          float4 p[8];
          uint32_t rj[8];
#pragma unroll
          for (uint32_t u = 0; u < 8; ++u) {
            p[u] = pt[j + u];
            rj[u] = rid[j + u];
          }
          uint32_t edges = 0;
#pragma unroll
          for (uint32_t u = 0; u < 8; ++u) {
            const bool edge = live && j + u > i && rj[u] != my_run && dist2(q.x, q.y, q.z, p[u].x, p[u].y, p[u].z) < r2;
            edges |= edge ? (1u << u) : 0u;
          }
          if (__ballot(edges != 0u)) {  // wave-uniform
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) park((edges >> u) & 1u, (i << 16) | (j + u));
          }
        }
        for (; j < n; ++j) {
          const float4 p = pt[j];
          const bool edge = live && j > i && rid[j] != my_run && dist2(q.x, q.y, q.z, p.x, p.y, p.z) < r2;
          park(edge, (i << 16) | j);
        }
      }
    }
    drain();
  }
  wg_sync<GS>();
  FX_STAMP(3);
  // roots: run heads first (only heads are ever linked), then every point through its head
  for (uint32_t i = threadIdx.x; i < n; i += NT) {
    const bool is_head = i == 0 || rid[i] != rid[i - 1];
    if (is_head) parent[i] = uf_find_ro<GS>(parent, i);
  }
  wg_sync<GS>();
  for (uint32_t i = threadIdx.x; i < n; i += NT) {
    const bool is_head = i == 0 || rid[i] != rid[i - 1];
    if (!is_head) parent[i] = parent[parent[i]];
  }
  wg_sync<GS>();
  if (table) {
    for (uint32_t sg = threadIdx.x; sg < n_segs; sg += NT) atomicAdd(&csize[parent[ST.start(sg)]], ST.start(sg + 1) - ST.start(sg));
  } else {
    for (uint32_t i = threadIdx.x; i < n; i += NT) atomicAdd(&csize[parent[i]], 1u);
  }
  wg_sync<GS>();
  FX_STAMP(4);
  return n_segs;
}

// Phase 1 of the cluster-order replay (std::__introsort_loop, csrc/fx_sort_replay.h) by ONE WAVEFRONT:
// every partition step costs a fixed handful of LDS round trips instead of one per element visited.
// The step follows the position-list rule proved in fx_sort_replay.h (partition_pivot_lists): with
// L = positions holding an element not smaller than the pivot (ascending) and R = positions holding one
// not larger (descending), the sequential loop swaps exactly the pairs (L_k, R_k) with L_k < R_k and cuts
// at min(L_s, R_{s-1}).  Ballots give every lane the rank of its positions in L and R; two small position
// tables pair them up.  Lane p + 64 w owns view position p + 64 w (n <= 64 W).  All 64 lanes must be here.
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int W, bool GS = false>
__device__ __noinline__ void sort_partition_wave(uint32_t *crec, int n, int *stk, uint16_t *Lpos, uint16_t *Rpos) {
  using namespace fx_sort_detail;
  if (n <= FX_SORT_THRESHOLD) return;
  const int lane = (int)(threadIdx.x & 63);
  RevView v{crec, n};
  const unsigned long long below = lane == 63 ? ~0ull >> 1 : ((1ull << lane) - 1ull);  // lanes strictly below
  const unsigned long long above = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);       // lanes strictly above
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) ++lg;
  int *stk_first = stk, *stk_last = stk + 40, *stk_depth = stk + 80;
  if (lane == 0) {
    stk_first[0] = 0;
    stk_last[0] = n;
    stk_depth[0] = 2 * lg;
  }
  wave_sync<GS>();
  int sp = 1;
  while (sp > 0) {
    --sp;
    int first = stk_first[sp], last = stk_last[sp], depth = stk_depth[sp];
    while (last - first > FX_SORT_THRESHOLD) {
      if (depth == 0) {  // depth budget spent (adversarial input): heap sort, sequential
        if (lane == 0) heap_sort(v, first, last);
        wave_sync<GS>();
        break;
      }
      --depth;
      if (lane == 0) median_to_first(v, first, first + 1, first + (last - first) / 2, last - 1);
      wave_sync<GS>();
      const uint32_t pivot = v.get(first);
      unsigned long long mL[W], mR[W];
      bool isL[W], isR[W];
      int nL = 0, nR = 0;
#pragma unroll
      for (int w = 0; w < W; ++w) {
        const int p = w * 64 + lane;
        const uint32_t a = p < n ? v.get(p) : 0u;
        isL[w] = p > first && p < last && !less_size(a, pivot);
        isR[w] = p >= first && p < last && !less_size(pivot, a);
        mL[w] = __ballot(isL[w]);
        mR[w] = __ballot(isR[w]);
        nL += __popcll(mL[w]);
        nR += __popcll(mR[w]);
      }
      int rankL[W], rankR[W];
#pragma unroll
      for (int w = 0; w < W; ++w) {
        int bl = 0, ar = 0;
#pragma unroll
        for (int x = 0; x < W; ++x) {
          if (x < w) bl += __popcll(mL[x]);
          if (x > w) ar += __popcll(mR[x]);
        }
        rankL[w] = bl + __popcll(mL[w] & below);
        rankR[w] = ar + __popcll(mR[w] & above);
        if (isL[w]) Lpos[rankL[w]] = (uint16_t)(w * 64 + lane);
        if (isR[w]) Rpos[rankR[w]] = (uint16_t)(w * 64 + lane);
      }
      wave_sync<GS>();
      int partner[W], s = 0;
      uint32_t incoming[W];
#pragma unroll
      for (int w = 0; w < W; ++w) {
        const int p = w * 64 + lane;
        partner[w] = -1;
        bool as_left = false;
        if (isL[w] && rankL[w] < nR) {
          const int r = Rpos[rankL[w]];
          if (p < r) partner[w] = r, as_left = true;
        }
        if (isR[w] && rankR[w] < nL) {
          const int l = Lpos[rankR[w]];
          if (l < p) partner[w] = l;  // (a position never swaps in both roles)
        }
        incoming[w] = partner[w] >= 0 ? v.get(partner[w]) : 0u;
        s += __popcll(__ballot(as_left));
      }
      wave_sync<GS>();
#pragma unroll
      for (int w = 0; w < W; ++w)
        if (partner[w] >= 0) v.set(w * 64 + lane, incoming[w]);
      int cut = 0x7fffffff;
      if (s < nL) cut = Lpos[s];
      if (s >= 1) cut = min(cut, (int)Rpos[s - 1]);
      if (lane == 0) {
        stk_first[sp] = cut;
        stk_last[sp] = last;
        stk_depth[sp] = depth;
      }
      wave_sync<GS>();
      ++sp;
      last = cut;
    }
  }
}

// Size-admissible components in discovery order (ascending smallest index), then PCL's final
// std::sort(rbegin, rend, bySize): its partition phase is replayed by one wavefront (only needed
// above 16 clusters; one lane, sequentially, beyond 192), its insertion phase — a stable sort — as a parallel ranking
// (csrc/fx_sort_replay.h).  crec[s] = (size << 16) | discovery ordinal, in the order PCL returns
// the clusters; croot[ordinal] = root index; tmp: scratch.  croot / crec / tmp hold ccap entries;
// returns the cluster count, which the caller must check against ccap (nothing is written past it,
// and nothing is ordered, when it does not fit).
template <int NT, bool GS = false>
__device__ __forceinline__ uint32_t cc_order(uint32_t n, const uint32_t *parent, const uint32_t *csize, uint32_t min_sz,
                             uint32_t max_sz, uint32_t *croot, uint32_t *crec, uint32_t *tmp, uint32_t ccap,
                             uint32_t *s_w, unsigned long long *stamps = nullptr) {
  FX_STAMP_INIT(stamps);
  uint32_t n_c = 0;
  for (uint32_t b0 = 0; b0 < n; b0 += NT) {
    const uint32_t i = b0 + threadIdx.x;
    bool acc = false;
    uint32_t sz = 0;
    if (i < n && parent[i] == i) {
      sz = csize[i];
      acc = sz >= min_sz && sz <= max_sz;
    }
    uint32_t tot;
    const uint32_t r = block_rank<NT>(acc, s_w, tot);
    if (acc) {
      const uint32_t c = n_c + r;
      if (c < ccap) {
        croot[c] = i;
        crec[c] = (sz << 16) | c;
      }
    }
    n_c += tot;
  }
  wg_sync<GS>();
  FX_STAMP(5);
  if (n_c > ccap) return n_c;
  if (n_c > FX_SORT_THRESHOLD) {
    if (threadIdx.x < 64) {  // one wavefront; tmp doubles as the two position tables
      uint16_t *pos = reinterpret_cast<uint16_t *>(tmp);
      if (n_c <= 64) {
        sort_partition_wave<1, GS>(crec, (int)n_c, (int *)(s_w + 32), pos, pos + n_c);
      } else if (n_c <= 128) {
        sort_partition_wave<2, GS>(crec, (int)n_c, (int *)(s_w + 32), pos, pos + n_c);
      } else if (n_c <= 192) {
        sort_partition_wave<3, GS>(crec, (int)n_c, (int *)(s_w + 32), pos, pos + n_c);
      } else if (n_c <= 256) {
        sort_partition_wave<4, GS>(crec, (int)n_c, (int *)(s_w + 32), pos, pos + n_c);
      } else if (threadIdx.x == 0) {  // more clusters than three words of lanes: one lane, sequentially
        fx_sort_detail::RevView v{crec, (int)n_c};
        fx_sort_partition_phase(v, (int)n_c, (int *)(s_w + 32));
      }
    }
    wg_sync<GS>();
  }
  if (n_c > 1) {
    for (uint32_t c = threadIdx.x; c < n_c; c += NT) {
      const uint32_t rec = crec[c], sz = rec >> 16;
      uint32_t pos = 0;
#pragma unroll 8
      for (uint32_t d = 0; d < n_c; ++d) {
        const uint32_t sd = crec[d] >> 16;
        pos += (sd > sz || (sd == sz && d < c)) ? 1u : 0u;
      }
      tmp[pos] = rec;
    }
    wg_sync<GS>();
    for (uint32_t c = threadIdx.x; c < n_c; c += NT) crec[c] = tmp[c];
    wg_sync<GS>();
  }
  FX_STAMP(6);
  return n_c;
}

}  // namespace

// ====================================================================== stage 1: prep
// One workgroup per scan streams the scan once: rotate (fp32, PCL's scalar order), apply
// the three PassThrough predicates at once and compact the survivors of a tile, in input
// order (wave ballot + prefix), into an LDS buffer.  The elevation angle (fp64) is computed
// by dense sweeps over the buffered survivors only — about one point in ten survives, and in
// firing order the survivors are spread over every wavefront.
// The same pass records, one bit per 64 consecutive points, whether any of them lies within the descriptor
// stage's reach of the filter box (every keypoint is a centroid of filtered points, hence inside the box; a point
// farther than the support radius from the box cannot support any keypoint): k_gather skips the others unread.
#ifndef FX_PREP_T
#define FX_PREP_T 512
#endif
#define FX_MAX_RINGS 1024  // fx_create checks n_rings against it
#ifndef FX_PREP_U
#define FX_PREP_U 4
#endif
static_assert(FX_PREP_U % 2 == 0, "the tile's points are rotated in pairs");
#define FX_PREP_TILE (FX_PREP_T * FX_PREP_U)  // 2048 points = 32 groups of 64 = one word of near bits
typedef float __attribute__((address_space(1))) gfloat;

// getElevationAngles (ref: node.cpp:147-156): az = atan2(y, x); xp = cos(az) x + sin(az) y;
// intensity = atan2(z, xp) * 180 / M_PI in double, stored as float.  cos(az) x + sin(az) y is |xy| up to a few
// ulp of double, so atan2(z, sqrt(x^2 + y^2)) rounds to the same float unless it falls within that error of a
// float rounding boundary; only then (about one point in 10^4) is the reference's own expression evaluated.
// (out of line: the fp64 atan2 / sin / cos code then costs k_prep 86 registers instead of 128 — the same speed alone,
//  3 % more throughput with four batches in flight, where the registers go to other batches' kernels)
__device__ __noinline__ float elevation_deg(float xf, float yf, float zf) {
  const double x = xf, y = yf, z = zf;
  const double e = atan2(z, sqrt(x * x + y * y)) * 180 / M_PI;
  const float f = (float)e;
  const float af = fabsf(f);
  // distance from e to the nearer rounding boundary of f, with the smaller of f's two ulps (conservative)
  const double ulp = (double)af - (double)__uint_as_float(__float_as_uint(af) - 1u);
  const double slack = 0.5 * ulp - fabs(e - (double)f);
  if (af >= FLT_MIN && af < INFINITY && slack > 1e-11 * fabs(e)) return f;
  const double az = atan2(y, x);
  const double xp = cos(az) * x + sin(az) * y;
  return (float)(atan2(z, xp) * 180 / M_PI);
}

// The same value for all but one point in sixteen thousand at a fifth of the instructions (the library's fp64 atan2 was 4 % of
// the whole batch's vector instructions): t = z / |xy| from a refined fp32 reciprocal square root,
// atan(|t|) from the degree-6 expansion about the nearest multiple of 1/64 (tab: FxBuffers::atan_tab in LDS), good to
// 2^-44 relative (expansion 2^-51.8 absolute on values >= 2^-7; the root and the quotient a few ulp of double) where the
// reference's own double, evaluated by the host's libm, is within a few ulp of the true value.  The float it rounds to
// is therefore the reference's unless the double lies within 2^14 of its ulps (2^-39 relative: a margin of 32) of the
// midpoint of two floats — the only place where round-to-nearest changes; such points, and whatever the expansion
// does not cover (|t| > 1, a denormal or huge |xy|^2, results below the normal floats), return false and take
// elevation_deg.  (tests/test_gpu_elevation.py compares the two paths on 10^7 points.)
__device__ __forceinline__ bool elevation_fast(float xf, float yf, float zf, const double *tab, float &out) {
  const double x = xf, y = yf, z = zf;
  const double s = x * x + y * y;  // (the products are exact)
  const float sf = (float)s;
  if (!(sf > 1e-30f && sf < 1e30f)) return false;
  const double y0 = (double)__builtin_amdgcn_rsqf(sf);  // 2^-22
  const double e = fma(-s, y0 * y0, 1.0);               // 1 - s y0^2, |e| < 2^-20
  const double y1 = fma(y0 * e, fma(0.375, e, 0.5), y0);  // y0 (1 + e/2 + 3 e^2/8): 1/sqrt(s) to 5/16 e^3
  const double t = z * y1;
  const double ta = fabs(t);
  if (!(ta <= 1.0)) return false;  // (also a NaN z)
  const float ti = rintf((float)ta * (float)FX_ATAN_N);
  const double d = ta - (double)ti * (1.0 / FX_ATAN_N);
  const double *a = tab + (FX_ATAN_DEG + 1) * (int)ti;
  double r = a[FX_ATAN_DEG];
#pragma unroll
  for (int kk = FX_ATAN_DEG - 1; kk >= 0; --kk) r = fma(r, d, a[kk]);
  const double deg = copysign(r, t) * 57.295779513082320876798154814105;
  const unsigned long long bits = (unsigned long long)__double_as_longlong(deg);
  if (((bits >> 52) & 0x7ffu) < 1023u - 100u) {  // |deg| < 2^-100: z = +-0 gives the reference's +-0; the rest is the exact path's
    out = (float)deg;
    return deg == 0.0;
  }
  const uint32_t dropped = (uint32_t)bits & 0x1fffffffu;  // the 29 bits the conversion rounds away; the midpoint is 2^28
  const uint32_t dist = dropped > 0x10000000u ? dropped - 0x10000000u : 0x10000000u - dropped;
  out = (float)deg;
  return dist > (1u << 14);
}

// rings a point belongs to (ref: node.cpp:200-201): used by k_prep (counts) and k_bucket (the split)
__device__ __forceinline__ uint32_t ring_membership(float el, const float2 *win, int n_rings, float el0, float inv_step,
                                                    int &r_first) {
  // candidate rings: the nearest centre and its two neighbours; membership by the exact windows
  const float t = (el - el0) * inv_step + 0.5f;
  int r0 = (t > -4.0f && t < 1.0e6f) ? (int)floorf(t) : -4;
  r_first = r0 - 1;
  uint32_t mask = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int r = r_first + d;
    if (r >= 0 && r < n_rings) {
      const float2 w = win[r];
      if (!(el < w.x || el > w.y)) mask |= 1u << d;
    }
  }
  return mask;
}

// The streaming pass's rotate-and-test, ONE statement for every kernel that needs it (prep_stream, and k_prep_count, whose
// counts the sliced pass's bases are: the two must agree bit for bit — ADVICE r5).  T = float, or fx_f2 for two points an
// instruction (packed fp32 is IEEE per component: the same bits).
// pcl::transformPointCloud, dense branch: ((m0 x + m1 y) + m2 z) + t, t = 0 (the + 0 only turns -0 into +0, which no range
// test sees; the sweep, whose values are stored, adds it).
template <typename T>
__device__ __forceinline__ void prep_rotate(const float *R, T x, T y, T z, T &rx, T &ry, T &rz) {
  rx = (R[0] * x + R[1] * y) + R[2] * z;
  ry = (R[3] * x + R[4] * y) + R[5] * z;
  rz = (R[6] * x + R[7] * y) + R[8] * z;
}
// Range tests with both ends clamped to the finite floats: a NaN or an infinite coordinate fails them (PassThrough drops
// non-finite points first, ref: SURVEY.md A.2), a finite one compares as in PCL's !(v < min || v > max); an infinite or
// NaN limit (no limit on that side; a NaN limit compares false in PCL too) becomes +-FLT_MAX.
struct PrepBox {
  float x0, x1, y0, y1, z0, z1;
  __device__ __forceinline__ PrepBox(const FxDevParams &P, float margin)
      : x0(fmaxf(P.x_min - margin, -FLT_MAX)), x1(fminf(P.x_max + margin, FLT_MAX)), y0(fmaxf(P.y_min - margin, -FLT_MAX)),
        y1(fminf(P.y_max + margin, FLT_MAX)), z0(fmaxf(P.z_min - margin, -FLT_MAX)), z1(fminf(P.z_max + margin, FLT_MAX)) {}
  // (the filter box itself: no arithmetic on the limits — P.x_min - 0.0f would turn a -0 limit... into the same -0: exact)
  __device__ __forceinline__ explicit PrepBox(const FxDevParams &P)
      : x0(fmaxf(P.x_min, -FLT_MAX)), x1(fminf(P.x_max, FLT_MAX)), y0(fmaxf(P.y_min, -FLT_MAX)), y1(fminf(P.y_max, FLT_MAX)),
        z0(fmaxf(P.z_min, -FLT_MAX)), z1(fminf(P.z_max, FLT_MAX)) {}
  __device__ __forceinline__ bool has(float rx, float ry, float rz) const {
    return rx >= x0 && rx <= x1 && ry >= y0 && ry <= y1 && rz >= z0 && rz <= z1;
  }
};

#ifndef FX_PREP_OCC
#define FX_PREP_OCC 4  // waves per SIMD the register budget is held to: 4 = two workgroups per CU (129 registers would mean one)
#endif
#ifndef FX_PREP_KEEP
#define FX_PREP_KEEP (FX_PREP_TILE + FX_PREP_TILE / 2)  // survivors buffered between sweeps
#endif
// LDS of the streaming pass (static arrays in k_prep, carved from the dynamic image in k_front)
struct PrepLds {
  uint32_t *cnt;       // [2][FX_PREP_T / 64] per wave: survivors of the tile; by tile parity
  float *keep;         // [3 * FX_PREP_KEEP] un-rotated survivors (x, y, z) waiting for the elevation sweep
  uint32_t *ring;      // [n_rings] survivors per ring (a window-boundary point counts in both rings)
  double *atan;        // [(FX_ATAN_N + 1) * (FX_ATAN_DEG + 1)] elevation_fast's table
  // the ring windows (a sweep then loads nothing from global memory: the wait for such a load would also be one for the
  // acknowledgement of the sweep's earlier stores — vmcnt counts loads and stores in issue order)
  float2 *win;         // [n_rings]
  float *el;           // KEEP only: [keep_cap] the survivors' elevation angles
  uint32_t keep_cap;   // KEEP only: survivors `keep` holds
};
// The streaming pass over one scan (n > 0) by one FX_PREP_T-thread workgroup: writes the filtered cloud and the near bits,
// leaves the ring counts in L.ring; returns the number of survivors.  Ends with a barrier.
// KEEP (k_front): the survivors STAY in the buffer — un-rotated x y z in L.keep, elevations in L.el, in input order —: the
// ring split then needs nothing from HBM (reading ~cloud back cost a wait for the sweeps' stores, an L2 round trip and
// 50 MB of HBM traffic a batch).  A scan with more survivors than L.keep_cap makes the pass return FX_NONE at the tile that
// no longer fits (the caller runs the recycling instance over the whole scan again: rare, and such a scan is
// k_front_redo's anyway).
// (t_begin, t_end, base0: the points [t_begin, t_end) of the scan only — t_begin a multiple of the tile —, their survivors
//  written from position base0 of the filtered cloud on: one of several workgroups of a scan, k_prep_sliced)
template <bool KEEP = false>
__device__ __forceinline__ uint32_t prep_stream(const FxDevParams &P, const FxBuffers &B, const FxScanMeta &M, uint32_t scan,
                                                float near_margin, float el0, float inv_step, const PrepLds &L, uint32_t t_begin = 0u,
                                                uint32_t t_end = 0xffffffffu, uint32_t base0 = 0u) {
  constexpr int NW = FX_PREP_T / 64;
  constexpr uint32_t kTile = FX_PREP_TILE;           // points per tile; wave w owns [256 w, 256 w + 256) of it
  constexpr uint32_t kKeep = FX_PREP_KEEP;
  uint32_t swept = 0;            // KEEP: survivors of the buffer the sweep has done
  uint32_t *const s_cnt = L.cnt;
  float *const s_keep = L.keep;
  uint32_t *const s_ring = L.ring;
  double *const s_atan = L.atan;
  float2 *const s_win = L.win;
  float4 *out = B.filt + (size_t)scan * P.max_points;
  uint32_t *near_bits = B.near_bits + (size_t)scan * P.near_words;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t base = base0, buffered = 0, parity = 0;
  const uint32_t n = min(M.n, t_end);
  // (global address space stated: a generic-pointer load would be a flat load, which also counts as an
  //  LDS access and gets waited for at the next LDS instruction)
  const gfloat *gpts = (const gfloat *)M.pts;
  const uint32_t R = (uint32_t)P.n_rings;
  FX_STAMP_INIT(B.stamps);
  FX_SS_INIT;
  for (uint32_t r = tid; r < R; r += FX_PREP_T) s_ring[r] = 0u;  // (ordered before the first sweep by the tile barriers)
  for (uint32_t r = tid; r < (FX_ATAN_N + 1) * (FX_ATAN_DEG + 1); r += FX_PREP_T) s_atan[r] = B.atan_tab[r];
  for (uint32_t r = tid; r < R; r += FX_PREP_T) s_win[r] = B.ring_win[r];
  // the loads of the next tile are issued before this tile's barrier, so the memory pipe stays full
  // while the tile is compacted
  // A tile's loads: a uniform base (the tile's first record: scalar registers) plus a 32-bit byte offset per lane, clamped to
  // the scan's last record (no branch around the load; only the last tile's lanes past the end are clamped) — one 12-byte
  // load a point, no 64-bit address arithmetic per load.  (The scan's table entry comes in through vector loads — the
  // compiler cannot know nothing writes it —: its fields are made scalar by hand.)
  typedef float gf3 __attribute__((ext_vector_type(3), aligned(4)));
  typedef float gf2 __attribute__((ext_vector_type(2), aligned(4)));
  typedef float gf4 __attribute__((ext_vector_type(4), aligned(4)));
  typedef const __attribute__((address_space(1))) char *gbytes;
  const uint32_t stride_b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(M.stride_f * 4u));  // (a scan is < 2^28 bytes: fx_create's limits)
  const gbytes gbase = (gbytes)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)((unsigned long long)gpts >> 32)) << 32) |
                                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(unsigned long long)gpts));
  const uint32_t n_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
  const uint32_t lane_byte = (wave * (64 * FX_PREP_U) + lane) * stride_b;
  auto load_tile = [&](uint32_t t0, float4 (&v)[FX_PREP_U], int u_begin = 0, int u_end = FX_PREP_U) {
    // (the tile after the last is loaded ahead and never used: every lane of it re-reads the scan's last record — one path,
    //  so that the loads land in the registers the tile is computed from: two paths meet in copies, which wait for the loads)
    const uint32_t t0c = min(t0, n_s - 1u);
    const gbytes tb = gbase + (size_t)t0c * stride_b;
    const uint32_t last = (n_s - 1u - t0c) * stride_b;  // the scan's last record, from the tile's first
#pragma unroll
    for (int u = 0; u < FX_PREP_U; ++u) {
      if (u < u_begin || u >= u_end) continue;
      const uint32_t off = lane_byte + (uint32_t)(u * 64) * stride_b;
#if defined(FX_TILE_X4)  // (experiment: the whole 16-byte record)
      const gf4 w = *(const __attribute__((address_space(1))) gf4 *)(tb + min(off, last));
      v[u] = make_float4(off <= last && t0 < n_s ? w.x : NAN, w.y, w.z, 0.f);
#elif !defined(FX_TILE_PAIR)
      // ONE 12-byte load a point: the memory pipeline takes a wavefront's strided load in ~50 cycles whatever its width —
      // the 4 + 8-byte pair the compiler makes of a conditional x costs two (k_front alone 0.305 -> 0.297 ms, one scan
      // 0.106 -> 0.101, headline +2 %; round 1's "12 against 16 bytes" compared the pair, not this)
      const gf3 w = *(const __attribute__((address_space(1))) gf3 *)(tb + min(off, last));
      // past the end: a NaN x makes all three rotated coordinates NaN, which fail every range test below
      v[u] = make_float4(off <= last && t0 < n_s ? w.x : NAN, w.y, w.z, 0.f);
#else
      //  (the x load under its own condition: unconditional, the compiler fuses the pair into the 12-byte load)
      const gbytes q = tb + min(off, last);
      float x = NAN;
      if (off <= last && t0 < n_s) x = *(const gfloat *)q;
      const gf2 yz = *(const __attribute__((address_space(1))) gf2 *)(q + 4);
      v[u] = make_float4(x, yz.x, yz.y, 0.f);
#endif
    }
  };
  // The fp64 elevation is a long dependent chain: it runs over the buffered survivors only when the buffer could
  // overflow on the next tile, with every lane busy, instead of after each tile with a few.
  // Called by the whole workgroup after a barrier.
  auto sweep = [&]() {
    for (uint32_t j = (KEEP ? swept : 0u) + tid; j < buffered; j += FX_PREP_T) {
      const float x = s_keep[3 * j], y = s_keep[3 * j + 1], z = s_keep[3 * j + 2];
      const float rx = ((M.R[0] * x + M.R[1] * y) + M.R[2] * z) + 0.0f;
      const float ry = ((M.R[3] * x + M.R[4] * y) + M.R[5] * z) + 0.0f;
      const float rz = ((M.R[6] * x + M.R[7] * y) + M.R[8] * z) + 0.0f;
      float el;
      if (!elevation_fast(x, y, z, s_atan, el)) el = elevation_deg(x, y, z);
      out[base + j] = make_float4(rx, ry, rz, el);
      if (KEEP) L.el[j] = el;
      // ring counts for the ring split (it then reads the filtered cloud once, not twice)
      int r_first;
      const uint32_t mask = isfinite(el) ? ring_membership(el, s_win, P.n_rings, el0, inv_step, r_first) : 0u;
#pragma unroll
      for (int d = 0; d < 3; ++d)
        if (mask & (1u << d)) atomicAdd(&s_ring[r_first + d], 1u);
    }
    if (!KEEP) {  // the buffer is emptied by every sweep
      base += buffered;
      buffered = 0;
    }
    swept = buffered;
    FX_STAMP(28);
    FX_SS(5);
  };
  // (Measured and not taken, profiles/r06_experiments.md §1: the twelve limits and the rotation made SCALAR by hand — the device
  //  has no scalar float arithmetic, so what the compiler computes or loads per lane stays in vector registers, 21 of them:
  //  k_prep 84 -> 72 registers, k_front 127 -> 109, and the headline 1 % lower either way.)
  const PrepBox near_box(P, near_margin), box(P);
  // (Measured, profiles/r05_experiments.md 12: a second tile of loads in flight, tile buffers that take turns without the register
  //  copies below, 16-byte loads and touching the tile after the next a period early all left the kernel's time where it was —
  //  the pass waits for its own instruction stream between the loads, not for the loads.  Round 4's "issuing the loads costs
  //  half a tile" was an artefact of the stamped build: its spilled offsets are reloaded behind s_waitcnt vmcnt(0).)
  float4 v[FX_PREP_U], nv[FX_PREP_U];
  const uint32_t t_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_begin);
  load_tile(t_first, v);
  FX_STAMP(24);
  FX_SS(0);
  for (uint32_t t0 = t_first; t0 < n_s; t0 += kTile) {
#ifdef FX_TILE_SPLIT  // (experiment: half of the next tile's loads now, half behind the barrier)
    load_tile(t0 + kTile, nv, 0, FX_PREP_U / 2);
#else
    load_tile(t0 + kTile, nv);
#endif
    FX_STAMP(30);
    FX_SS(1);
    bool keep[FX_PREP_U];
    unsigned long long mask[FX_PREP_U];
    uint32_t wave_cnt = 0;
    uint32_t nq[FX_PREP_U];  // 1: some point of this lane's group of four (one 64-byte sector) is near the box
    // pcl::transformPointCloud, dense branch: ((m0 x + m1 y) + m2 z) + t, t = 0 — two points an instruction (packed fp32: the
    // kernel is bound by instruction issue; the + 0 of the translation only turns -0 into +0, which no range test sees: the
    // sweep, whose values are stored, keeps it)
    float rxs[FX_PREP_U], rys[FX_PREP_U], rzs[FX_PREP_U];
#pragma unroll
    for (int u = 0; u < FX_PREP_U; u += 2) {
      const fx_f2 X = {v[u].x, v[u + 1].x}, Y = {v[u].y, v[u + 1].y}, Z = {v[u].z, v[u + 1].z};
      fx_f2 a, b, c;
      prep_rotate<fx_f2>(M.R, X, Y, Z, a, b, c);
      rxs[u] = a.x, rxs[u + 1] = a.y, rys[u] = b.x, rys[u + 1] = b.y, rzs[u] = c.x, rzs[u + 1] = c.y;
    }
#pragma unroll
    for (int u = 0; u < FX_PREP_U; ++u) {
      const float rx = rxs[u], ry = rys[u], rz = rzs[u];
      const bool near = near_box.has(rx, ry, rz);
      const bool k = box.has(rx, ry, rz);
      keep[u] = k;
      mask[u] = __ballot(k);
      wave_cnt += (uint32_t)__popcll(mask[u]);
      // sector s of this load holds lanes 4 s .. 4 s + 3: the OR over the four, in all of them (two quad swizzles)
      uint32_t q = near ? 1u : 0u;
      q |= (uint32_t)__builtin_amdgcn_mov_dpp((int)q, 0xB1, 0xf, 0xf, true);  // quad_perm [1, 0, 3, 2]
      q |= (uint32_t)__builtin_amdgcn_mov_dpp((int)q, 0x4E, 0xf, 0xf, true);  // quad_perm [2, 3, 0, 1]
      nq[u] = q;
    }
    // The wavefront's 64 sector bits (load u's sixteen at 16 u ..: pure sector order) in one ballot: lane 4 k + g of sector k
    // offers load g's flag, lane 16 g + k fetches it.  (Squeezing every fourth bit out of four ballots cost a hundred scalar
    // instructions a tile — a quarter of the loop, which is bound by instruction issue.)
    static_assert(FX_PREP_U == 4, "one ballot holds the sector bits of four loads");
    const uint32_t offer = (lane & 2u) ? ((lane & 1u) ? nq[3] : nq[2]) : ((lane & 1u) ? nq[1] : nq[0]);
    const unsigned long long sect =
        __ballot(__builtin_amdgcn_ds_bpermute((int)((((lane & 15u) << 2) | (lane >> 4)) << 2), (int)offer) != 0);
    if (lane == 0) {
      s_cnt[parity * NW + wave] = wave_cnt;
      // (bit s of the scan is bit s % 32 of word s / 32)
      near_bits[(t0 / kTile) * (kTile / 128) + wave * 2] = (uint32_t)sect;
      near_bits[(t0 / kTile) * (kTile / 128) + wave * 2 + 1] = (uint32_t)(sect >> 32);
    }
    // One barrier per tile: it orders this tile's counts before their readers, the previous sweep's
    // reads of s_keep before this tile's writes, and (a wave cannot be two tiles ahead of another) the
    // readers of the other parity's counts before they are overwritten next tile.
    FX_STAMP(25);
    FX_SS(2);
    __syncthreads();
#ifdef FX_TILE_SPLIT
    load_tile(t0 + kTile, nv, FX_PREP_U / 2, FX_PREP_U);
#endif
    FX_STAMP(26);
    FX_SS(3);
    // buffer slot = survivors already buffered + those of earlier waves + of earlier slices of my wave
    //               + of earlier lanes of my slice: input order is kept
    uint32_t before = 0, tile_total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const uint32_t c = s_cnt[parity * NW + w];
      before += (w < (int)wave) ? c : 0u;
      tile_total += c;
    }
    if (KEEP && buffered + tile_total > L.keep_cap) {  // (workgroup-uniform) the survivors no longer fit
      __syncthreads();
      return FX_NONE;
    }
    uint32_t pos = buffered + before;
#pragma unroll
    for (int u = 0; u < FX_PREP_U; ++u) {
      if (keep[u]) {
        const uint32_t d = 3u * (pos + lanes_below(mask[u]));
        s_keep[d] = v[u].x;
        s_keep[d + 1] = v[u].y;
        s_keep[d + 2] = v[u].z;
      }
      pos += (uint32_t)__popcll(mask[u]);
    }
    buffered += tile_total;
    parity ^= 1u;
    FX_STAMP(27);
    FX_SS(4);
    // recycling: the next tile might not fit; KEEP: enough survivors wait for the sweep to keep every lane busy (workgroup-uniform)
    if (KEEP ? buffered - swept >= 2u * FX_PREP_T : buffered > kKeep - kTile) {
      __syncthreads();
      sweep();
    }
#pragma unroll
    for (int u = 0; u < FX_PREP_U; ++u) v[u] = nv[u];
    FX_STAMP(29);
    FX_SS(6);
  }
  __syncthreads();
  sweep();
  __syncthreads();
  FX_SS(5);
  FX_SS_FLUSH(B.stamps);
  return base + buffered;
}

__global__ __launch_bounds__(FX_PREP_T, FX_PREP_OCC) void k_prep(FxDevParams P, FxBuffers B, float near_margin, float el0, float inv_step, uint32_t clk_slot) {
  constexpr int NW = FX_PREP_T / 64;
  const uint32_t scan = blockIdx.x;
  const FxScanMeta M = B.meta[scan];
  __shared__ uint32_t s_cnt[2 * NW];
  __shared__ float s_keep[3 * FX_PREP_KEEP];
  __shared__ uint32_t s_ring[FX_MAX_RINGS];
  __shared__ double s_atan[(FX_ATAN_N + 1) * (FX_ATAN_DEG + 1)];
  __shared__ float2 s_win[FX_MAX_RINGS];
  const uint32_t tid = threadIdx.x;
  // execution span of this launch on the device's constant-rate clock (first workgroup's start, last workgroup's end):
  // what rocprofv3 reports as the kernel's duration; HIP events around the launch also count its wait for free CUs
  if (tid == 0) atomicMin(&B.clk[2 * clk_slot], (unsigned long long)wall_clock64());
  const uint32_t R = (uint32_t)P.n_rings;
  if (M.n == 0) {  // empty scan (ref: node.cpp:209-210, 263-264): its pointer may be null — nothing is loaded
    if (tid == 0) {
      B.n_filt[scan] = 0u;
      B.flags[scan] = 0u;
    }
    for (uint32_t r = tid; r < R; r += FX_PREP_T) B.ring_cnt[(size_t)scan * R + r] = 0u;
    if (scan == 0 && tid < FX_N_COUNTERS) B.counters[tid] = 0u;
    return;
  }
  const PrepLds L{s_cnt, s_keep, s_ring, s_atan, s_win, nullptr, 0u};
  const uint32_t base = prep_stream<false>(P, B, M, scan, near_margin, el0, inv_step, L);
  for (uint32_t r = tid; r < R; r += FX_PREP_T) B.ring_cnt[(size_t)scan * R + r] = s_ring[r];
  if (tid == 0) {
    B.n_filt[scan] = base;
    B.flags[scan] = 0u;
  }
  if (scan == 0 && tid < FX_N_COUNTERS) B.counters[tid] = 0u;  // work-list counters of the batch (used from k_rings_runs on)
  if (tid == 0) atomicMax(&B.clk[2 * clk_slot + 1], (unsigned long long)wall_clock64());
}

// ---- several workgroups a scan (batches that leave most of the chip idle with one: 64 scans of 262 144 points are 64
// workgroups on 256 CUs, each streaming at what ONE CU's memory pipe gives).  Slice s of S takes the tiles
// [s per, (s + 1) per) of the scan.  Its survivors must land behind those of the slices before it: a counting pass first
// (k_prep_count: rotate, test, count — nothing stored), then the streaming pass proper with every slice's base known
// (k_prep_sliced).  The scan is read twice, by eight times as many workgroups; nothing waits on another workgroup.
__device__ __forceinline__ uint32_t prep_slice_tiles(uint32_t n, uint32_t S) {
  const uint32_t tiles = (n + FX_PREP_TILE - 1u) / FX_PREP_TILE;
  return (tiles + S - 1u) / S;
}
extern "C" __global__ __launch_bounds__(FX_PREP_T) void k_prep_count(FxDevParams P, FxBuffers B) {
  const uint32_t slice = blockIdx.x, S = gridDim.x, scan = blockIdx.y, tid = threadIdx.x;
  const FxScanMeta M = B.meta[scan];
  __shared__ uint32_t s_tot[FX_PREP_T / 64];
  uint32_t cnt = 0;
  if (M.n) {
    const uint32_t per = prep_slice_tiles(M.n, S) * FX_PREP_TILE;
    const uint32_t lo = slice * per, hi = min(M.n, lo + per);
    const gfloat *gpts = (const gfloat *)M.pts;
    const PrepBox box(P);
    for (uint32_t i0 = lo; i0 < hi; i0 += 4u * FX_PREP_T) {  // (four loads in flight a lane)
      float x[4], y[4], z[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t i = i0 + u * FX_PREP_T + tid;
        typedef float gf3 __attribute__((ext_vector_type(3), aligned(4)));
        const gf3 w = *(const __attribute__((address_space(1))) gf3 *)(gpts + (size_t)min(i, hi - 1u) * M.stride_f);  // (one 12-byte load: see prep_stream)
        x[u] = i < hi ? w.x : NAN, y[u] = w.y, z[u] = w.z;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {  // prep_stream's predicate: the SAME functions (prep_rotate, PrepBox::has)
        float rx, ry, rz;
        prep_rotate<float>(M.R, x[u], y[u], z[u], rx, ry, rz);
        cnt += box.has(rx, ry, rz) ? 1u : 0u;
      }
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, d, 64);
  if ((tid & 63u) == 0) s_tot[tid >> 6] = cnt;
  __syncthreads();
  if (tid == 0) {
    uint32_t t = 0;
    for (int w = 0; w < FX_PREP_T / 64; ++w) t += s_tot[w];
    B.prep_cnt[(size_t)scan * S + slice] = t;
    if (slice == 0) B.flags[scan] = 0u;  // (the batch's first launch: k_prep_sliced's slices only OR into it)
  }
}
extern "C" __global__ __launch_bounds__(FX_PREP_T, FX_PREP_OCC) void k_prep_sliced(FxDevParams P, FxBuffers B, float near_margin, float el0, float inv_step,
                                                                                 uint32_t clk_slot) {
  constexpr int NW = FX_PREP_T / 64;
  const uint32_t slice = blockIdx.x, S = gridDim.x, scan = blockIdx.y, tid = threadIdx.x;
  const FxScanMeta M = B.meta[scan];
  __shared__ uint32_t s_cnt[2 * NW];
  __shared__ float s_keep[3 * FX_PREP_KEEP];
  __shared__ uint32_t s_ring[FX_MAX_RINGS];
  __shared__ double s_atan[(FX_ATAN_N + 1) * (FX_ATAN_DEG + 1)];
  __shared__ float2 s_win[FX_MAX_RINGS];
  if (tid == 0) atomicMin(&B.clk[2 * clk_slot], (unsigned long long)wall_clock64());
  const uint32_t R = (uint32_t)P.n_rings;
  uint32_t *prc = B.prep_ring_cnt + ((size_t)scan * S + slice) * R;
  if (scan == 0 && slice == 0 && tid < FX_N_COUNTERS) B.counters[tid] = 0u;  // work-list counters of the batch (used from k_rings_runs on)
  if (M.n == 0) {  // empty scan: its pointer may be null — nothing is loaded
    if (tid == 0 && slice == 0) {
      B.n_filt[scan] = 0u;
      B.flags[scan] = 0u;
    }
    for (uint32_t r = tid; r < R; r += FX_PREP_T) prc[r] = 0u;
    return;
  }
  uint32_t base0 = 0;
  for (uint32_t j = 0; j < slice; ++j) base0 += B.prep_cnt[(size_t)scan * S + j];  // (workgroup-uniform)
  const uint32_t per = prep_slice_tiles(M.n, S) * FX_PREP_TILE;
  const PrepLds L{s_cnt, s_keep, s_ring, s_atan, s_win, nullptr, 0u};
  const uint32_t end = prep_stream<false>(P, B, M, scan, near_margin, el0, inv_step, L, slice * per, slice * per + per, base0);
  for (uint32_t r = tid; r < R; r += FX_PREP_T) prc[r] = s_ring[r];
  if (tid == 0) {
    // (the scan's flag word was cleared by k_prep_count, the launch before: a slice that fails the self-check below ORs into it)
    if (slice == S - 1u) B.n_filt[scan] = end;
    // self-check: this pass's survivors of the slice against the counting pass's (the other slices' bases were built from
    // those counts: a disagreement means overlapping or missing stretches of ~cloud)
    if (end - base0 != B.prep_cnt[(size_t)scan * S + slice]) atomicOr(&B.flags[scan], FX_FLAG_INTERNAL);
  }
  if (tid == 0) atomicMax(&B.clk[2 * clk_slot + 1], (unsigned long long)wall_clock64());
}

// ====================================================================== stage 2a: ring buckets
// estimateKeypoints' ring loop (ref: node.cpp:195-202) runs 16 PassThrough filters over the
// filtered cloud.  Here one workgroup per scan deals the filtered points to their rings in
// one stable pass (a point on a window boundary belongs to both rings, A.3), so the ring
// workgroups read exactly their own points, already in ring order.
#ifndef FX_BUCKET_T
#define FX_BUCKET_T 512
#endif
#define FX_BUCKET_NW (FX_BUCKET_T / 64)
// MANY: sensors of more than 24 rings (the sort-based ranking below is compiled only into that instance)
// (prc: the scan is split over S workgroups — k_bucket_sliced —, this one takes the survivors [f_begin, f_end) that workgroup
//  `slice` of k_prep_sliced wrote, whose per-ring counts are prc[(scan S + j) R + r]: a ring's total is their sum, this
//  slice's first place in the ring the sum over the slices before it)
template <bool MANY, int NT>
__device__ __forceinline__ void bucket_body(const FxDevParams &P, const FxBuffers &B, uint32_t scan, float el0, float inv_step, uint32_t *smem,
                                            const uint32_t *prc = nullptr, uint32_t S = 1u, uint32_t slice = 0u, uint32_t f_begin = 0u,
                                            uint32_t f_end = 0xffffffffu) {
  constexpr int NWV = NT / 64;
  const uint32_t R = (uint32_t)P.n_rings;
  uint32_t *s_w = smem;            // 48: block helpers, per-wave ring ranges
  uint32_t *cnt = smem + 48;       // [R] total per ring, then running fill
  uint32_t *off = cnt + R;         // [R + 1]
  uint32_t *cw = off + R + 1;      // [NT / 64][R] per-wave counts of the current chunk
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t nf = min(B.n_filt[scan], f_end);
  const float4 *f = B.filt + (size_t)scan * P.max_points;
  for (uint32_t r = tid; r < R; r += NT) {
    uint32_t c = 0, fill = 0;  // the ring's points in all slices / in the slices before this one
    if (prc) {
      for (uint32_t j = 0; j < S; ++j) {
        const uint32_t v = prc[((size_t)scan * S + j) * R + r];
        c += v;
        fill += j < slice ? v : 0u;
      }
    } else {
      c = B.ring_cnt[(size_t)scan * R + r];  // counted by k_prep's sweep
    }
    cnt[r] = c;
    cw[r] = fill;  // (cw is free until the chunk loop: parked here until cnt becomes the running fill)
  }
  __syncthreads();
  uint32_t total = 0;
  for (uint32_t b0 = 0; b0 < R; b0 += NT) {
    const uint32_t r = b0 + tid;
    const uint32_t c = r < R ? cnt[r] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan<NT>(c, s_w, tot);
    if (r < R) off[r] = total + ex;
    total += tot;
  }
  __syncthreads();
  const bool overflow = total > P.ring_slot_cap;
  uint32_t *g_off = B.ring_off + (size_t)scan * R, *g_cnt = B.ring_cnt + (size_t)scan * R;
  for (uint32_t r = tid; r < R; r += NT) {
    if (slice == 0) {
      g_off[r] = overflow ? 0u : off[r];
      g_cnt[r] = overflow ? 0u : cnt[r];
    }
    cnt[r] = cw[r];  // becomes the running fill (from the slices before this one)
  }
  if (overflow) {
    if (tid == 0 && slice == 0) atomicOr(&B.flags[scan], FX_FLAG_RING_OVERFLOW);
    return;
  }
  __syncthreads();
  float4 *dst = B.ring_pts + (size_t)scan * P.ring_slot_cap;
  for (uint32_t b0 = f_begin; b0 < nf; b0 += NT) {
    const uint32_t i = b0 + tid;
    float4 v = make_float4(0, 0, 0, 0);
    int r_first = 0;
    uint32_t mask = 0;
    if (i < nf) {
      v = f[i];
      if (isfinite(v.w)) mask = ring_membership(v.w, B.ring_win, P.n_rings, el0, inv_step, r_first);
    }
    // rings present in this chunk: [lo, hi)
    int lo = mask ? r_first : 0x7fffffff, hi = mask ? r_first + 3 : -1;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      lo = min(lo, __shfl_xor(lo, d, 64));
      hi = max(hi, __shfl_xor(hi, d, 64));
    }
    if (lane == 0) {
      s_w[wave] = (uint32_t)lo;
      s_w[NWV + wave] = (uint32_t)hi;
    }
    __syncthreads();
    lo = 0x7fffffff, hi = -1;
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
      lo = min(lo, (int)s_w[w]);
      hi = max(hi, (int)s_w[NWV + w]);
    }
    lo = max(lo, 0);
    hi = min(hi, (int)R);
    uint32_t my_rank[3] = {0, 0, 0};
    // Rank of every point among the points of its ring in this wavefront's 64, and the wavefront's count per ring.
    // Few rings in the chunk: one ballot per ring.  Many (a 64- or 128-ring sensor in firing order: every ring, every chunk):
    // the lanes sort (ring, lane) keys — a bitonic network over the wavefront, 21 exchanges whatever the number of rings —,
    // a ring's points are then neighbours in lane order, and a point's rank is its distance from the ring's first.
    // (Only for wavefronts without a point on a window boundary — such a point is in two rings; those take the ballots.)
    const bool by_sort = MANY && hi - lo > 24 && !__ballot(mask & (mask - 1u));
    if (by_sort) {
      for (int r = lo + (int)lane; r < hi; r += 64) cw[wave * R + r] = 0u;
      const uint32_t d0 = mask ? (uint32_t)__ffs((int)mask) - 1u : 0u;
      // (lanes without a point sort behind every ring and keep their lane number: the way back below is a permutation)
      uint32_t key = ((mask ? (uint32_t)(r_first + (int)d0) : 0x3ffffffu) << 6) | lane;
#pragma unroll
      for (uint32_t kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
          const uint32_t other = (uint32_t)__shfl_xor((int)key, (int)j, 64);
          const bool up = (lane & kk) == 0u, low = (lane & j) == 0u;  // ascending block / lower lane of the pair
          key = (up == low) ? min(key, other) : max(key, other);
        }
      }
      const uint32_t ring_s = key >> 6, prev = (uint32_t)__shfl_up((int)key, 1, 64), next = (uint32_t)__shfl_down((int)key, 1, 64);
      const bool valid = ring_s != 0x3ffffffu;
      const bool first = valid && (lane == 0 || (prev >> 6) != ring_s);
      const unsigned long long starts = __ballot(first);
      const unsigned long long below = starts & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
      const uint32_t rank = valid ? lane - (63u - (uint32_t)__clzll((long long)below)) : 0u;
      if (valid && (lane == 63 || (next >> 6) != ring_s)) cw[wave * R + ring_s] = rank + 1u;  // (the ring's last point here)
      // back to the lane the point came from
      const uint32_t mine = (uint32_t)__builtin_amdgcn_ds_permute((int)((key & 63u) << 2), (int)rank);
      if (mask) my_rank[d0] = mine;
    } else {
      for (int r = lo; r < hi; ++r) {
        const int d = r - r_first;
        const bool in = mask && d >= 0 && d < 3 && (mask & (1u << d));
        const unsigned long long m = __ballot(in);
        if (in) my_rank[d] = lanes_below(m);
        if (lane == 0) cw[wave * R + r] = (uint32_t)__popcll(m);
      }
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (mask & (1u << d)) {
        const uint32_t r = (uint32_t)(r_first + d);
        uint32_t before = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) before += (w < (int)wave) ? cw[w * R + r] : 0u;
        dst[off[r] + cnt[r] + before + my_rank[d]] = v;
      }
    }
    __syncthreads();
    for (int r = lo + (int)tid; r < hi; r += NT) {
      uint32_t c = 0;
#pragma unroll
      for (int w = 0; w < NWV; ++w) c += cw[w * R + r];
      cnt[r] += c;
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void clk_reset(const FxBuffers &B, uint32_t clk_next) {  // the clock slot the next batch's k_prep stamps
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    B.clk[2 * clk_next] = ~0ull;
    B.clk[2 * clk_next + 1] = 0ull;
  }
}
extern "C" __global__ __launch_bounds__(FX_BUCKET_T) void k_bucket(FxDevParams P, FxBuffers B, float el0, float inv_step, uint32_t clk_next) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  clk_reset(B, clk_next);
  bucket_body<false, FX_BUCKET_T>(P, B, blockIdx.x, el0, inv_step, smem);
}
extern "C" __global__ __launch_bounds__(FX_BUCKET_T) void k_bucket_many(FxDevParams P, FxBuffers B, float el0, float inv_step, uint32_t clk_next) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  clk_reset(B, clk_next);
  bucket_body<true, FX_BUCKET_T>(P, B, blockIdx.x, el0, inv_step, smem);
}

// the ring split of a scan by S workgroups: workgroup `slice` deals the survivors slice `slice` of k_prep_sliced wrote
extern "C" __global__ __launch_bounds__(FX_BUCKET_T) void k_bucket_sliced(FxDevParams P, FxBuffers B, float el0, float inv_step, uint32_t clk_next) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t slice = blockIdx.x, S = gridDim.x, scan = blockIdx.y;
  if (scan == 0 && slice == 0 && threadIdx.x == 0) {
    B.clk[2 * clk_next] = ~0ull;
    B.clk[2 * clk_next + 1] = 0ull;
  }
  uint32_t f_begin = 0;
  for (uint32_t j = 0; j < slice; ++j) f_begin += B.prep_cnt[(size_t)scan * S + j];
  const uint32_t f_end = f_begin + B.prep_cnt[(size_t)scan * S + slice];
  if (P.n_rings > 24)
    bucket_body<true, FX_BUCKET_T>(P, B, scan, el0, inv_step, smem, B.prep_ring_cnt, S, slice, f_begin, f_end);
  else
    bucket_body<false, FX_BUCKET_T>(P, B, scan, el0, inv_step, smem, B.prep_ring_cnt, S, slice, f_begin, f_end);
}

// ====================================================================== stage 2b: rings
// LDS of one ring: 7 words per point (float4 point, parent, size|position, run / member rank)
// + 8 words per cluster (root, sort record, scratch / member offset, slot, float4 centroid).
struct RingLds {
  float4 *pt, *cc;
  uint32_t *parent, *csize, *rank, *croot, *crec, *ckoff, *cslot, *s_w;
};
#define FX_RING_WORDS_PER_POINT 7
#define FX_RING_WORDS_PER_CLUSTER 8
// (gs: the per-point and per-cluster arrays go to that scratch region of HBM instead — the slow tier, k_slow; the scratch
//  words in front stay in LDS)
template <int NT>
__device__ __forceinline__ RingLds ring_carve(uint32_t *smem, uint32_t cap, uint32_t ccap, uint32_t *gs = nullptr) {
  RingLds L;
  L.s_w = smem;
  uint32_t *p = gs ? gs : smem + SegCfg<NT>::kWords;  // multiple of 4 words: the float4 arrays stay 16-byte aligned
  L.pt = reinterpret_cast<float4 *>(p), p += 4 * cap;
  L.cc = reinterpret_cast<float4 *>(p), p += 4 * ccap;
  L.parent = p, p += cap;
  L.csize = p, p += cap;
  L.rank = p, p += cap;
  L.croot = p, p += ccap;
  L.crec = p, p += ccap;
  L.ckoff = p, p += ccap;
  L.cslot = p, p += ccap;
  return L;
}

// One (scan, ring): getCylinderSegments (ref: node.cpp:261-327) on the ring's points:
// Euclidean clustering, centroid + diameter gate, candidates in PCL's cluster order, member
// points for keypoint_cloud.  Returns false when the ring does not fit this tier (more than
// `cap` points or more than `ccap` size-admissible clusters): the caller defers it to a larger one.
template <int NT, bool GS = false>
__device__ __forceinline__ bool ring_body(const FxDevParams &P, const FxBuffers &B, uint32_t scan, uint32_t ring, uint32_t cap,
                          uint32_t ccap, uint32_t *smem, bool last_tier, uint32_t *gs = nullptr) {
  RingLds L = ring_carve<NT>(smem, cap, ccap, GS ? gs : nullptr);
  unsigned long long *const stamp_base = B.stamps ? B.stamps + (NT == 64 ? 0 : 16) : nullptr;
  FX_STAMP_INIT(stamp_base);
  const uint32_t tid = threadIdx.x;
  const size_t ring_slot = (size_t)scan * P.n_rings + ring;
  const uint32_t n = B.ring_cnt[ring_slot], off = B.ring_off[ring_slot];
  if (n > cap) {
    if (!last_tier) return false;
    if (tid == 0) {
      atomicOr(&B.flags[scan], FX_FLAG_RING_OVERFLOW);
      B.ring_cand_cnt[ring_slot] = 0;
      B.kpc_ring_cnt[ring_slot] = 0;
    }
    return true;
  }
  if (n == 0) {  // ref: node.cpp:263-264
    if (tid == 0) {
      B.ring_cand_cnt[ring_slot] = 0;
      B.kpc_ring_cnt[ring_slot] = 0;
    }
    return true;
  }
  const float4 *src = B.ring_pts + (size_t)scan * P.ring_slot_cap + off;
  for (uint32_t i = tid; i < n; i += NT) L.pt[i] = src[i];
  wg_sync<GS>();
  FX_STAMP(1);

  // ---- pcl::EuclideanClusterExtraction (ref: node.cpp:269-276)
  const uint32_t n_segs = cc_label<NT, GS>(L.pt, n, P.r2_cluster, L.parent, L.csize, L.rank, L.s_w, stamp_base);
  const uint32_t n_c = cc_order<NT, GS>(n, L.parent, L.csize, P.min_count, P.max_count, L.croot, L.crec, L.ckoff, ccap, L.s_w,
                                    stamp_base);
  if (n_c > ccap) return false;  // (ccap == cap in the last tier, so this cannot happen there)
#ifdef FX_STAMPS
  stamp_prev_ = __builtin_amdgcn_s_memtime();
#endif
  // ---- cluster position in PCL's order, packed next to the size: csize[root] = size | (s + 1) << 16;
  //      xy bounding box per cluster (ref: node.cpp:289-305) starts at the reference's +-1000
  uint32_t *bb = reinterpret_cast<uint32_t *>(L.cc);  // [s][minx, maxx, miny, maxy] until the centroids go there
  for (uint32_t s = tid; s < n_c; s += NT) {
    const uint32_t root = L.croot[L.crec[s] & 0xffffu];
    L.csize[root] |= (s + 1u) << 16;
    bb[4 * s + 0] = f2ord(1000.0f);
    bb[4 * s + 1] = f2ord(-1000.0f);
    bb[4 * s + 2] = f2ord(1000.0f);
    bb[4 * s + 3] = f2ord(-1000.0f);
  }
  wg_sync<GS>();
  // min/max are exact whatever the order, so segments (or points) fold into their cluster's box
  if (n_segs <= SegCfg<NT>::kMax) {
    const SegTable<NT> ST(L.s_w);
    for (uint32_t sg = tid; sg < n_segs; sg += NT) {
      const uint32_t pos = L.csize[L.parent[ST.start(sg)]] >> 16;
      if (pos == 0) continue;  // not a size-admissible cluster
      uint32_t *b = bb + 4 * (pos - 1u);
      atomicMin(&b[0], f2ord(ST.box(FX_MINX, sg)));
      atomicMax(&b[1], f2ord(ST.box(FX_MAXX, sg)));
      atomicMin(&b[2], f2ord(ST.box(FX_MINY, sg)));
      atomicMax(&b[3], f2ord(ST.box(FX_MAXY, sg)));
    }
  } else {
    for (uint32_t i = tid; i < n; i += NT) {
      const uint32_t pos = L.csize[L.parent[i]] >> 16;
      if (pos == 0) continue;
      uint32_t *b = bb + 4 * (pos - 1u);
      const float4 q = L.pt[i];
      atomicMin(&b[0], f2ord(q.x));
      atomicMax(&b[1], f2ord(q.x));
      atomicMin(&b[2], f2ord(q.y));
      atomicMax(&b[3], f2ord(q.y));
    }
  }
  wg_sync<GS>();
  FX_STAMP(7);
  // ---- diameter gate per cluster, in PCL's cluster order (ref: node.cpp:314-316)
  for (uint32_t s = tid; s < n_c; s += NT) {
    const double minx = ord2f(bb[4 * s + 0]), maxx = ord2f(bb[4 * s + 1]);
    const double miny = ord2f(bb[4 * s + 2]), maxy = ord2f(bb[4 * s + 3]);
    const double ddx = maxx - minx, ddy = maxy - miny;
    const double diameter = sqrt(ddx * ddx + ddy * ddy);
    L.cslot[s] = (diameter < P.gate_diameter) ? 1u : 0u;
  }
  wg_sync<GS>();
  FX_STAMP(8);
  // ---- centroid of the clusters that pass: fp64 sums in ascending member order
  //      (ref: node.cpp:293-297, 317-320); the walk also ranks the members for keypoint_cloud
  for (uint32_t s = tid; s < n_c; s += NT) {
    if (L.cslot[s] == 0u) continue;
    const uint32_t rec = L.crec[s];
    const uint32_t sz = rec >> 16, root = L.croot[rec & 0xffffu];
    double sumx = 0.0, sumy = 0.0, sumz = 0.0;
    uint32_t cnt = 0;
    // (four points per LDS round trip: label and coordinates loaded before either is used; the
    //  members of a ring cluster are almost always consecutive)
    for (uint32_t i = root; i < n && cnt < sz; i += 4) {
      uint32_t pr[4];
      float4 q[4];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t iu = min(i + u, n - 1u);
        pr[u] = L.parent[iu];
        q[u] = L.pt[iu];
      }
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        if (i + u >= n || pr[u] != root) continue;
        sumx += (double)q[u].x;
        sumy += (double)q[u].y;
        sumz += (double)q[u].z;
        L.rank[i + u] = cnt++;
      }
    }
    L.cc[s] = make_float4((float)(sumx / (double)sz), (float)(sumy / (double)sz), (float)(sumz / (double)sz), L.pt[root].w);
  }
  wg_sync<GS>();
  FX_STAMP(9);

  // ---- slots of the gate-passing clusters and offsets of their member runs
  uint32_t n_pass = 0, n_mem = 0;
  for (uint32_t b0 = 0; b0 < n_c; b0 += NT) {
    const uint32_t s = b0 + tid;
    const bool pass = s < n_c && L.cslot[s] != 0u;
    const uint32_t sz = pass ? (L.crec[s] >> 16) : 0u;
    uint32_t tot_p, tot_m;
    const uint32_t slot = block_rank<NT>(pass, L.s_w, tot_p);
    const uint32_t koff = block_excl_scan<NT>(sz, L.s_w, tot_m);
    if (s < n_c) {
      L.cslot[s] = pass ? (n_pass + slot) : FX_NONE;
      L.ckoff[s] = n_mem + koff;
    }
    n_pass += tot_p;
    n_mem += tot_m;
  }
  wg_sync<GS>();
  FX_STAMP(10);

  // ---- candidates of this ring (cylinderCentroids, ref: node.cpp:322)
  float4 *rc = B.ring_cand + ring_slot * P.max_ring_cands;
  uint32_t *rcs = B.ring_cand_size + ring_slot * P.max_ring_cands;
  for (uint32_t s = tid; s < n_c; s += NT) {
    const uint32_t slot = L.cslot[s];
    if (slot < P.max_ring_cands) {
      rc[slot] = L.cc[s];
      rcs[slot] = L.crec[s] >> 16;
    }
  }
  // ---- member points (cylinderCloud, ref: node.cpp:310, 323): the ring's chunk of the pool
  //      starts where the ring's points start (members are a subset of them); k_merge lays
  //      the chunks out back to back in ring order.
  if (tid == 0) {
    if (n_pass > P.max_ring_cands) atomicOr(&B.flags[scan], FX_FLAG_CAND_OVERFLOW);
    B.ring_cand_cnt[ring_slot] = n_pass < P.max_ring_cands ? n_pass : P.max_ring_cands;
    B.kpc_ring_cnt[ring_slot] = n_mem;
  }
  if (n_mem) {
    float4 *pool = B.kpc_pool + (size_t)scan * P.ring_slot_cap + off;
    uint32_t *pool_c = B.kpc_pool_cand + (size_t)scan * P.ring_slot_cap + off;
    for (uint32_t i = tid; i < n; i += NT) {
      const uint32_t pos = L.csize[L.parent[i]] >> 16;
      if (pos == 0) continue;
      const uint32_t s = pos - 1u;
      const uint32_t slot = L.cslot[s];
      if (slot == FX_NONE) continue;
      const uint32_t dst = L.ckoff[s] + L.rank[i];
      pool[dst] = L.pt[i];
      pool_c[dst] = slot;
    }
  }
  wg_sync<GS>();
  FX_STAMP(11);
  return true;
}

// ---------------------------------------------------------------- run tier
// getCylinderSegments on one wavefront with nothing per point in LDS.  A ring arrives azimuth ordered, so its
// clusters are unions of RUNS (maximal chains of consecutive points closer than the tolerance): the tier keeps a table
// of runs and of their segments (start, xy box) and does everything else on it —
//   * run labelling streams the points from L2 (they were just written ring-major by k_bucket), each chunk of 64
//     beside its predecessor; the union-find runs over run indices (root = smallest run = the run holding the
//     component's smallest point index: PCL's indices[0] and discovery order);
//   * cross-run edges as in cc_label: near run pairs from the run boxes, the points of the earlier run against the
//     later run's box, then its segments' boxes, then its points (read from L2) — every pair that could be an edge
//     is examined;
//   * sizes are sums of run lengths, cluster boxes folds of segment boxes; cc_order (PCL's order) runs on the run
//     table unchanged; one lane per gate-passing cluster walks its runs' points in ascending order for the fp64
//     centroid; members are copied out segment by segment.
// LDS per ring is 10 KB whatever the ring's size (the point-holding wave tier needed 13 KB for 256 points, the
// workgroup tier 32 KB for 640), so 15 rings per CU are in flight instead of 11 + 5, and rings of several hundred points
// — ground and wall arcs: a handful of long runs — no longer need a workgroup.  Rings with more than 128 runs or
// segments, more than 64 admissible clusters or too many near run pairs (unordered input) go to the workgroup tiers.
// Two instances: 128 segments / 128 runs and clusters (a cluster is at least a run) for every ring first — 3067 words, six 2 KiB
// granules —, and 384 / 256 (24 KB) behind it for the rings of dense many-ring sensors, which otherwise need a 1024-thread
// workgroup each.
#define FX_RR_QUEUE 96
#ifndef FX_RR_U
#define FX_RR_U 4  // LDS / L2 reads in flight in the run tier's pair loops (8 cost 40 registers more: 3 wavefronts a SIMD instead of 4)
#endif
#ifndef FX_RR_CACHE
#define FX_RR_CACHE 64  // the ring's first points kept in LDS (64 / 96 / 128 / 160 measured: 64 makes it 16 wavefronts a CU — 0.123 ms for 0.130)
#endif
#ifndef FX_RR_S
#define FX_RR_S 128   // first run tier: segments
#endif
#ifndef FX_RR_RN
#define FX_RR_RN 128  // first run tier: runs and clusters
#endif
template <uint32_t S, uint32_t RN>
__host__ __device__ constexpr uint32_t rr_words() {
  return 152 + 4 * RN + (S + 4) + 4 * S + (RN + 1) + 4 * RN + 3 * RN + 3 * FX_RR_CACHE;
}
template <uint32_t S, uint32_t RN>
__device__ __forceinline__ bool ring_runs_body(const FxDevParams &P, const FxBuffers &B, uint32_t scan, uint32_t ring, uint32_t max_pts,
                                               uint32_t *smem) {
  constexpr uint32_t CC = RN;
  const uint32_t lane = threadIdx.x;
  unsigned long long *const stamp_base = B.stamps ? B.stamps : nullptr;
  FX_STAMP_INIT(stamp_base);
  uint32_t *s_w = smem;                    // [152]: block helpers, broadcast slots, sort stack / near-pair list
  uint32_t *rbox = smem + 152;             // [RN][min x, max x, min y, max y] run boxes ...
  float4 *cc = reinterpret_cast<float4 *>(rbox);  // ... and, once the edges are in, [CC] cluster boxes, then centroids (16-byte aligned)
  uint32_t *seg_tab = rbox + 4 * RN;       // [S + 1 (+ 3: alignment)] first point of each segment | its run << 16; seg_tab[n_segs] = n
  uint32_t *seg_box = seg_tab + S + 4;     // [S][min x, max x, min y, max y] (ordered uints, then float bits)
  uint32_t *rseg = seg_box + 4 * S;        // [RN + 1] first segment of each run; rseg[n_runs] = n_segs
  uint32_t *rparent = rseg + RN + 1;       // [RN] union-find over runs
  uint32_t *rsize = rparent + RN;          // [RN] points of the component (at its root) | position in PCL's order << 16
  uint32_t *roff = rsize + RN;             // [RN] offset of the run's points inside its cluster
  float *rw = reinterpret_cast<float *>(roff + RN);  // [RN] elevation angle of the run's first point
  // Per cluster: croot[c] = root run of the cluster cc_order numbered c (low half) | slot of the cluster at position c of
  // PCL's order (high half: the gate's 0 / 1, then the candidate slot, 0xffff = none), crec, ctmp.  The queue of parked
  // (point, run) items lives in crec until cc_order writes it.  (Every word saved here is occupancy: the kernel waits on
  // LDS and L2 round trips, and 10.1 KB instead of 12.3 — with 119 registers instead of 156 — is 16 wavefronts a CU for 12.)
  uint32_t *croot = roff + 2 * RN, *crec = croot + CC, *ctmp = crec + CC;
  uint32_t *wq = crec;                     // [FX_RR_QUEUE] parked (point, run) items
  static_assert(FX_RR_QUEUE <= CC, "the queue borrows crec");
  auto cl_root = [&](uint32_t c) { return croot[c] & 0xffffu; };
  auto cl_slot = [&](uint32_t s) { return croot[s] >> 16; };
  auto cl_set_slot = [&](uint32_t s, uint32_t v) { croot[s] = (croot[s] & 0xffffu) | (v << 16); };
  constexpr uint32_t kNoSlot = 0xffffu;
  // the ring's first NC points (x, y, z): later phases read those from LDS, the rest from L2 (more wavefronts a CU are worth
  // more than a longer cache: FX_RR_CACHE)
  constexpr uint32_t NC = FX_RR_CACHE;
  float *px = reinterpret_cast<float *>(ctmp + CC), *py = px + NC, *pz = py + NC;
  auto sst = [&](uint32_t sg) { return seg_tab[sg] & 0xffffu; };
  auto srun = [&](uint32_t sg) { return seg_tab[sg] >> 16; };
  const size_t ring_slot = (size_t)scan * P.n_rings + ring;
  const uint32_t n = B.ring_cnt[ring_slot], off = B.ring_off[ring_slot];
  if (n == 0) {  // ref: node.cpp:263-264
    if (lane == 0) {
      B.ring_cand_cnt[ring_slot] = 0;
      B.kpc_ring_cnt[ring_slot] = 0;
    }
    return true;
  }
  if (n > 65535u || n > max_pts) return false;  // (beyond limits.max_ring_points: the last workgroup tier flags the scan)
  const float4 *src = B.ring_pts + (size_t)scan * P.ring_slot_cap + off;
  const float r2 = P.r2_cluster;
  for (uint32_t t = lane; t < S; t += 64) {
    seg_box[4 * t + 0] = f2ord(INFINITY), seg_box[4 * t + 1] = f2ord(-INFINITY);
    seg_box[4 * t + 2] = f2ord(INFINITY), seg_box[4 * t + 3] = f2ord(-INFINITY);
  }
  for (uint32_t t = lane; t < RN; t += 64) {
    rbox[4 * t + 0] = f2ord(INFINITY), rbox[4 * t + 1] = f2ord(-INFINITY);
    rbox[4 * t + 2] = f2ord(INFINITY), rbox[4 * t + 3] = f2ord(-INFINITY);
  }
  wave_sync_lds();
  FX_STAMP(1);
  // ---- run labelling: a point starts a run when it is not closer than the tolerance to its predecessor
  const uint32_t seg_len = max(8u, (n + (S * 3u / 4u) - 1u) / (S * 3u / 4u));  // (at most 3/4 S segments by length; runs start the others)
  const unsigned long long le_mask = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
  uint32_t n_runs = 0, n_segs = 0, head_carry = 0;
  float4 prev_last = make_float4(0, 0, 0, 0);
  float4 qv[4];     // the first four chunks stay in registers for the member copy at the end,
  uint32_t sgv[4];  // each point with its segment
#pragma unroll
  for (uint32_t k = 0; k < 4; ++k) qv[k] = make_float4(0, 0, 0, 0), sgv[k] = 0;
  float4 nxt = lane < n ? src[lane] : make_float4(0, 0, 0, 0);
  for (uint32_t b0 = 0; b0 < n; b0 += 64) {
    const uint32_t i = b0 + lane;
    const bool in = i < n;
    const float4 q = nxt;
    if (b0 + 64 < n) nxt = (i + 64 < n) ? src[i + 64] : make_float4(0, 0, 0, 0);  // the next chunk is in flight
    float4 p;
    p.x = __shfl_up(q.x, 1, 64), p.y = __shfl_up(q.y, 1, 64), p.z = __shfl_up(q.z, 1, 64);
    if (lane == 0) p = prev_last;
    const bool start = in && (i == 0 || !(dist2(q.x, q.y, q.z, p.x, p.y, p.z) < r2));
    const unsigned long long m = __ballot(start);
    const unsigned long long below = m & le_mask;
    const uint32_t head = below ? b0 + (63u - (uint32_t)__clzll((long long)below)) : head_carry;
    const uint32_t r = n_runs + (uint32_t)__popcll(below) - 1u;
    const bool seg_first = in && (start || ((i - head) % seg_len) == 0u);
    const unsigned long long ms = __ballot(seg_first);
    const uint32_t sg = n_segs + (uint32_t)__popcll(ms & le_mask) - 1u;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      if (b0 == 64 * k) qv[k] = q, sgv[k] = sg;
    }
    if (i < NC && in) px[i] = q.x, py[i] = q.y, pz[i] = q.z;
    if (in && sg < S && r < RN) {
      if (seg_first) {
        seg_tab[sg] = i | (r << 16);
        if (start) rseg[r] = sg, rw[r] = q.w;
      }
      const uint32_t ox = f2ord(q.x), oy = f2ord(q.y);
      atomicMin(&seg_box[4 * sg + 0], ox);
      atomicMax(&seg_box[4 * sg + 1], ox);
      atomicMin(&seg_box[4 * sg + 2], oy);
      atomicMax(&seg_box[4 * sg + 3], oy);
    }
    if (m) head_carry = b0 + (63u - (uint32_t)__clzll((long long)m));
    n_runs += (uint32_t)__popcll(m);
    n_segs += (uint32_t)__popcll(ms);
    prev_last.x = __shfl(q.x, 63, 64), prev_last.y = __shfl(q.y, 63, 64), prev_last.z = __shfl(q.z, 63, 64);
  }
  if (n_runs > RN || n_segs > S) return false;  // (wave-uniform) unordered or very fragmented ring: workgroup tiers
  wave_sync_lds();
  FX_STAMP(2);
  if (lane == 0) {
    FX_COUNT(12, 1);
    FX_COUNT(13, n_runs);
    FX_COUNT(14, n_segs);
  }
  // ---- segment boxes -> floats, folded into the run boxes; union-find over runs
  for (uint32_t sg = lane; sg < n_segs; sg += 64) {
    const uint32_t r = srun(sg);
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      const uint32_t o = seg_box[4 * sg + k];
      seg_box[4 * sg + k] = __float_as_uint(ord2f(o));
      if (k & 1)
        atomicMax(&rbox[4 * r + k], o);
      else
        atomicMin(&rbox[4 * r + k], o);
    }
  }
  if (lane == 0) {
    seg_tab[n_segs] = n;
    rseg[n_runs] = n_segs;
    s_w[16] = 0;
  }
  wave_sync_lds();
  for (uint32_t r = lane; r < n_runs; r += 64) {
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) rbox[4 * r + k] = __float_as_uint(ord2f(rbox[4 * r + k]));
    rparent[r] = r;
    rsize[r] = 0;
  }
  wave_sync_lds();
  const float4 *sbox4 = reinterpret_cast<const float4 *>(seg_box);
  auto rb = [&](uint32_t k, uint32_t r) { return __uint_as_float(rbox[4 * r + k]); };
  // points i .. i + U - 1, clamped to last: from the LDS copy when it holds them all, else U loads from L2 in flight
  auto fetch = [&](auto U_, uint32_t i, uint32_t last, float *x, float *y, float *z) {
    constexpr uint32_t U = decltype(U_)::value;
    if (last < NC) {
#pragma unroll
      for (uint32_t u = 0; u < U; ++u) {
        const uint32_t j = min(i + u, last);
        x[u] = px[j], y[u] = py[j], z[u] = pz[j];
      }
    } else {
#pragma unroll
      for (uint32_t u = 0; u < U; ++u) {
        const float4 t = src[min(i + u, last)];
        x[u] = t.x, y[u] = t.y, z[u] = t.z;
      }
    }
  };
  if (n_runs > 1) {
    const float r2_pad = r2 * 1.001f;  // box distances are lower bounds; pad them against fp32 rounding
    // near run pairs (azimuth-ordered rings have few): the list borrows the sort stack, idle until cc_order
    uint32_t *rp = s_w + 32;
    constexpr uint32_t kPairCap = FX_SORT_STACK_WORDS;
    // a run per lane, its box in registers, against every later run: one 16-byte broadcast read per trip
    const float4 *rbox4 = reinterpret_cast<const float4 *>(rbox);
    for (uint32_t a0 = 0; a0 + 1u < n_runs; a0 += 64) {
      const uint32_t a = a0 + lane;
      const float4 ba = rbox4[min(a, n_runs - 1u)];
      for (uint32_t b0 = a0 + 1u; b0 < n_runs; b0 += FX_RR_U) {  // (FX_RR_U reads in flight: the wavefront has little company on its SIMD)
        float4 bx[FX_RR_U];
#pragma unroll
        for (uint32_t u = 0; u < FX_RR_U; ++u) bx[u] = rbox4[min(b0 + u, n_runs - 1u)];
        uint32_t near = 0;
#pragma unroll
        for (uint32_t u = 0; u < FX_RR_U; ++u) {
          const float dx = fmaxf(fmaxf(bx[u].x - ba.y, ba.x - bx[u].y), 0.0f);
          const float dy = fmaxf(fmaxf(bx[u].z - ba.w, ba.z - bx[u].w), 0.0f);
          near |= (b0 + u < n_runs && b0 + u > a && !(dx * dx + dy * dy > r2_pad)) ? (1u << u) : 0u;
        }
        while (near) {  // (rare)
          const uint32_t u = (uint32_t)__ffs((int)near) - 1u;
          near &= near - 1u;
          const uint32_t slot = atomicAdd(&s_w[16], 1u);
          if (slot < kPairCap) rp[slot] = (a << 16) | (b0 + u);
        }
      }
    }
    wave_sync_lds();
    const uint32_t n_rp = s_w[16];
    if (lane == 0) {
      FX_COUNT(15, n_rp);
    }
    if (n_rp > kPairCap) return false;  // runs all over each other (unordered input): workgroup tiers
    uint32_t wq_n = 0;
    auto drain = [&]() {
      wave_sync_lds();
      for (uint32_t t = lane; t < wq_n; t += 64) {
        const uint32_t item = wq[t];
        const uint32_t i = item & 0xffffu, b = (item >> 16) & 0xffu, a = item >> 24;
        if (uf_find(rparent, a) == uf_find(rparent, b)) continue;  // already one component
        float3 q;
        fetch(std::integral_constant<uint32_t, 1>{}, i, i, &q.x, &q.y, &q.z);
        bool linked = false;
        for (uint32_t sg = rseg[b]; sg < rseg[b + 1] && !linked; ++sg) {
          const float4 sb = sbox4[sg];
          const float dx = fmaxf(fmaxf(sb.x - q.x, q.x - sb.y), 0.0f);
          const float dy = fmaxf(fmaxf(sb.z - q.y, q.y - sb.w), 0.0f);
          if (dx * dx + dy * dy > r2_pad) continue;
          const uint32_t j1 = sst(sg + 1);
          for (uint32_t j = sst(sg); j < j1 && !linked; j += FX_RR_U) {
            float jx[FX_RR_U], jy[FX_RR_U], jz[FX_RR_U];
            fetch(std::integral_constant<uint32_t, FX_RR_U>{}, j, j1 - 1u, jx, jy, jz);
#pragma unroll
            for (uint32_t u = 0; u < FX_RR_U; ++u) linked |= dist2(q.x, q.y, q.z, jx[u], jy[u], jz[u]) < r2;
          }
          if (linked) uf_union(rparent, b, a);  // the two runs are one component now; more edges add nothing
        }
      }
      wave_sync_lds();
      wq_n = 0;
    };
    // the points of the earlier run of every near pair against the later run's box: the (pair, point) items are
    // laid end to end (roff, idle until the centroids, holds the pairs' first item) so that one trip tests 64 of
    // them whatever the runs' lengths
    uint32_t n_items = 0;
    for (uint32_t t0 = 0; t0 < n_rp; t0 += 64) {
      const uint32_t t = t0 + lane;
      uint32_t len = 0;
      if (t < n_rp) {
        const uint32_t a = rp[t] >> 16;
        len = sst(rseg[a + 1]) - sst(rseg[a]);
      }
      uint32_t tot;
      const uint32_t ex = block_excl_scan<64>(len, s_w, tot);
      if (t < n_rp) roff[t] = n_items + ex;
      n_items += tot;
    }
    wave_sync_lds();
    for (uint32_t w0 = 0; w0 < n_items; w0 += 64) {
      const uint32_t w = w0 + lane;
      bool ok = false;
      uint32_t item = 0;
      if (w < n_items) {
        uint32_t lo = 0, hi = n_rp - 1u;  // the pair of item w: the last one starting at or before it
        while (lo < hi) {
          const uint32_t mid = (lo + hi + 1u) >> 1;
          if (roff[mid] <= w)
            lo = mid;
          else
            hi = mid - 1u;
        }
        const uint32_t a = rp[lo] >> 16, b = rp[lo] & 0xffffu;
        const uint32_t i = sst(rseg[a]) + (w - roff[lo]);
        float3 q;
        fetch(std::integral_constant<uint32_t, 1>{}, i, i, &q.x, &q.y, &q.z);
        const float dx = fmaxf(fmaxf(rb(0, b) - q.x, q.x - rb(1, b)), 0.0f);
        const float dy = fmaxf(fmaxf(rb(2, b) - q.y, q.y - rb(3, b)), 0.0f);
        ok = !(dx * dx + dy * dy > r2_pad);
        item = i | (b << 16) | (a << 24);
      }
      const unsigned long long mk = __ballot(ok);
      if (mk) {
        if (ok) wq[wq_n + lanes_below(mk)] = item;
        wq_n += (uint32_t)__popcll(mk);
        if (wq_n > FX_RR_QUEUE - 64) drain();
      }
    }
    if (wq_n) drain();
  }
  wave_sync_lds();
  FX_STAMP(3);
  // ---- roots (read-only finds, then the owners overwrite), sizes = sums of run lengths
  uint32_t my_root[(RN + 63) / 64];
#pragma unroll
  for (uint32_t u = 0; u < (RN + 63) / 64; ++u) {
    const uint32_t r = lane + 64 * u;
    my_root[u] = r < n_runs ? uf_find_ro(rparent, r) : 0u;
  }
  wave_sync_lds();
#pragma unroll
  for (uint32_t u = 0; u < (RN + 63) / 64; ++u) {
    const uint32_t r = lane + 64 * u;
    if (r < n_runs) {
      rparent[r] = my_root[u];
      atomicAdd(&rsize[my_root[u]], sst(rseg[r + 1]) - sst(rseg[r]));
    }
  }
  wave_sync_lds();
  FX_STAMP(4);
  // ---- PCL's cluster order (unchanged code: the "points" are the runs, a component's root its smallest run)
  const uint32_t n_c = cc_order<64>(n_runs, rparent, rsize, P.min_count, P.max_count, croot, crec, ctmp, CC, s_w, stamp_base);
  if (n_c > CC) return false;
#ifdef FX_STAMPS
  stamp_prev_ = __builtin_amdgcn_s_memtime();
#endif
  uint32_t *bb = reinterpret_cast<uint32_t *>(cc);  // [s][min x, max x, min y, max y] until the centroids go there
  for (uint32_t sI = lane; sI < n_c; sI += 64) {
    const uint32_t root = cl_root(crec[sI] & 0xffffu);
    rsize[root] |= (sI + 1u) << 16;  // position in PCL's order, next to the size
    bb[4 * sI + 0] = f2ord(1000.0f);  // ref: node.cpp:289-290
    bb[4 * sI + 1] = f2ord(-1000.0f);
    bb[4 * sI + 2] = f2ord(1000.0f);
    bb[4 * sI + 3] = f2ord(-1000.0f);
  }
  wave_sync_lds();
  for (uint32_t sg = lane; sg < n_segs; sg += 64) {
    const uint32_t pos = rsize[rparent[srun(sg)]] >> 16;
    if (pos == 0) continue;  // not a size-admissible cluster
    uint32_t *bx = bb + 4 * (pos - 1u);
    const float4 sb = sbox4[sg];
    atomicMin(&bx[0], f2ord(sb.x));
    atomicMax(&bx[1], f2ord(sb.y));
    atomicMin(&bx[2], f2ord(sb.z));
    atomicMax(&bx[3], f2ord(sb.w));
  }
  wave_sync_lds();
  FX_STAMP(7);
  // ---- diameter gate per cluster, in PCL's cluster order (ref: node.cpp:314-316)
  for (uint32_t sI = lane; sI < n_c; sI += 64) {
    const double minx = ord2f(bb[4 * sI + 0]), maxx = ord2f(bb[4 * sI + 1]);
    const double miny = ord2f(bb[4 * sI + 2]), maxy = ord2f(bb[4 * sI + 3]);
    const double ddx = maxx - minx, ddy = maxy - miny;
    cl_set_slot(sI, (sqrt(ddx * ddx + ddy * ddy) < P.gate_diameter) ? 1u : 0u);
  }
  wave_sync_lds();
  FX_STAMP(8);
  // ---- centroid of the clusters that pass: fp64 sums in ascending member order = the cluster's runs in run order,
  //      each run's points in turn (ref: node.cpp:293-297, 317-320)
  for (uint32_t sI = lane; sI < n_c; sI += 64) {
    if (cl_slot(sI) == 0u) continue;
    const uint32_t rec = crec[sI];
    const uint32_t sz = rec >> 16, root = cl_root(rec & 0xffffu);
    double sumx = 0.0, sumy = 0.0, sumz = 0.0;
    uint32_t cnt = 0;
    for (uint32_t r = root; r < n_runs && cnt < sz; ++r) {
      if (rparent[r] != root) continue;
      roff[r] = cnt;
      const uint32_t i0 = sst(rseg[r]), i1 = sst(rseg[r + 1]);
      for (uint32_t i = i0; i < i1; i += 4) {
        float x[4], y[4], z[4];
        fetch(std::integral_constant<uint32_t, 4>{}, i, i1 - 1u, x, y, z);
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
          if (i + u >= i1) continue;
          sumx += (double)x[u];
          sumy += (double)y[u];
          sumz += (double)z[u];
        }
      }
      cnt += i1 - i0;
    }
    cc[sI] = make_float4((float)(sumx / (double)sz), (float)(sumy / (double)sz), (float)(sumz / (double)sz), rw[root]);
  }
  wave_sync_lds();
  FX_STAMP(9);
  // ---- slots of the gate-passing clusters and offsets of their member runs
  uint32_t n_pass = 0, n_mem = 0;
  for (uint32_t b0 = 0; b0 < n_c; b0 += 64) {
    const uint32_t sI = b0 + lane;
    const bool pass = sI < n_c && cl_slot(sI) != 0u;
    const uint32_t sz = pass ? (crec[sI] >> 16) : 0u;
    uint32_t tot_p, tot_m;
    const uint32_t slot = block_rank<64>(pass, s_w, tot_p);
    const uint32_t koff = block_excl_scan<64>(sz, s_w, tot_m);
    if (sI < n_c) {
      cl_set_slot(sI, pass ? (n_pass + slot) : kNoSlot);
      ctmp[sI] = n_mem + koff;
    }
    n_pass += tot_p;
    n_mem += tot_m;
  }
  wave_sync_lds();
  FX_STAMP(10);
  // ---- candidates of this ring (cylinderCentroids, ref: node.cpp:322)
  float4 *rc = B.ring_cand + ring_slot * P.max_ring_cands;
  uint32_t *rcs = B.ring_cand_size + ring_slot * P.max_ring_cands;
  for (uint32_t sI = lane; sI < n_c; sI += 64) {
    const uint32_t slot = cl_slot(sI);
    if (slot != kNoSlot && slot < P.max_ring_cands) {
      rc[slot] = cc[sI];
      rcs[slot] = crec[sI] >> 16;
    }
  }
  if (lane == 0) {
    if (n_pass > P.max_ring_cands) atomicOr(&B.flags[scan], FX_FLAG_CAND_OVERFLOW);
    B.ring_cand_cnt[ring_slot] = n_pass < P.max_ring_cands ? n_pass : P.max_ring_cands;
    B.kpc_ring_cnt[ring_slot] = n_mem;
  }
  // ---- member points (cylinderCloud, ref: node.cpp:310, 323), a point per lane: the first 256 are still in registers
  if (n_mem) {
    float4 *pool = B.kpc_pool + (size_t)scan * P.ring_slot_cap + off;
    uint32_t *pool_c = B.kpc_pool_cand + (size_t)scan * P.ring_slot_cap + off;
    auto emit = [&](uint32_t i, const float4 &q, uint32_t sg) {
      const uint32_t r = srun(sg);
      const uint32_t pos = rsize[rparent[r]] >> 16;
      if (pos == 0) return;
      const uint32_t slot = cl_slot(pos - 1u);
      if (slot == kNoSlot) return;
      const uint32_t dst = ctmp[pos - 1u] + roff[r] + (i - sst(rseg[r]));
      pool[dst] = q;
      pool_c[dst] = slot;
    };
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      if (64 * k + lane < n) emit(64 * k + lane, qv[k], sgv[k]);
    }
    for (uint32_t i = 256 + lane; i < n; i += 64) {
      const float4 q = src[i];
      uint32_t lo = 0, hi = n_segs - 1u;  // the segment of point i: the last one starting at or before it
      while (lo < hi) {
        const uint32_t mid = (lo + hi + 1u) >> 1;
        if (sst(mid) <= i)
          lo = mid;
        else
          hi = mid - 1u;
      }
      emit(i, q, lo);
    }
  }
  wave_sync_lds();
  FX_STAMP(11);
  return true;
}
#ifndef FX_RUNS_OCC
#define FX_RUNS_OCC 4  // wavefronts a SIMD the register budget is held to (128 registers)
#endif
extern "C" __global__ __launch_bounds__(64, FX_RUNS_OCC) void k_rings_runs(FxDevParams P, FxBuffers B, uint32_t max_pts, uint32_t n_items) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  // persistent wavefronts over the (scan, ring) items, dealt by XCD class
  const uint32_t R = (uint32_t)P.n_rings, n_scans = n_items / R;
  const uint32_t cls = blockIdx.x & 7u, slot = blockIdx.x >> 3, per_cls = gridDim.x >> 3;  // (grid is a multiple of 8)
  if (n_scans < 64u) {
    // a handful of scans (streaming): a ring per wavefront, whatever its XCD — dealt by class, one scan's sixteen rings were
    // two wavefronts' work, eight rings one after the other (0.117 ms of a 0.37 ms host-to-host call)
    const uint32_t item = blockIdx.x;  // (the launcher's grid covers the items)
    if (item < n_items && !ring_runs_body<FX_RR_S, FX_RR_RN>(P, B, item / R, item % R, max_pts, smem)) {
      if (threadIdx.x == 0) {
        const uint32_t pos = atomicAdd(&B.counters[FX_CNT_LARGE + cls], 1u);
        B.huge_rings[(size_t)cls * P.ring_list_cap + pos] = item;
      }
    }
    return;
  }
  const uint32_t cls_items = ((n_scans + 7u - cls) / 8u) * R;
  for (uint32_t q = slot; q < cls_items; q += per_cls) {
    const uint32_t scan = cls + 8u * (q / R), ring = q % R;
    const uint32_t item = scan * R + ring;
    if (!ring_runs_body<FX_RR_S, FX_RR_RN>(P, B, scan, ring, max_pts, smem)) {
      if (threadIdx.x == 0) {  // the next tier's list of this XCD class (it too then finds the ring's points in its own L2)
        const uint32_t pos = atomicAdd(&B.counters[FX_CNT_LARGE + cls], 1u);
        B.huge_rings[(size_t)cls * P.ring_list_cap + pos] = item;
      }
    }
    wave_sync_lds();
  }
}
// second run tier (sensors of more than 16 rings only): the rings the first one handed over, with tables twice to three
// times as long; what does not fit these either goes on to the workgroup tier
extern "C" __global__ __launch_bounds__(64) void k_rings_runs2(FxDevParams P, FxBuffers B, uint32_t max_pts) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t cls = blockIdx.x & 7u;  // block b takes the list of XCD class b mod 8 (the grid is a multiple of 8)
  const uint32_t n_big = B.counters[FX_CNT_LARGE + cls];
  const uint32_t *items = B.huge_rings + (size_t)cls * P.ring_list_cap;
  // (rings by ticket: these cost 0.1 - 0.2 ms each and differ; dealt by stride, a batch of config 3 — 434 rings a class on
  //  192 wavefronts — took as long as the wavefront with three of them)
  while (true) {
    uint32_t w = 0;
    if (threadIdx.x == 0) w = atomicAdd(&B.counters[FX_CNT_RUNS2_TICKET + cls], 1u);
    w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
    if (w >= n_big) break;
    const uint32_t item = items[w];
    if (!ring_runs_body<384, 256>(P, B, item / P.n_rings, item % P.n_rings, max_pts, smem)) {
      if (threadIdx.x == 0) {
        const uint32_t pos = atomicAdd(&B.counters[FX_CNT_LARGE2 + cls], 1u);
        B.huge_rings2[(size_t)cls * P.ring_list_cap + pos] = item;
      }
    }
    wave_sync_lds();
  }
}

// workgroup tier: the rings the run tier hands over (more than 128 runs, segments or clusters: unordered input, dense
// many-ring sensors) get a whole 1024-thread workgroup with every point in LDS — one per CU, by LDS.  (A 256-thread tier
// between the two, five per CU, made BASELINE configs 3 and 5 slower, not faster, once the run tier existed.)
#define FX_RING_LARGE_T 1024
extern "C" __global__ __launch_bounds__(FX_RING_LARGE_T) void k_rings_large(FxDevParams P, FxBuffers B, uint32_t cap, uint32_t ccap,
                                                                           uint32_t after_runs2) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t cls = blockIdx.x & 7u;  // block b takes the list of XCD class b mod 8 (the grid is a multiple of 8): see k_rings_runs
  const uint32_t n_big = B.counters[(after_runs2 ? FX_CNT_LARGE2 : FX_CNT_LARGE) + cls];
  const uint32_t *items = (after_runs2 ? B.huge_rings2 : B.huge_rings) + (size_t)cls * P.ring_list_cap;
  for (uint32_t w = blockIdx.x >> 3; w < n_big; w += gridDim.x >> 3) {
    const uint32_t item = items[w];
    // (more points than this tier's LDS holds: the slow tier, which also flags what exceeds limits.max_ring_points)
    if (!ring_body<FX_RING_LARGE_T>(P, B, item / P.n_rings, item % P.n_rings, cap, ccap, smem, false) && threadIdx.x == 0)
      slow_ring(P, B, item / P.n_rings, item % P.n_rings);
    __syncthreads();
  }
}

// ====================================================================== stage 3: merge
// Secondary merge (ref: node.cpp:209-257): keypoints_full = the per-ring candidates in ring order (:205), z replaced
// by the scaled elevation (:217), a second pcl::EuclideanClusterExtraction (:222-229), centroids of the clusters'
// true xyz -> keypoints (:238-257); and the scan's keypoint_cloud chunks laid out in ring order (:206).
//
// Candidates are not azimuth ordered (each ring lists its clusters in PCL's size order), so the ring kernels' run
// labelling finds nothing here.  Instead the candidate ids are counting-sorted by xy cell:
//  * cells are `tolerance` wide, so every pair closer than the tolerance lies in the same or in adjacent cells (the
//    pseudo z only adds distance); the cell table is periodic — cells a period apart share a bin, which only costs
//    distance tests: every pair that could be an edge is examined, and tested with the exact predicate;
//  * every candidate tests the later candidates of the nine bins around it, skipping pairs already in one
//    component; links go through the lock-free union-find of the ring kernels (root = smallest index = PCL's
//    discovery order);
//  * sizes, then PCL's cluster order (cc_order);
//  * the members of every admissible cluster are listed, ranked by index within their cluster (each member counts
//    the smaller ids of its list), and one lane per cluster adds them in ascending index order in fp64 — the
//    reference's loop order (ref: node.cpp:242-253).
// Three tiers share this code: coordinates in LDS (k_merge_small: <= 512 candidates, one workgroup per scan;
// k_merge_big: what fits 160 KB), or — scans with more candidates than LDS holds as points (a 128-ring scan under the
// launch preset has ~5000) — coordinates left in HBM (k_merge_huge: the scan's `cand` rows, L2 resident).
#ifndef FX_MERGE_BIN_TICKET
#define FX_MERGE_BIN_TICKET 8u  // bins a wavefront of k_merge_huge's pair loop draws at a time (a power of two)
#endif
#ifndef FX_MERGE_HUGE_U
#define FX_MERGE_HUGE_U 8  // entries of a bin k_merge_huge's pair loop loads per trip
#endif
#define FX_MERGE_HEAD 160  // scratch words in front: block helpers [0..15], broadcast [16..31], sort stack [32..151]
__host__ __device__ constexpr uint32_t merge_bins(uint32_t cap) { return cap <= 1024u ? 1024u : 4096u; }
__host__ __device__ constexpr uint32_t merge_aux_words(uint32_t cap) {
  // bin table + candidate ids (uint16); later the sizes
  return merge_bins(cap) + 4 + (cap + 1) / 2 > cap ? merge_bins(cap) + 4 + (cap + 1) / 2 : cap;
}
// LDS words: lds_pts tiers hold (x, y, pseudo z, elevation), true z and the member lists on chip
__host__ __device__ constexpr size_t merge_words(uint32_t cap, uint32_t ccap, uint32_t n_rings, bool lds_pts) {
  return FX_MERGE_HEAD + ((2 * (n_rings + 1) + 3) & ~3u) + (lds_pts ? 5 * (size_t)cap + cap : 0)  // pt, cz, member lists (2 x uint16 per candidate)
         + cap                                                                                       // parent
         + merge_aux_words(cap) + 3 * (size_t)ccap;                                                  // croot, crec, tmp / member-list bases
}

// largest r with base[r] <= idx, base = exclusive prefix with base[R] = total > idx
__device__ __forceinline__ uint32_t prefix_owner(const uint32_t *base, uint32_t R, uint32_t idx) {
  uint32_t lo = 0, hi = R;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (base[mid] <= idx)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// The candidates of a scan as the fused front kernel (k_front) leaves them in LDS: per ring-cluster position g in PCL's
// order, rec[g] = size << 16 | root run, slot[g] = bit 31: the cluster passed the diameter gate | its ordinal in
// keypoints_full; centroid[root run] = (x, y, z, elevation).
struct FrontCands {
  const uint32_t *rec, *slot;
  const float4 *centroid;
  uint32_t n_c, C;
};
// Returns false when the scan has more candidates than this tier holds (the caller defers it to the next one).
// FRONT: the candidates come from k_front's LDS tables (FC) instead of the ring kernels' rows in HBM, and keypoint_cloud
// has been written by the caller.
// GS (with LDS_PTS false): parents, bin table, ids, cluster tables and the bin-ordered copy live in the scratch region `gs` of
// HBM instead of LDS (k_slow: any candidate count the limits allow); the head and the ring bases stay in LDS.
// PHASE (the large tier, coordinates in HBM, for batches of FEW scans: 64 scans of config 5 are 64 workgroups on 256 CUs, and
// 97 % of k_merge_huge is the pair loop — dependent loads a bin, sixteen wavefronts a scan to hide them with): the body as THREE
// launches.  1: candidates in, cell sort, the bin table to the scan's region `hp`; 2: the pair loop alone, by SEVERAL
// workgroups a scan — slice s takes the bins by ticket from a counter in `hp` and unites what it finds in a union-find of
// its OWN in LDS (a union-find in HBM shared by the slices would pay an L2 or memory round trip per hop of every find), then
// writes every candidate's root in its forest to `hp`; 3: one workgroup a scan unites the slices' forests — the edges
// (candidate, its root in slice s) of all slices have the components of all pairs — and does everything after the pair loop
// as before.  0: all of it in one launch (one workgroup a scan).  Which of the two runs is the host's choice by the batch's
// size: the same edges, the same components, the same result (roots are their components' smallest members either way).
// hp: [FX_MERGE_SLICES][cap] roots by slice, [merge_bins(cap) + 4] bin table, [0] status (candidates; FX_NONE: not this
// tier's / nothing to do), [1] ticket, [2] slices of the pair launch.
#define FX_MERGE_SLICES 8u
__host__ __device__ constexpr size_t merge_hp_words(uint32_t cap) { return ((size_t)FX_MERGE_SLICES * cap + merge_bins(cap) + 4 + 4 + 3) & ~(size_t)3; }
template <int NT, bool LDS_PTS, bool FRONT = false, bool GS = false, int PHASE = 0>
__device__ __forceinline__ bool merge_body(const FxDevParams &P, const FxBuffers &B, uint32_t scan, uint32_t cap, uint32_t ccap,
                                           uint32_t *smem, bool last_tier, const FrontCands *FC = nullptr, uint32_t *gs = nullptr,
                                           uint32_t *hp = nullptr, uint32_t slice = 0u) {
  static_assert(!GS || (!LDS_PTS && !FRONT), "the scratch tier keeps the coordinates in HBM");
  static_assert(PHASE == 0 || (!GS && !LDS_PTS && !FRONT), "the phases are the large tier's");
  uint32_t *const hp_bin = hp + (size_t)FX_MERGE_SLICES * cap, *const hp_state = hp_bin + merge_bins(cap) + 4;
  unsigned long long *const stamp_base = B.stamps ? B.stamps + 32 : nullptr;
  FX_STAMP_INIT(stamp_base);
  const uint32_t tid = threadIdx.x;
  const uint32_t R = (uint32_t)P.n_rings;
  const uint32_t NB = merge_bins(cap), nb_mask = NB == 1024u ? 31u : 63u, nb_shift = NB == 1024u ? 5u : 6u;
  uint32_t *s_w = smem;
  uint32_t *rbase = smem + FX_MERGE_HEAD;  // [R + 1]
  uint32_t *kbase = rbase + (R + 1);       // [R + 1]
  uint32_t *p = GS ? gs : smem + FX_MERGE_HEAD + ((2 * (R + 1) + 3) & ~3u);
  float4 *gs_sorted = nullptr;  // GS: the bin-ordered copy of the merge points
  if (GS) gs_sorted = reinterpret_cast<float4 *>(p), p += 4 * (size_t)cap;
  float4 *pt = nullptr;   // (x, y, pseudo z, elevation)
  float *cz = nullptr;    // true z
  uint16_t *mlist = nullptr;  // [2][cap]: members of the admissible clusters, unordered then ordered
  if (LDS_PTS) {
    pt = reinterpret_cast<float4 *>(p), p += 4 * cap;
    cz = reinterpret_cast<float *>(p), p += cap;
    mlist = reinterpret_cast<uint16_t *>(p), p += cap;
  }
  uint32_t *parent = p;
  p += cap;
  uint32_t *aux = p;
  p += merge_aux_words(cap);
  uint32_t *croot = p, *crec = croot + ccap, *tmp = crec + ccap;
  uint32_t *bin = aux;                                                  // [NB + 1]: counts -> starts -> ends
  uint16_t *sorted = reinterpret_cast<uint16_t *>(aux + NB + 4);       // candidate ids, bin by bin
  uint32_t *csize = aux;                                                // once the links are made
  const uint32_t *rcnt = B.ring_cand_cnt + (size_t)scan * R;

  // ---- ring bases
  uint32_t C = 0;
  if (FRONT) {
    C = FC->C;
  } else {
    for (uint32_t b0 = 0; b0 < R; b0 += NT) {
      const uint32_t r = b0 + tid;
      const uint32_t c = r < R ? rcnt[r] : 0u;
      uint32_t tot;
      const uint32_t ex = block_excl_scan<NT>(c, s_w, tot);
      if (r < R) rbase[r] = C + ex;
      C += tot;
    }
    if (tid == 0) rbase[R] = C;
  }
  if (PHASE >= 2 && hp_state[0] == FX_NONE) return true;  // (phase 1 settled the scan: overflow flagged, or handed to the slow tier)
  if (PHASE == 2 && C == 0) return true;
  if (PHASE <= 1 && C > P.max_candidates) {
    if (tid == 0) {
      atomicOr(&B.flags[scan], FX_FLAG_CAND_OVERFLOW);
      B.n_cand[scan] = 0;
      B.n_kp[scan] = 0;
      B.n_kpc[scan] = 0;
      if (PHASE == 1) hp_state[0] = FX_NONE;
    }
    return true;
  }
  if (PHASE <= 1 && C > cap && !last_tier) {  // (cap == max_candidates in the last tier)
    if (PHASE == 1 && tid == 0) hp_state[0] = FX_NONE;
    return false;
  }
  if (PHASE <= 1) {
    for (uint32_t t = tid; t <= NB; t += NT) bin[t] = 0;
  }
  wg_sync<GS>();

  float4 *cand = B.cand + (size_t)scan * P.max_candidates;
  uint32_t *cand_size = B.cand_size + (size_t)scan * P.max_candidates;
  int32_t *cand_kp = B.cand_kp + (size_t)scan * P.max_candidates;
  if (!LDS_PTS) mlist = reinterpret_cast<uint16_t *>(cand_kp);  // the row is scratch until its values are written at the end
  const float inv_w = 1.0f / (sqrtf(P.r2_merge) * 1.01f);
  auto bin_of = [&](int cx, int cy) { return (uint32_t)((cx & (int)nb_mask) | ((cy & (int)nb_mask) << nb_shift)); };
  auto cell_x = [&](float x) { return (int)floorf((x - P.x_min) * inv_w); };
  auto cell_y = [&](float y) { return (int)floorf((y - P.y_min) * inv_w); };
  // ref: node.cpp:217  z = intensity*0.75*clusterRadiusThreshold/2  (double, left to right)
  auto pseudo_z = [&](float el) { return (float)((double)el * 0.75 * P.crt / 2); };
  auto merge_pt = [&](uint32_t i) {  // (x, y, pseudo z, elevation)
    if (LDS_PTS) return pt[i];
    const float4 v = cand[i];
    return make_float4(v.x, v.y, pseudo_z(v.w), v.w);
  };
  auto true_pt = [&](uint32_t i) {  // (x, y, true z, elevation)
    if (LDS_PTS) {
      const float4 q = pt[i];
      return make_float4(q.x, q.y, cz[i], q.w);
    }
    return cand[i];
  };
  // all candidates in parallel (each finds its ring in the prefix table)
  for (uint32_t t = tid; PHASE <= 1 && t < (FRONT ? FC->n_c : C); t += NT) {
    uint32_t idx = t, size;
    float4 v;
    if (FRONT) {
      const uint32_t w = FC->slot[t];
      if (!(w >> 31)) continue;  // the cluster did not pass the gate
      idx = w & 0x7fffffffu;
      const uint32_t rec = FC->rec[t];
      v = FC->centroid[rec & 0xffffu];
      size = rec >> 16;
    } else {
      const uint32_t r = prefix_owner(rbase, R, idx), j = idx - rbase[r];
      v = B.ring_cand[((size_t)scan * R + r) * P.max_ring_cands + j];
      size = B.ring_cand_size[((size_t)scan * R + r) * P.max_ring_cands + j];
    }
    if (LDS_PTS) {
      pt[idx] = make_float4(v.x, v.y, pseudo_z(v.w), v.w);
      cz[idx] = v.z;
    }
    cand[idx] = v;
    cand_size[idx] = size;
    if (PHASE == 0) parent[idx] = idx;
    atomicAdd(&bin[bin_of(cell_x(v.x), cell_y(v.y))], 1u);
  }
  wg_sync<GS>();
  if (!LDS_PTS) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // `cand` is re-read below by other waves of this workgroup (see wg_global_sync)
  FX_STAMP(1);
  uint32_t n_c = 0;
  if (PHASE == 1 && C == 0) {  // (nothing to sort or pair: phase 3 writes the empty results)
    if (tid == 0) hp_state[0] = 0u;
    return true;
  }
  if (C > 0) {  // ref: node.cpp:209-210
    if (PHASE <= 1 && tid < 64) {  // counts -> exclusive starts, in place, by one wavefront
      const uint32_t per = NB / 64;
      uint32_t sum = 0;
      for (uint32_t u = 0; u < per; ++u) sum += bin[tid * per + u];
      uint32_t incl = sum;
      incl = wave_incl_scan(incl);
      uint32_t run = incl - sum;
      for (uint32_t u = 0; u < per; ++u) {
        const uint32_t c = bin[tid * per + u];
        bin[tid * per + u] = run;
        run += c;
      }
    }
    wg_sync<GS>();
    // (coordinates left in HBM: a copy of the merge points in bin order, id in .w, so that the pair tests below read a bin's
    //  entries from consecutive addresses instead of one dependent L2 load per id — 86 % of k_merge_huge was that loop)
    float4 *msort = LDS_PTS ? nullptr : (GS ? gs_sorted : B.merge_sorted + (size_t)scan * P.max_candidates);
    for (uint32_t idx = tid; PHASE <= 1 && idx < C; idx += NT) {  // each id to its bin: a start becomes the bin's end
      const float4 v = merge_pt(idx);
      const uint32_t pos = atomicAdd(&bin[bin_of(cell_x(v.x), cell_y(v.y))], 1u);
      sorted[pos] = (uint16_t)idx;
      if (!LDS_PTS) msort[pos] = make_float4(v.x, v.y, v.z, __uint_as_float(idx));
    }
    if (tid == 0) s_w[152] = 0u;  // (the pair loop's bin ticket)
    wg_sync<GS>();
    if (PHASE == 1) {  // the bin table (ends) for the pair launch's workgroups; the scan's state: its candidates, ticket 0
      for (uint32_t t = tid; t <= NB; t += NT) hp_bin[t] = bin[t];
      if (tid == 0) hp_state[0] = C, hp_state[1] = 0u;  // (hp_state[2], the slices, is the pair launch's own)
      return true;
    }
    if (PHASE == 2) {  // this slice's own forest; the bin table
      for (uint32_t i = tid; i < C; i += NT) parent[i] = i;
      for (uint32_t t = tid; t <= NB; t += NT) bin[t] = hp_bin[t];
      wg_sync<GS>();
    }
    if (!LDS_PTS) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (msort is read below by other waves of this workgroup)
    FX_STAMP(2);
    // ---- pcl::EuclideanClusterExtraction on (x, y, pseudo z) (ref: node.cpp:222-229)
    // (one work item per (candidate, neighbouring bin): a wavefront's trip count is then the longest single bin of its
    //  lanes, not the sum over nine bins of the longest; two entries per trip, loaded before either is used; the
    //  distance test comes before any union-find lookup — one LDS round trip against several dependent ones)
    if (!LDS_PTS && PHASE != 3) {
      // Coordinates in HBM (k_merge_huge): bin by bin.  A wavefront takes a bin T, its entries 64 at a time one per lane
      // (the TARGETS: one coalesced load), and walks the entries of the nine bins around it (the SOURCES: up to 64 of
      // their concatenation per coalesced load, handed round by readlane).  Every lane then tests its target against the
      // same source — all lanes busy in a pole's bin (a candidate per ring: 100 x 100 pairs), eight cache lines per load
      // where one work item per (candidate, bin) with its own stride through the bin cost a cache line per entry and lane:
      // 550 000 scattered 16-byte loads a scan of config 5, the L1's whole time.  Pairs that are already in one set
      // (after the first few unions of a pole: nearly all) are recognised by their parents before any union is tried.
      const uint32_t lane = tid & 63u, wave = tid >> 6;
      const int nbm = (int)nb_mask;
      auto pair_bin = [&](uint32_t T) {  // the targets of bin T against the sources of the nine bins around it
        const uint32_t t0 = T ? bin[T - 1u] : 0u, t1 = bin[T];
        if (t1 == t0) return;
        const int cxT = (int)(T & nb_mask), cyT = (int)(T >> nb_shift);
        uint32_t st[9], cum[10];  // the nine source bins: first entry, entries before it in their concatenation
        cum[0] = 0u;
#pragma unroll
        for (int d = 0; d < 9; ++d) {
          const uint32_t S = (uint32_t)(((cxT + d % 3 - 1) & nbm) | (((cyT + d / 3 - 1) & nbm) << nb_shift));
          const uint32_t s0 = S ? bin[S - 1u] : 0u;
          st[d] = s0;
          cum[d + 1] = cum[d] + (bin[S] - s0);
        }
        const uint32_t total = cum[9];
        for (uint32_t tc = t0; tc < t1; tc += 64u) {
          const bool has = tc + lane < t1;
          const float4 vj = msort[min(tc + lane, t1 - 1u)];
          const uint32_t j = has ? __float_as_uint(vj.w) : 0u;  // (0: never larger than a source's id)
          uint32_t rj = has ? uf_find<GS>(parent, j) : 0u;
          for (uint32_t base = 0; base < total; base += 64u) {
            const uint32_t g = min(base + lane, total - 1u);
            uint32_t s_at = st[0] + g;
#pragma unroll
            for (int d = 1; d < 9; ++d)
              if (g >= cum[d]) s_at = st[d] + (g - cum[d]);
            const float4 vs = msort[s_at];
            const uint32_t n = min(64u, total - base);
            for (uint32_t e = 0; e < n; ++e) {
              const float sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vs.x), (int)e));
              const float sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vs.y), (int)e));
              const float sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vs.z), (int)e));
              const uint32_t i = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(vs.w), (int)e);
              // every pair once (j > i)
#ifdef FX_STAMPS
              if (B.stamps && lane == 0) atomicAdd(&B.stamps[60], 1ull);
              if (B.stamps && j > i && dist2(sx, sy, sz, vj.x, vj.y, vj.z) < P.r2_merge) {
                atomicAdd(&B.stamps[61], 1ull);
                if (uf_load<GS>(parent, i) != rj) atomicAdd(&B.stamps[62], 1ull);
              }
#endif
              if (j > i && dist2(sx, sy, sz, vj.x, vj.y, vj.z) < P.r2_merge) {
                const uint32_t pi = uf_load<GS>(parent, i);
                if (pi != rj) rj = uf_union_root<GS>(parent, rj, pi);  // (from the ancestors at hand: shorter finds, and the root comes back)
              }
            }
          }
        }
      };
      (void)wave;
      while (true) {  // bins by ticket, a few at a time: a pole's bin costs a hundred times an empty one, and behind the loop is a barrier
        uint32_t T0 = 0;
        if (lane == 0) T0 = atomicAdd(PHASE == 2 ? &hp_state[1] : &s_w[152], FX_MERGE_BIN_TICKET);
        T0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)T0);
        if (T0 >= NB) break;  // (NB is a multiple of the ticket)
        for (uint32_t T = T0; T < T0 + FX_MERGE_BIN_TICKET; ++T) pair_bin(T);
      }
    }
    for (uint32_t t = tid; LDS_PTS && t < 9u * C; t += NT) {
      const uint32_t i = t / 9u, d = t - 9u * i;
      const float4 v = merge_pt(i);
      const uint32_t b = bin_of(cell_x(v.x) + (int)(d % 3u) - 1, cell_y(v.y) + (int)(d / 3u) - 1);
      const uint32_t q0 = b ? bin[b - 1] : 0u, q1 = bin[b];
      for (uint32_t q = q0; q < q1; q += 4) {  // (four entries per trip, ids and points loaded before any is used: a pole's bin holds a candidate per ring)
        uint32_t j[4];
        float4 u[4];
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) j[e] = sorted[min(q + e, q1 - 1u)];
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) u[e] = merge_pt(j[e]);
        // every pair once (j > i)
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e)
          if (q + e < q1 && j[e] > i && dist2(v.x, v.y, v.z, u[e].x, u[e].y, u[e].z) < P.r2_merge) {
            // (already under one parent — after the first few unions of a pole that many rings saw: most pairs — is two reads
            //  instead of two finds: k_merge_big on config 3 0.208 -> 0.182 ms; not inside k_front, where a pole has a
            //  candidate per ring of sixteen and the reads cost more than they save)
            if (FRONT || uf_load<GS>(parent, j[e]) != uf_load<GS>(parent, i)) uf_union<GS>(parent, j[e], i);
          }
      }
    }
    if (PHASE == 2) {  // every candidate's root in this slice's forest (the launch's end is the barrier between the slices)
      wg_sync<GS>();
      uint32_t *mine = hp + (size_t)slice * cap;
      for (uint32_t i = tid; i < C; i += NT) mine[i] = uf_find_ro<GS>(parent, i);
      return true;
    }
    if (PHASE == 3) {  // the slices' forests united: (candidate, its root in slice s) are edges enough
      for (uint32_t i = tid; i < C; i += NT) parent[i] = i;
      wg_sync<GS>();
      const uint32_t S = hp_state[2];
      for (uint32_t sl = 0; sl < S; ++sl) {
        const uint32_t *theirs = hp + (size_t)sl * cap;
        for (uint32_t i = tid; i < C; i += NT) {
          const uint32_t r = theirs[i];
          if (r != i && uf_load<GS>(parent, i) != uf_load<GS>(parent, r)) uf_union<GS>(parent, i, r);
        }
      }
    }
#if defined(FX_MERGE_STOP) && FX_MERGE_STOP == 1
    if (PHASE == 3) return true;
#endif
    wg_sync<GS>();
    FX_STAMP(3);
    for (uint32_t i = tid; i < C; i += NT) parent[i] = uf_find_ro<GS>(parent, i);
    for (uint32_t i = tid; i < C; i += NT) csize[i] = 0u;  // (bin table and ids are done with)
    wg_sync<GS>();
    for (uint32_t i = tid; i < C; i += NT) atomicAdd(&csize[parent[i]], 1u);
    wg_sync<GS>();
#if defined(FX_MERGE_STOP) && FX_MERGE_STOP == 2
    if (PHASE == 3) return true;
#endif
    FX_STAMP(4);
    n_c = cc_order<NT, GS>(C, parent, csize, P.ndc, P.secondary_max, croot, crec, tmp, ccap, s_w, stamp_base);
    if (n_c > ccap) {  // (large tier only: ccap >= max_keypoints there, so the scan overflows its keypoints anyway)
      if (tid == 0) atomicOr(&B.flags[scan], FX_FLAG_KP_OVERFLOW);
      n_c = 0;
    }
#ifdef FX_STAMPS
    stamp_prev_ = __builtin_amdgcn_s_memtime();
#endif
  }

#if defined(FX_MERGE_STOP) && FX_MERGE_STOP == 3
    if (PHASE == 3) return true;
#endif
  // ---- clusters -> keypoints (ref: node.cpp:238-257)
  uint32_t K = 0;
  if (C > 0) {
    K = n_c < P.max_keypoints ? n_c : P.max_keypoints;
    if (n_c > P.max_keypoints && tid == 0) atomicOr(&B.flags[scan], FX_FLAG_KP_OVERFLOW);
    // position in PCL's order next to the size; crec[s] becomes (size, root index) and mbase[s] the start of the
    // cluster's member list (tmp is free after cc_order; croot is free once the roots are in crec)
    uint32_t *mbase = tmp, *mfill = croot;
    uint32_t run = 0;
    for (uint32_t b0 = 0; b0 < n_c; b0 += NT) {
      const uint32_t s = b0 + tid;
      uint32_t sz = 0, root = 0;
      if (s < n_c) {
        const uint32_t rec = crec[s];
        sz = rec >> 16;
        root = croot[rec & 0xffffu];
        csize[root] |= (s + 1u) << 16;
      }
      uint32_t tot;
      const uint32_t ex = block_excl_scan<NT>(sz, s_w, tot);
      if (s < n_c) {
        crec[s] = (sz << 16) | root;  // (root < 65536: candidate ids fit 16 bits)
        mbase[s] = run + ex;
      }
      run += tot;
    }
    wg_sync<GS>();
    for (uint32_t s = tid; s < n_c; s += NT) mfill[s] = 0u;
    wg_sync<GS>();
    uint16_t *unord = mlist, *ord = mlist + cap;
    for (uint32_t i = tid; i < C; i += NT) {
      const uint32_t pos = csize[parent[i]] >> 16;
      if (pos == 0) continue;
      unord[mbase[pos - 1u] + atomicAdd(&mfill[pos - 1u], 1u)] = (uint16_t)i;
    }
    wg_sync<GS>();
    if (!LDS_PTS) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (uint32_t i = tid; i < C; i += NT) {  // rank within the cluster = members with a smaller index
      const uint32_t pos = csize[parent[i]] >> 16;
      if (pos == 0) continue;
      const uint32_t s = pos - 1u, m0 = mbase[s], sz = crec[s] >> 16;
      uint32_t rank = 0;
      for (uint32_t m = 0; m < sz; ++m) rank += (uint32_t)unord[m0 + m] < i ? 1u : 0u;
      ord[m0 + rank] = (uint16_t)i;
    }
    wg_sync<GS>();
    if (!LDS_PTS) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#if defined(FX_MERGE_STOP) && FX_MERGE_STOP == 4
    if (PHASE == 3) return true;
#endif
    float4 *kp = B.keypoints + (size_t)scan * P.max_keypoints;
    uint32_t *kps = B.kp_size + (size_t)scan * P.max_keypoints;
    for (uint32_t s = tid; s < K; s += NT) {
      const uint32_t rec = crec[s];
      const uint32_t sz = rec >> 16, root = rec & 0xffffu, m0 = mbase[s];
      double sumx = 0.0, sumy = 0.0, sumz = 0.0;
      // (eight members a trip, their ids and then their points loaded before any is added: with the coordinates in HBM a member
      //  is two DEPENDENT loads — its id, its point —, and a pole that 128 rings saw was 256 round trips one after the other
      //  in the one lane that owns the keypoint: most of what k_merge_huge does after its pair loop; the sums stay in member order)
      for (uint32_t m = 0; m < sz; m += 8u) {
        uint32_t id[8];
        float4 q[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) id[u] = ord[m0 + min(m + u, sz - 1u)];
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) q[u] = true_pt(id[u]);
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u)
          if (m + u < sz) {
            sumx += (double)q[u].x;
            sumy += (double)q[u].y;
            sumz += (double)q[u].z;
          }
      }
      kp[s] = make_float4((float)(sumx / (double)sz), (float)(sumy / (double)sz), (float)(sumz / (double)sz), true_pt(root).w);
      kps[s] = sz;
      B.kp_nbrs[(size_t)scan * P.max_keypoints + s] = 0u;  // (k_gather flags the keypoints that have a neighbour; the descriptor kernels count)
    }
#if defined(FX_MERGE_STOP) && FX_MERGE_STOP == 5
    if (PHASE == 3) return true;
#endif
    wg_sync<GS>();
    FX_STAMP(8);
    for (uint32_t i = tid; i < C; i += NT) {
      const uint32_t pos = csize[parent[i]] >> 16;
      cand_kp[i] = (pos != 0 && pos - 1u < K) ? (int32_t)(pos - 1u) : -1;
    }
    FX_STAMP(9);
  }

  if (FRONT) {  // (keypoint_cloud and its count are the caller's)
    if (tid == 0) {
      B.n_cand[scan] = C;
      B.n_kp[scan] = K;
    }
    wg_sync<GS>();
    return true;
  }
  // ---- keypoint_cloud: chunks into ring order, candidate slot -> ordinal in keypoints_full
  const uint32_t *koff = B.ring_off + (size_t)scan * R;
  const uint32_t *kcnt = B.kpc_ring_cnt + (size_t)scan * R;
  const float4 *pool = B.kpc_pool + (size_t)scan * P.ring_slot_cap;
  const uint32_t *pool_c = B.kpc_pool_cand + (size_t)scan * P.ring_slot_cap;
  float4 *kpc = B.kpc + (size_t)scan * P.max_kpc;
  uint32_t *kpc_c = B.kpc_cand + (size_t)scan * P.max_kpc;
  uint32_t run = 0;
  for (uint32_t b0 = 0; b0 < R; b0 += NT) {
    const uint32_t r = b0 + tid;
    const uint32_t c = r < R ? kcnt[r] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan<NT>(c, s_w, tot);
    if (r < R) kbase[r] = run + ex;
    run += tot;
  }
  if (tid == 0) kbase[R] = run;
  wg_sync<GS>();
  FX_STAMP(10);
  if (run > P.max_kpc) {
    if (tid == 0) atomicOr(&B.flags[scan], FX_FLAG_KPC_OVERFLOW);
    run = 0;
  }
  // (four elements per trip, loaded before any is stored; the index is clamped so that no load is conditional)
  for (uint32_t t0 = tid; t0 < run; t0 += 4 * NT) {
    const uint32_t ta = t0, tb = min(t0 + NT, run - 1u), tc = min(t0 + 2 * NT, run - 1u), td = min(t0 + 3 * NT, run - 1u);
    const uint32_t ra = prefix_owner(kbase, R, ta), rb = prefix_owner(kbase, R, tb), rc = prefix_owner(kbase, R, tc),
                   rd = prefix_owner(kbase, R, td);
    const uint32_t sa = koff[ra] + (ta - kbase[ra]), sb = koff[rb] + (tb - kbase[rb]), sc = koff[rc] + (tc - kbase[rc]),
                   sd = koff[rd] + (td - kbase[rd]);
    const float4 va = pool[sa], vb = pool[sb], vc = pool[sc], vd = pool[sd];
    const uint32_t ca = rbase[ra] + pool_c[sa], cb = rbase[rb] + pool_c[sb], cc = rbase[rc] + pool_c[sc], cd = rbase[rd] + pool_c[sd];
    kpc[ta] = va, kpc_c[ta] = ca;  // (clamped duplicates rewrite the last element with the same value)
    kpc[tb] = vb, kpc_c[tb] = cb;
    kpc[tc] = vc, kpc_c[tc] = cc;
    kpc[td] = vd, kpc_c[td] = cd;
  }
  if (tid == 0) {
    B.n_cand[scan] = C;
    B.n_kp[scan] = K;
    B.n_kpc[scan] = run;
  }
  wg_sync<GS>();
  FX_STAMP(11);
  return true;
}

#ifndef FX_MSMALL_T
#define FX_MSMALL_T 256
#endif
extern "C" __global__ __launch_bounds__(FX_MSMALL_T) void k_merge_small(FxDevParams P, FxBuffers B, uint32_t cap) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  {  // the support-list counters of the batch are cleared here, a slice per scan (k_gather fills them)
    const uint32_t per = (P.max_total_kp + gridDim.x - 1) / gridDim.x;
    const uint32_t z0 = blockIdx.x * per, z1 = min(z0 + per, P.max_total_kp);
    for (uint32_t t = z0 + threadIdx.x; t < z1; t += FX_MSMALL_T) B.s_cnt[t] = 0u;
    if (threadIdx.x == 0) B.ovf_cnt[blockIdx.x] = 0u;  // entries in the scan's overflow region (k_gather)
  }
  // (a scan with a ring still waiting for the slow tier is merged there, once, from all its rings: a merge of the rings that
  //  are ready could split a cluster in two and flag a keypoint overflow the whole scan does not have)
  if (B.slow_state[blockIdx.x] != 0u) return;
  if (!merge_body<FX_MSMALL_T, true>(P, B, blockIdx.x, cap, cap, smem, false)) {
    if (threadIdx.x == 0) {
      const uint32_t pos = atomicAdd(&B.counters[1], 1u);
      B.big_merge[pos] = blockIdx.x;
    }
  }
}
#define FX_MBIG_T 1024
extern "C" __global__ __launch_bounds__(FX_MBIG_T) void k_merge_big(FxDevParams P, FxBuffers B, uint32_t cap, uint32_t last) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t n_big = B.counters[1];
  for (uint32_t w = blockIdx.x; w < n_big; w += gridDim.x) {
    const uint32_t scan = B.big_merge[w];
    if (!merge_body<FX_MBIG_T, true>(P, B, scan, cap, cap, smem, last != 0)) {
      if (threadIdx.x == 0) B.huge_merge[atomicAdd(&B.counters[9], 1u)] = scan;  // more candidates than LDS holds as points
    }
    __syncthreads();
  }
}
extern "C" __global__ __launch_bounds__(FX_MBIG_T) void k_merge_huge(FxDevParams P, FxBuffers B, uint32_t cap, uint32_t ccap) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t n_big = B.counters[9];
  for (uint32_t w = blockIdx.x; w < n_big; w += gridDim.x) {
    // (cap: what this tier's LDS holds; a scan with more candidates, up to limits.max_candidates, takes the slow tier)
    if (!merge_body<FX_MBIG_T, false>(P, B, B.huge_merge[w], cap, ccap, smem, cap >= P.max_candidates) && threadIdx.x == 0) slow_push(B, B.huge_merge[w]);
    __syncthreads();
  }
}

// the large tier as three launches (merge_body's PHASE): batches of few scans, several workgroups a scan in the pair loop
extern "C" __global__ __launch_bounds__(FX_MBIG_T) void k_merge_huge_a(FxDevParams P, FxBuffers B, uint32_t cap, uint32_t ccap) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t n_big = B.counters[9];
  for (uint32_t w = blockIdx.x; w < n_big; w += gridDim.x) {
    const uint32_t scan = B.huge_merge[w];
    if (!merge_body<FX_MBIG_T, false, false, false, 1>(P, B, scan, cap, ccap, smem, cap >= P.max_candidates, nullptr, nullptr,
                                                       B.merge_hp + (size_t)scan * merge_hp_words(cap)) && threadIdx.x == 0)
      slow_push(B, scan);  // (more candidates than this tier's LDS holds: the slow tier; the scan's state says so to the launches behind)
    __syncthreads();
  }
}
extern "C" __global__ __launch_bounds__(FX_MBIG_T) void k_merge_huge_b(FxDevParams P, FxBuffers B, uint32_t cap, uint32_t ccap) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t n_big = B.counters[9];
  for (uint32_t w = blockIdx.y; w < n_big; w += gridDim.y) {
    const uint32_t scan = B.huge_merge[w];
    uint32_t *hp = B.merge_hp + (size_t)scan * merge_hp_words(cap);
    if (blockIdx.x == 0 && threadIdx.x == 0) hp[(size_t)FX_MERGE_SLICES * cap + merge_bins(cap) + 4 + 2] = gridDim.x;  // (the slices phase 3 unites)
    merge_body<FX_MBIG_T, false, false, false, 2>(P, B, scan, cap, ccap, smem, true, nullptr, nullptr, hp, blockIdx.x);
    __syncthreads();
  }
}
extern "C" __global__ __launch_bounds__(FX_MBIG_T) void k_merge_huge_c(FxDevParams P, FxBuffers B, uint32_t cap, uint32_t ccap) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t n_big = B.counters[9];
  for (uint32_t w = blockIdx.x; w < n_big; w += gridDim.x) {
    const uint32_t scan = B.huge_merge[w];
    merge_body<FX_MBIG_T, false, false, false, 3>(P, B, scan, cap, ccap, smem, true, nullptr, nullptr, B.merge_hp + (size_t)scan * merge_hp_words(cap));
    __syncthreads();
  }
}

// ====================================================================== stages 1-3 in one launch: k_front
// Scans whose filtered cloud fits LDS (every VLP-16-class scan: ~2600 survivors of 28 800 points) go from the input to
// keypoints in ONE launch — filter, ring split, per-ring clustering, secondary merge — without their intermediates ever
// leaving the chip (ref: node.cpp:147-259 is one call chain).  The separate kernels wrote the filtered cloud, read it
// back to write the ring-major copy, read that to write candidates and member pools, and read those to merge:
// 0.26 GB of a 1024-scan batch's 1.12 GB, and four of its fifteen dependent launches.
//   A  the streaming pass of k_prep (prep_stream): ~cloud and near bits to HBM, ring counts in LDS;
//   B  the ring split of k_bucket with an LDS destination: ring-major x / y / z (+ the survivor's index: elevation and
//      the member copies come from ~cloud, which the workgroup has just written — L2 hits);
//   C  getCylinderSegments for ALL rings of the scan at once by the whole workgroup (a scan's rings are unequal — two
//      ground rings of ~500 points, most of ~70 — so a ring per wavefront would leave the workgroup waiting for its two
//      slowest wavefronts): the run tier's scheme (ring_runs_body) on one table of runs / 16-point segments over the
//      concatenated rings — runs never cross a ring start, near run pairs are looked for inside a ring only —; PCL's
//      cluster order per ring (the replay's partition phase: a ring per wavefront; its ranking: a cluster per thread);
//   D  the secondary merge (merge_body) on the candidates in LDS; keypoint_cloud written straight in its final order.
// Anything that does not fit the tables below hands the scan to k_front_redo, which runs the general kernels' bodies on it
// from ~cloud: more ring entries than FX_FRONT_CAP, more runs than FX_FRONT_RUNS, more near run pairs than
// FX_FRONT_PAIRS, more candidates than the small merge tier holds, or any of the limits the general kernels flag.
#define FX_FRONT_T FX_PREP_T
#define FX_FRONT_NW (FX_FRONT_T / 64)
#ifndef FX_FRONT_CAP
#define FX_FRONT_CAP (FX_PREP_KEEP + 256)  // ring-major entries (and survivors): the staging buffer of the streaming pass becomes the points (the bench scenes have 2580 on average, 3219 at most)
#endif
#ifndef FX_FRONT_RUNS
#define FX_FRONT_RUNS 512          // runs (and clusters) of all rings together (the bench scenes have 370 on average, 450 at most)
#endif
#define FX_FRONT_SEGS (FX_FRONT_CAP / 16 + FX_FRONT_RUNS)  // a segment starts at every run start and at every multiple of 16
#define FX_FRONT_RMAX 32           // rings
#define FX_FRONT_PAIRS 512         // near run pairs
#define FX_FRONT_QUEUE 96          // parked (point, run) items per wavefront
#define FX_FRONT_BLOCKS (FX_FRONT_CAP / 64)
#define FX_FRONT_MERGE 512         // candidates (k_merge_small's capacity)
#define FX_FRONT_SORTW (FX_SORT_STACK_WORDS + 192)  // per wavefront: the replay's stack and position tables (192 clusters a ring; beyond: one lane)
__host__ __device__ constexpr uint32_t a4(uint32_t v) { return (v + 3u) & ~3u; }
struct FrontOff {  // word offsets into the LDS image
  static constexpr uint32_t px = 0, py = FX_FRONT_CAP, pz = 2 * FX_FRONT_CAP, sidx = 3 * FX_FRONT_CAP;  // sidx: uint16
  static constexpr uint32_t s_w = sidx + FX_FRONT_CAP / 2;                // [64] block helpers 0..15, broadcast slots 16..
  static constexpr uint32_t r_off = s_w + 64;                             // [RMAX + 1] first ring-major entry of each ring
  static constexpr uint32_t r_cnt = r_off + FX_FRONT_RMAX + 4;            // [RMAX] entries per ring
  static constexpr uint32_t r_run0 = r_cnt + FX_FRONT_RMAX;               // [RMAX + 1] first run of each ring
  static constexpr uint32_t r_cb = r_run0 + FX_FRONT_RMAX + 4;            // [RMAX + 1] first cluster of each ring
  static constexpr uint32_t smask = r_cb + FX_FRONT_RMAX + 4;             // [BLOCKS] uint64: run starts
  static constexpr uint32_t gmask = smask + 2 * FX_FRONT_BLOCKS;          // [BLOCKS] uint64: segment starts
  static constexpr uint32_t run_base = gmask + 2 * FX_FRONT_BLOCKS;       // [BLOCKS] runs before the block
  static constexpr uint32_t seg_base = run_base + FX_FRONT_BLOCKS;        // [BLOCKS] segments before the block
  static constexpr uint32_t seg_start = seg_base + FX_FRONT_BLOCKS;       // uint16 [SEGS + 1]
  static constexpr uint32_t seg_box = a4(seg_start + (FX_FRONT_SEGS + 2) / 2);  // [SEGS][min x, max x, min y, max y]
  static constexpr uint32_t rbox = seg_box + 4 * FX_FRONT_SEGS;           // [RUNS][4] run boxes, then cluster boxes, then centroids
  static constexpr uint32_t rseg = rbox + 4 * FX_FRONT_RUNS;              // uint16 [RUNS + 1] first segment of each run
  static constexpr uint32_t rparent = a4(rseg + (FX_FRONT_RUNS + 2) / 2); // [RUNS] union-find over runs
  static constexpr uint32_t rsize = rparent + FX_FRONT_RUNS;              // [RUNS] points of the component | (position in PCL's order + 1) << 16
  static constexpr uint32_t roff = rsize + FX_FRONT_RUNS;                 // uint16 [RUNS] clusters before the run, then the run's offset inside its cluster
  static constexpr uint32_t croot = a4(roff + FX_FRONT_RUNS / 2);         // uint16 [RUNS + 8] root run by discovery ordinal, then member offsets
  static constexpr uint32_t crec = croot + FX_FRONT_RUNS / 2 + 4;         // [RUNS + 4]
  static constexpr uint32_t ctmp = crec + FX_FRONT_RUNS + 4;              // [RUNS + 4]
  static constexpr uint32_t end = ctmp + FX_FRONT_RUNS + 4;
  // overlays: the streaming pass's tables, the survivors' elevations and the split's counters live in the (not yet used)
  // segment / run tables, the ring-start
  // bits, the near pairs and the wavefronts' queues in the (not yet used) cluster tables, the replay's scratch in the
  // (no longer used) segment boxes
  static constexpr uint32_t atan = seg_box, win = atan + a4(2 * (FX_ATAN_N + 1) * (FX_ATAN_DEG + 1)), cnt = win + 2 * FX_FRONT_RMAX,
                            cw = cnt + 2 * FX_FRONT_NW, el = cw + 2 * FX_FRONT_NW * FX_FRONT_RMAX, a_end = el + FX_FRONT_CAP;
  static constexpr uint32_t rsm = croot, pairs = croot, queues = pairs + FX_FRONT_PAIRS, q_end = queues + FX_FRONT_NW * FX_FRONT_QUEUE;
};
static_assert(FrontOff::end * 4 <= 80 * 1024, "two workgroups of k_front a CU");
static_assert(FrontOff::a_end <= FrontOff::croot && FrontOff::q_end <= FrontOff::end, "k_front overlays");
static_assert(FX_FRONT_NW * FX_FRONT_SORTW <= 4 * FX_FRONT_SEGS, "k_front: the replay's scratch borrows the segment boxes");
static_assert(FX_FRONT_RUNS <= 1024 && FX_FRONT_CAP <= 4096 && FX_FRONT_BLOCKS <= 64 && FX_FRONT_RMAX <= 64 && FX_FRONT_CAP % 64 == 0 &&
                  FX_FRONT_CAP >= FX_PREP_KEEP, "k_front packings");
static_assert((FrontOff::smask % 2) == 0 && (FrontOff::rsm % 2) == 0 && (FrontOff::seg_box % 4) == 0 && (FrontOff::rbox % 4) == 0, "k_front alignment");
__host__ __device__ inline size_t front_lds_bytes() { return (size_t)FrontOff::end * 4; }
// The same tables WITHOUT the points (k_front_cdl, round 6): phases C and D read the ring-major records from B.ring_pts —
// every phase but the rare general edge search reads them as a stream — and the image is the tables alone, with the merge's
// image (which borrows the points in FrontOff) laid over the tables that are dead by then: 37 KB instead of 79, four workgroups
// a CU's LDS instead of two.  Same names, same overlays (pairs / queues over croot..ctmp, the replay's scratch over seg_box).
struct FrontLeanOff {
  static constexpr uint32_t px = 0, py = 0, pz = 0, sidx = 0;              // (no points: never dereferenced in the lean instance)
  static constexpr uint32_t merge = 0;                                     // merge_body's image: [0, merge_end) — over what follows up to croot
  static constexpr uint32_t seg_box = 0;                                   // [SEGS][4]
  static constexpr uint32_t smask = seg_box + 4 * FX_FRONT_SEGS;           // [BLOCKS] uint64
  static constexpr uint32_t gmask = smask + 2 * FX_FRONT_BLOCKS;
  static constexpr uint32_t run_base = gmask + 2 * FX_FRONT_BLOCKS;
  static constexpr uint32_t seg_base = run_base + FX_FRONT_BLOCKS;
  static constexpr uint32_t seg_start = seg_base + FX_FRONT_BLOCKS;        // uint16 [SEGS + 1]
  static constexpr uint32_t rseg = a4(seg_start + (FX_FRONT_SEGS + 2) / 2); // uint16 [RUNS + 1]
  static constexpr uint32_t rparent = a4(rseg + (FX_FRONT_RUNS + 2) / 2);
  static constexpr uint32_t rsize = rparent + FX_FRONT_RUNS;
  static constexpr uint32_t roff = rsize + FX_FRONT_RUNS;                  // uint16 [RUNS]
  static constexpr uint32_t merge_end = a4((uint32_t)merge_words(FX_FRONT_MERGE, FX_FRONT_MERGE, FX_FRONT_RMAX, true));
  static constexpr uint32_t croot_min = a4(roff + FX_FRONT_RUNS / 2);
  static constexpr uint32_t croot = (croot_min + FX_FRONT_RUNS / 2 + 4 > merge_end ? croot_min : merge_end - (FX_FRONT_RUNS / 2 + 4));  // uint16 [RUNS + 8]: the last of the tables the merge's image lies over
  static constexpr uint32_t crec = croot + FX_FRONT_RUNS / 2 + 4;          // ---- live through the merge from here on
  static constexpr uint32_t ctmp = crec + FX_FRONT_RUNS + 4;
  static constexpr uint32_t rbox = a4(ctmp + FX_FRONT_RUNS + 4);           // [RUNS][4] run boxes, cluster boxes, centroids
  static constexpr uint32_t s_w = rbox + 4 * FX_FRONT_RUNS;
  static constexpr uint32_t r_off = s_w + 64;
  static constexpr uint32_t r_cnt = r_off + FX_FRONT_RMAX + 4;
  static constexpr uint32_t r_run0 = r_cnt + FX_FRONT_RMAX;
  static constexpr uint32_t r_cb = r_run0 + FX_FRONT_RMAX + 4;
  static constexpr uint32_t end = r_cb + FX_FRONT_RMAX + 4;
  static constexpr uint32_t rsm = croot, pairs = croot, queues = pairs + FX_FRONT_PAIRS, q_end = queues + FX_FRONT_NW * FX_FRONT_QUEUE;
};
static_assert(FrontLeanOff::crec >= FrontLeanOff::merge_end, "k_front_cdl: the merge's image ends before the tables it needs");
static_assert(FrontLeanOff::q_end <= FrontLeanOff::rbox, "k_front_cdl overlays");
static_assert((FrontLeanOff::smask % 2) == 0 && (FrontLeanOff::rsm % 2) == 0 && (FrontLeanOff::seg_box % 4) == 0 && (FrontLeanOff::rbox % 4) == 0, "k_front_cdl alignment");
static_assert(FrontLeanOff::end * 4 <= 40 * 1024, "four workgroups of k_front_cdl a CU");
__host__ __device__ inline size_t front_lean_lds_bytes() { return (size_t)FrontLeanOff::end * 4; }

__device__ __forceinline__ unsigned long long le_mask64(uint32_t lane) { return lane == 63u ? ~0ull : ((2ull << lane) - 1ull); }

// Phases C and D of the fused front kernel — getCylinderSegments for all rings of the scan at once, then the secondary merge —
// on the LDS image FrontOff describes: the ring-major points in px / py / pz, r_off / r_cnt set, n entries in all.  Shared by
// k_front (the scan never left the workgroup; RM = false: a member's record is ~cloud's, through the survivor index sidx) and
// k_front_cd (RM = true: the ring-major records k_front_ab wrote to B.ring_pts, entry i = record i).  Ends the workgroup's
// work for the scan: results written, or the scan handed to k_front_redo.
template <bool RM, class O = FrontOff, uint32_t NT_ = FX_FRONT_T>
__device__ __forceinline__ void front_cluster_merge(const FxDevParams &P, const FxBuffers &B, uint32_t scan, uint32_t *smem, uint32_t n,
                                                    uint32_t clk_slot, uint32_t merge_cap) {
  constexpr bool LEAN = std::is_same<O, FrontLeanOff>::value;  // the points stay in B.ring_pts (k_front_cdl)
  static_assert(!LEAN || RM, "the lean image reads ring-major records");
  constexpr uint32_t NT = NT_, NW = NT_ / 64, RUNS = FX_FRONT_RUNS, SEGS = FX_FRONT_SEGS;
  static_assert(NW <= FX_FRONT_NW, "the image's per-wavefront areas are sized for FX_FRONT_NW");
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t R = (uint32_t)P.n_rings;
  float *px = reinterpret_cast<float *>(smem + O::px), *py = reinterpret_cast<float *>(smem + O::py), *pz = reinterpret_cast<float *>(smem + O::pz);
  const uint16_t *sidx = reinterpret_cast<const uint16_t *>(smem + O::sidx);
  uint32_t *s_w = smem + O::s_w, *r_off = smem + O::r_off, *r_cnt = smem + O::r_cnt, *r_run0 = smem + O::r_run0, *r_cb = smem + O::r_cb;
  unsigned long long *smask = reinterpret_cast<unsigned long long *>(smem + O::smask), *gmask = reinterpret_cast<unsigned long long *>(smem + O::gmask);
  uint32_t *run_base = smem + O::run_base, *seg_base = smem + O::seg_base;
  uint16_t *seg_start = reinterpret_cast<uint16_t *>(smem + O::seg_start), *rseg = reinterpret_cast<uint16_t *>(smem + O::rseg),
           *roff = reinterpret_cast<uint16_t *>(smem + O::roff);
  uint32_t *seg_box = smem + O::seg_box, *rbox = smem + O::rbox, *rparent = smem + O::rparent, *rsize = smem + O::rsize;
  uint16_t *croot = reinterpret_cast<uint16_t *>(smem + O::croot);
  uint32_t *crec = smem + O::crec, *ctmp = smem + O::ctmp;
  FX_STAMP_INIT(B.stamps);
  auto stamp_end = [&]() {
    if (tid == 0) atomicMax(&B.clk[2 * clk_slot + 1], (unsigned long long)wall_clock64());
  };
  auto no_keypoints = [&]() {  // ref: node.cpp:209-210, 263-264
    if (tid == 0) {
      B.n_cand[scan] = 0u;
      B.n_kp[scan] = 0u;
      B.n_kpc[scan] = 0u;
    }
  };
  auto redo = [&]() {  // (workgroup-uniform) the general kernels' bodies take the scan from ~cloud: k_front_redo
    if (tid == 0) B.redo[atomicAdd(&B.counters[FX_CNT_REDO], 1u)] = scan;
    stamp_end();
  };
  // a ring-major entry's record (rotated x y z, elevation)
  const float4 *src = RM ? B.ring_pts + (size_t)scan * P.ring_slot_cap : B.filt + (size_t)scan * P.max_points;
  auto member = [&](uint32_t i) -> float4 { return RM ? src[i] : src[sidx[i]]; };
  // a ring-major entry's point: LDS, or — the lean image — its record in HBM (one 16-byte load)
  auto ld = [&](uint32_t i) -> float4 { return LEAN ? src[i] : make_float4(px[i], py[i], pz[i], 0.0f); };
  (void)no_keypoints;
  (void)r_cnt;
  (void)NW;
  // ---------------------------------------------------------------- C: getCylinderSegments, all rings (ref: node.cpp:261-327)
  FX_STAMP(2);
  const float r2 = P.r2_cluster;
  const uint32_t nblk = (n + 63u) >> 6;
  auto run_at = [&](uint32_t i) { return run_base[i >> 6] + (uint32_t)__popcll(smask[i >> 6] & le_mask64(i & 63u)) - 1u; };
  auto run_first = [&](uint32_t r) { return (uint32_t)seg_start[rseg[r]]; };
  {
    unsigned long long *rsm = reinterpret_cast<unsigned long long *>(smem + O::rsm);  // bit i: entry i is the first of its ring
    for (uint32_t t = tid; t < 4 * SEGS; t += NT) seg_box[t] = (t & 1u) ? f2ord(-INFINITY) : f2ord(INFINITY);
    for (uint32_t t = tid; t < 4 * RUNS; t += NT) rbox[t] = (t & 1u) ? f2ord(-INFINITY) : f2ord(INFINITY);
    if (tid < FX_FRONT_BLOCKS) rsm[tid] = 0ull;
    __syncthreads();
    if (tid < R && r_cnt[tid]) atomicOr(&rsm[r_off[tid] >> 6], 1ull << (r_off[tid] & 63u));
    __syncthreads();
    // ---- run labelling: an entry starts a run when it starts its ring or is not closer than the tolerance to its predecessor
    for (uint32_t b0 = 0; b0 < n; b0 += NT) {
      const uint32_t i = b0 + tid, k = i >> 6;
      const bool in = i < n;
      bool start = false;
      if (in) {
        start = (rsm[k] >> lane) & 1ull;
        if (!start) {
          const float4 a = ld(i), b = ld(i - 1u);
          start = !(dist2(a.x, a.y, a.z, b.x, b.y, b.z) < r2);  // (entry 0 starts a ring)
        }
      }
      const unsigned long long m = __ballot(start), g = __ballot(in && (start || (i & 15u) == 0u));
      if (lane == 0 && b0 + wave * 64u < n) smask[k] = m, gmask[k] = g;
    }
    __syncthreads();
    if (tid < 64) {
      const uint32_t rc = tid < nblk ? (uint32_t)__popcll(smask[tid]) : 0u, sc = tid < nblk ? (uint32_t)__popcll(gmask[tid]) : 0u;
      uint32_t ir = rc, is = sc;
      ir = wave_incl_scan(ir), is = wave_incl_scan(is);
      if (tid < FX_FRONT_BLOCKS) run_base[tid] = ir - rc, seg_base[tid] = is - sc;
      if (tid == 63) s_w[20] = ir, s_w[21] = is;
    }
    __syncthreads();
  }
  const uint32_t n_runs = s_w[20], n_segs = s_w[21];
  FX_STAMP(3);
  if (n_runs > RUNS) {  // (then the segments fit too: at most one per run and one per 16 entries)
    redo();
    return;
  }
  for (uint32_t b0 = 0; b0 < n; b0 += NT) {
    const uint32_t i = b0 + tid;
    if (i < n) {
      const unsigned long long sm = smask[i >> 6], gm = gmask[i >> 6], le = le_mask64(lane);
      const uint32_t r = run_base[i >> 6] + (uint32_t)__popcll(sm & le) - 1u, sg = seg_base[i >> 6] + (uint32_t)__popcll(gm & le) - 1u;
      if ((gm >> lane) & 1ull) seg_start[sg] = (uint16_t)i;
      if ((sm >> lane) & 1ull) rseg[r] = (uint16_t)sg;
      const float4 pi = ld(i);
      const uint32_t ox = f2ord(pi.x), oy = f2ord(pi.y);
      atomicMin(&seg_box[4 * sg + 0], ox);
      atomicMax(&seg_box[4 * sg + 1], ox);
      atomicMin(&seg_box[4 * sg + 2], oy);
      atomicMax(&seg_box[4 * sg + 3], oy);
    }
  }
  if (tid == 0) {
    seg_start[n_segs] = (uint16_t)n;
    rseg[n_runs] = (uint16_t)n_segs;
    s_w[16] = 0u;  // near run pairs
    s_w[17] = 0u;  // a ring with more candidates than max_ring_candidates
  }
  if (tid <= R) r_run0[tid] = (tid < R && r_off[tid] < n) ? run_at(r_off[tid]) : n_runs;  // (an empty ring: the next ring's first run)
  __syncthreads();
  // ---- segment boxes -> floats, folded into the run boxes; union-find over runs
  for (uint32_t sg = tid; sg < n_segs; sg += NT) {
    const uint32_t r = run_at(seg_start[sg]);
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      const uint32_t o = seg_box[4 * sg + k];
      seg_box[4 * sg + k] = __float_as_uint(ord2f(o));
      if (k & 1)
        atomicMax(&rbox[4 * r + k], o);
      else
        atomicMin(&rbox[4 * r + k], o);
    }
  }
  __syncthreads();
  for (uint32_t r = tid; r < n_runs; r += NT) {
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) rbox[4 * r + k] = __float_as_uint(ord2f(rbox[4 * r + k]));
    rparent[r] = r;
    rsize[r] = 0u;
  }
  __syncthreads();
  const float4 *sbox4 = reinterpret_cast<const float4 *>(seg_box);
  FX_STAMP(4);
  float4 *rbox4 = reinterpret_cast<float4 *>(rbox);
  const float r2_pad = r2 * 1.001f;  // box distances are lower bounds; pad them against fp32 rounding
  {
    // ---- near run pairs of the same ring.  A run's work is the later runs of its ring — two ground rings of ~90 runs hold
    //      most of a scan's (a, b) tests —, so the tests are laid end to end (prefix of the runs' work in rsize, idle until
    //      the sizes are summed) and dealt to the lanes in equal contiguous shares.
    uint32_t *pairs = smem + O::pairs;
    uint32_t W = 0;
    for (uint32_t b0 = 0; b0 < n_runs; b0 += NT) {
      const uint32_t a = b0 + tid;
      uint32_t w_a = 0;
      if (a < n_runs) {
        const uint32_t end_a = r_run0[prefix_owner(r_run0, R, a) + 1u];  // first run of the next ring
        roff[a] = (uint16_t)end_a;
        w_a = end_a - a - 1u;
      }
      uint32_t tot;
      const uint32_t ex = block_excl_scan<NT>(w_a, s_w, tot);
      if (a < n_runs) rsize[a] = W + ex;
      W += tot;
    }
    __syncthreads();
    {
      const uint32_t q = (W + NT - 1u) / NT;
      uint32_t flat = tid * q;
      const uint32_t flat_end = min(flat + q, W);
      if (flat < flat_end) {
        uint32_t lo = 0, hi = n_runs;  // the run of test `flat`: the last one whose tests start at or before it (it has tests: see the prefix)
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (rsize[mid] <= flat)
            lo = mid;
          else
            hi = mid;
        }
        uint32_t a = lo, end_a = roff[a], b = a + 1u + (flat - rsize[a]);
        float4 ba = rbox4[a];
        for (; flat < flat_end; ++flat) {
          const float4 bx = rbox4[b];
          const float dx = fmaxf(fmaxf(bx.x - ba.y, ba.x - bx.y), 0.0f);
          const float dy = fmaxf(fmaxf(bx.z - ba.w, ba.z - bx.w), 0.0f);
          if (!(dx * dx + dy * dy > r2_pad)) {  // (rare)
            const uint32_t slot = atomicAdd(&s_w[16], 1u);
            if (slot < FX_FRONT_PAIRS) pairs[slot] = (a << 16) | b;
          }
          if (++b == end_a && flat + 1u < flat_end) {  // the next run that has later runs in its ring
            do {
              ++a;
              end_a = roff[a];
            } while (end_a == a + 1u);
            ba = rbox4[a];
            b = a + 1u;
          }
        }
      }
    }
    __syncthreads();
    const uint32_t n_rp = s_w[16];
    FX_STAMP(5);
    if (tid == 0) {
      FX_COUNT(20, n_rp);
      FX_COUNT(21, n_runs);
      FX_COUNT(22, n);
      FX_COUNT(23, 1);
    }
    if (n_rp > FX_FRONT_PAIRS) {  // runs all over each other (unordered input)
      redo();
      return;
    }
    // ---- cross-run edges: the points of the earlier run of every near pair against the later run's box; near
    //      (point, run) items are parked and drained with all lanes busy: the run's segments' boxes, then their points.
    //      Every pair that could be an edge is examined: exact for any input order.
    uint32_t *wq = smem + O::queues + wave * FX_FRONT_QUEUE;
    uint32_t wq_n = 0;
    auto drain = [&]() {
      wave_sync_lds();
      for (uint32_t t = lane; t < wq_n; t += 64) {
        const uint32_t item = wq[t];
        const uint32_t i = item & 0xfffu, b = (item >> 12) & 0x3ffu, a = item >> 22;
        if (uf_find(rparent, a) == uf_find(rparent, b)) continue;  // already one component
        const float4 pq = ld(i);
        const float qx = pq.x, qy = pq.y, qz = pq.z;
        bool linked = false;
        for (uint32_t sg = rseg[b]; sg < rseg[b + 1] && !linked; ++sg) {
          const float4 sb = sbox4[sg];
          const float dx = fmaxf(fmaxf(sb.x - qx, qx - sb.y), 0.0f);
          const float dy = fmaxf(fmaxf(sb.z - qy, qy - sb.w), 0.0f);
          if (dx * dx + dy * dy > r2_pad) continue;
          const uint32_t j1 = seg_start[sg + 1];
          for (uint32_t j = seg_start[sg]; j < j1 && !linked; j += 4) {
            float jx[4], jy[4], jz[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
              const uint32_t ju = min(j + u, j1 - 1u);
              const float4 pj = ld(ju);
              jx[u] = pj.x, jy[u] = pj.y, jz[u] = pj.z;
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) linked |= dist2(qx, qy, qz, jx[u], jy[u], jz[u]) < r2;
          }
          if (linked) uf_union(rparent, b, a);  // the two runs are one component now; more edges add nothing
        }
      }
      wave_sync_lds();
      wq_n = 0;
    };
    // In an azimuth-ordered ring the edge between two near runs, when there is one, is mostly between the end of the earlier
    // and the start of the later (an arc behind a pole's shadow): found at once, the pair needs nothing else.  That test a pair
    // a LANE (it is a chain of dependent LDS reads: a pair a wavefront spent most of this step waiting on them 64 lanes wide);
    // what it does not settle, a pair a wavefront.
    for (uint32_t t = tid; t < n_rp; t += NT) {
      const uint32_t pr = pairs[t], a = pr >> 16, b = pr & 0xffffu;
      const uint32_t ia = run_first(a + 1u) - 1u, ib = run_first(b);
      const float4 pb = ld(ib), pa = ld(ia);
      if (dist2(pb.x, pb.y, pb.z, pa.x, pa.y, pa.z) < r2) {
        uf_union(rparent, b, a);
        pairs[t] = FX_NONE;
      }
    }
    __syncthreads();
    for (uint32_t t = wave; t < n_rp; t += NW) {
      const uint32_t pr = pairs[t], a = pr >> 16, b = pr & 0xffffu;
      if (pr == FX_NONE) continue;  // (wave-uniform)
      const uint32_t i_end = run_first(a + 1u);
      const float4 bb = rbox4[b];
      for (uint32_t i0 = run_first(a); i0 < i_end; i0 += 64) {
        const uint32_t i = i0 + lane;
        bool ok = false;
        if (i < i_end) {
          const float4 pq = ld(i);
          const float qx = pq.x, qy = pq.y;
          const float dx = fmaxf(fmaxf(bb.x - qx, qx - bb.y), 0.0f);
          const float dy = fmaxf(fmaxf(bb.z - qy, qy - bb.w), 0.0f);
          ok = !(dx * dx + dy * dy > r2_pad);
        }
        const unsigned long long mk = __ballot(ok);
        if (mk) {
          if (ok) wq[wq_n + lanes_below(mk)] = i | (b << 12) | (a << 22);
          wq_n += (uint32_t)__popcll(mk);
          if (wq_n > FX_FRONT_QUEUE - 64) drain();
        }
      }
    }
    if (wq_n) drain();
    __syncthreads();
  }
  // ---- roots (read-only finds, then the owners overwrite), sizes = sums of run lengths
  FX_STAMP(6);
  for (uint32_t r = tid; r < n_runs; r += NT) {
    ctmp[r] = uf_find_ro(rparent, r);
    rsize[r] = 0u;  // (held the near-pair work prefix)
  }
  __syncthreads();
  for (uint32_t r = tid; r < n_runs; r += NT) {
    const uint32_t root = ctmp[r];
    rparent[r] = root;
    atomicAdd(&rsize[root], run_first(r + 1u) - run_first(r));
  }
  __syncthreads();
  // ---- size-admissible clusters in discovery order (ascending root run: ring by ring, and inside a ring by smallest
  FX_STAMP(7);
  //      point index: SURVEY.md A.5); crec = size << 16 | discovery ordinal (only the size is ever compared)
  uint32_t n_c = 0;
  for (uint32_t b0 = 0; b0 < n_runs; b0 += NT) {
    const uint32_t r = b0 + tid;
    bool acc = false;
    uint32_t sz = 0;
    if (r < n_runs && rparent[r] == r) {
      sz = rsize[r];
      acc = sz >= P.min_count && sz <= P.max_count;
    }
    uint32_t tot;
    const uint32_t rank = block_rank<NT>(acc, s_w, tot);
    if (r < n_runs) roff[r] = (uint16_t)(n_c + rank);  // clusters before run r
    if (acc) {
      croot[n_c + rank] = (uint16_t)r;
      crec[n_c + rank] = (sz << 16) | (n_c + rank);
    }
    n_c += tot;
  }
  __syncthreads();
  if (tid <= R) r_cb[tid] = (tid < R && r_run0[tid] < n_runs) ? (uint32_t)roff[r_run0[tid]] : n_c;
  __syncthreads();
  FX_STAMP(8);
  // ---- PCL's cluster order, ring by ring: std::sort(rbegin, rend, bySize) replayed (csrc/fx_sort_replay.h) — its
  //      partition phase (only above 16 clusters) a ring per wavefront, its insertion phase — a stable sort — as a ranking
  {
    int *stk = reinterpret_cast<int *>(seg_box + wave * FX_FRONT_SORTW);
    uint16_t *pos = reinterpret_cast<uint16_t *>(stk + FX_SORT_STACK_WORDS);
    for (uint32_t ring = wave; ring < R; ring += NW) {
      const uint32_t lo = r_cb[ring], nc = r_cb[ring + 1] - lo;
      if (nc <= FX_SORT_THRESHOLD) continue;
      if (nc <= 64) {
        sort_partition_wave<1>(crec + lo, (int)nc, stk, pos, pos + nc);
      } else if (nc <= 128) {
        sort_partition_wave<2>(crec + lo, (int)nc, stk, pos, pos + nc);
      } else if (nc <= 192) {
        sort_partition_wave<3>(crec + lo, (int)nc, stk, pos, pos + nc);
      } else if (lane == 0) {
        fx_sort_detail::RevView v{crec + lo, (int)nc};
        fx_sort_partition_phase(v, (int)nc, stk);
      }
    }
  }
  __syncthreads();
  FX_STAMP(9);
  for (uint32_t g = tid; g < n_c; g += NT) {
    const uint32_t ring = prefix_owner(r_cb, R, g), lo = r_cb[ring], hi = r_cb[ring + 1];
    const uint32_t rec = crec[g], sz = rec >> 16;
    uint32_t p = lo;
    for (uint32_t d = lo; d < hi; ++d) {
      const uint32_t sd = crec[d] >> 16;
      p += (sd > sz || (sd == sz && d < g)) ? 1u : 0u;
    }
    ctmp[p] = rec;
  }
  __syncthreads();
  FX_STAMP(10);
  // ---- cluster boxes = folds of run boxes (min / max are exact whatever the order): roots to ordered uints, the others in
  for (uint32_t r = tid; r < n_runs; r += NT) {
    if (rparent[r] != r) continue;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) rbox[4 * r + k] = f2ord(__uint_as_float(rbox[4 * r + k]));
  }
  for (uint32_t g = tid; g < n_c; g += NT) crec[g] = ctmp[g];
  __syncthreads();
  for (uint32_t r = tid; r < n_runs; r += NT) {
    const uint32_t root = rparent[r];
    if (root == r) continue;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      const uint32_t o = f2ord(__uint_as_float(rbox[4 * r + k]));
      if (k & 1)
        atomicMax(&rbox[4 * root + k], o);
      else
        atomicMin(&rbox[4 * root + k], o);
    }
  }
  __syncthreads();
  // ---- diameter gate per cluster, in PCL's cluster order (ref: node.cpp:289-290, 314-316); crec becomes size << 16 | root run
  for (uint32_t g = tid; g < n_c; g += NT) {
    const uint32_t rec = crec[g], root = croot[rec & 0xffffu];
    const double minx = fminf(1000.0f, ord2f(rbox[4 * root + 0])), maxx = fmaxf(-1000.0f, ord2f(rbox[4 * root + 1]));
    const double miny = fminf(1000.0f, ord2f(rbox[4 * root + 2])), maxy = fmaxf(-1000.0f, ord2f(rbox[4 * root + 3]));
    const double ddx = maxx - minx, ddy = maxy - miny;
    ctmp[g] = (sqrt(ddx * ddx + ddy * ddy) < P.gate_diameter) ? 1u : 0u;
    rsize[root] |= (g + 1u) << 16;  // position in PCL's order (all rings), next to the size
  }
  __syncthreads();  // (every reader of croot's ordinals is done: crec may take the roots)
  for (uint32_t g = tid; g < n_c; g += NT) {
    const uint32_t rec = crec[g];
    crec[g] = (rec & 0xffff0000u) | croot[rec & 0xffffu];
  }
  __syncthreads();
  FX_STAMP(11);
  // ---- centroid of the clusters that pass: fp64 sums in ascending member order = the cluster's runs in run order,
  //      each run's points in turn (ref: node.cpp:293-297, 317-320); the intensity is the lowest-index member's elevation
  for (uint32_t g = tid; g < n_c; g += NT) {
    if (ctmp[g] == 0u) continue;
    const uint32_t rec = crec[g], sz = rec >> 16, root = rec & 0xffffu;
    const float el = member(run_first(root)).w;
    const uint32_t r_end = r_run0[prefix_owner(r_cb, R, g) + 1u];
    double sumx = 0.0, sumy = 0.0, sumz = 0.0;
    uint32_t cnt = 0;
    for (uint32_t r = root; r < r_end && cnt < sz; ++r) {
      if (rparent[r] != root) continue;
      roff[r] = (uint16_t)cnt;
      const uint32_t i0 = run_first(r), i1 = run_first(r + 1u);
      for (uint32_t i = i0; i < i1; i += 4) {
        float x[4], y[4], z[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
          const uint32_t iu = min(i + u, i1 - 1u);
          const float4 pu = ld(iu);
          x[u] = pu.x, y[u] = pu.y, z[u] = pu.z;
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
          if (i + u >= i1) continue;
          sumx += (double)x[u];
          sumy += (double)y[u];
          sumz += (double)z[u];
        }
      }
      cnt += i1 - i0;
    }
    rbox4[root] = make_float4((float)(sumx / (double)sz), (float)(sumy / (double)sz), (float)(sumz / (double)sz), el);
  }
  __syncthreads();
  FX_STAMP(12);
  // ---- ordinals in keypoints_full (ring order, then PCL's order: ref node.cpp:205) and offsets of the members in
  //      keypoint_cloud (:206, 323) of the clusters that pass
  uint32_t C = 0, n_mem = 0;
  uint16_t *ckoff = croot;
  for (uint32_t b0 = 0; b0 < n_c; b0 += NT) {
    const uint32_t g = b0 + tid;
    const bool pass = g < n_c && ctmp[g] != 0u;
    const uint32_t sz = pass ? (crec[g] >> 16) : 0u;
    uint32_t tot_p, tot_m;
    const uint32_t rank = block_rank<NT>(pass, s_w, tot_p);
    const uint32_t koff = block_excl_scan<NT>(sz, s_w, tot_m);
    if (g < n_c) {
      ctmp[g] = (C + rank) | (pass ? 0x80000000u : 0u);
      ckoff[g] = (uint16_t)(n_mem + koff);  // (at most FX_FRONT_CAP members)
    }
    C += tot_p;
    n_mem += tot_m;
  }
  if (tid == 0) ctmp[n_c] = C;
  __syncthreads();
  if (tid < R && (ctmp[r_cb[tid + 1]] & 0x7fffffffu) - (ctmp[r_cb[tid]] & 0x7fffffffu) > P.max_ring_cands) s_w[17] = 1u;
  __syncthreads();
  if (s_w[17] || C > merge_cap || n_mem > P.max_kpc) {  // (the general kernels flag what they have to)
    redo();
    return;
  }
  FX_STAMP(13);
  // ---- keypoint_cloud in its final order (every entry: its run, its cluster, its place)
  {
    float4 *kpc = B.kpc + (size_t)scan * P.max_kpc;
    uint32_t *kpc_c = B.kpc_cand + (size_t)scan * P.max_kpc;
    for (uint32_t i = tid; i < n; i += NT) {
      const uint32_t r = run_at(i), g1 = rsize[rparent[r]] >> 16;
      if (g1 == 0u) continue;  // not a size-admissible cluster
      const uint32_t w = ctmp[g1 - 1u];
      if (!(w >> 31)) continue;
      const uint32_t dst = ckoff[g1 - 1u] + roff[r] + (i - run_first(r));
      kpc[dst] = member(i);
      kpc_c[dst] = w & 0x7fffffffu;
    }
    if (tid == 0) B.n_kpc[scan] = n_mem;
  }
  FX_STAMP(14);
#if defined(FX_FRONT_STOP) && FX_FRONT_STOP == 3
  no_keypoints();
  stamp_end();
  return;
#endif
  // ---------------------------------------------------------------- D: secondary merge (ref: node.cpp:209-257)
  if (LEAN) __syncthreads();  // (the merge's image lies over the run tables the keypoint_cloud loop above still reads: in FrontOff it lies over the points)
  const FrontCands FC{crec, ctmp, rbox4, n_c, C};
  merge_body<NT, true, true>(P, B, scan, merge_cap, merge_cap, LEAN ? smem + FrontLeanOff::merge : smem, true, &FC);
  FX_STAMP(15);
  stamp_end();
}


extern "C" __global__ __launch_bounds__(FX_FRONT_T, FX_PREP_OCC) void k_front(FxDevParams P, FxBuffers B, float near_margin, float el0, float inv_step,
                                                                           uint32_t clk_slot, uint32_t merge_cap, uint32_t force_redo) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  using O = FrontOff;
  constexpr uint32_t NT = FX_FRONT_T, NW = FX_FRONT_NW, CAP = FX_FRONT_CAP, RUNS = FX_FRONT_RUNS, SEGS = FX_FRONT_SEGS;
  static_assert(merge_words(FX_FRONT_MERGE, FX_FRONT_MERGE, FX_FRONT_RMAX, true) <= 3 * FX_FRONT_CAP, "the merge's image borrows the points");
  const uint32_t scan = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t R = (uint32_t)P.n_rings;
  float *px = reinterpret_cast<float *>(smem + O::px), *py = reinterpret_cast<float *>(smem + O::py), *pz = reinterpret_cast<float *>(smem + O::pz);
  uint16_t *sidx = reinterpret_cast<uint16_t *>(smem + O::sidx);
  uint32_t *s_w = smem + O::s_w, *r_off = smem + O::r_off, *r_cnt = smem + O::r_cnt, *r_run0 = smem + O::r_run0, *r_cb = smem + O::r_cb;
  unsigned long long *smask = reinterpret_cast<unsigned long long *>(smem + O::smask), *gmask = reinterpret_cast<unsigned long long *>(smem + O::gmask);
  uint32_t *run_base = smem + O::run_base, *seg_base = smem + O::seg_base;
  uint16_t *seg_start = reinterpret_cast<uint16_t *>(smem + O::seg_start), *rseg = reinterpret_cast<uint16_t *>(smem + O::rseg),
           *roff = reinterpret_cast<uint16_t *>(smem + O::roff);
  uint32_t *seg_box = smem + O::seg_box, *rbox = smem + O::rbox, *rparent = smem + O::rparent, *rsize = smem + O::rsize;
  uint16_t *croot = reinterpret_cast<uint16_t *>(smem + O::croot);
  uint32_t *crec = smem + O::crec, *ctmp = smem + O::ctmp;
  {  // the support-list counters of the batch are cleared here, a slice per scan (k_gather fills them)
    const uint32_t per = (P.max_total_kp + gridDim.x - 1) / gridDim.x;
    const uint32_t z0 = scan * per, z1 = min(z0 + per, P.max_total_kp);
    for (uint32_t t = z0 + tid; t < z1; t += NT) B.s_cnt[t] = 0u;
    if (tid == 0) B.ovf_cnt[scan] = 0u;  // entries in the scan's overflow region (k_gather)
  }
  const FxScanMeta M = B.meta[scan];
  FX_STAMP_INIT(B.stamps);
  if (tid == 0) atomicMin(&B.clk[2 * clk_slot], (unsigned long long)wall_clock64());
  auto stamp_end = [&]() {
    if (tid == 0) atomicMax(&B.clk[2 * clk_slot + 1], (unsigned long long)wall_clock64());
  };
  auto no_keypoints = [&]() {  // ref: node.cpp:209-210, 263-264
    if (tid == 0) {
      B.n_cand[scan] = 0u;
      B.n_kp[scan] = 0u;
      B.n_kpc[scan] = 0u;
    }
  };
  if (M.n == 0) {  // empty scan: its pointer may be null — nothing is loaded
    if (tid == 0) {
      B.n_filt[scan] = 0u;
      B.flags[scan] = 0u;
    }
    no_keypoints();
    for (uint32_t r = tid; r < R; r += NT) B.ring_cnt[(size_t)scan * R + r] = 0u;
    if (scan == 0 && tid < FX_N_COUNTERS) B.counters[tid] = 0u;
    return;
  }
  // ---------------------------------------------------------------- A: the streaming pass
  float *s_el = reinterpret_cast<float *>(smem + O::el);
  const PrepLds PL{smem + O::cnt, px, r_cnt, reinterpret_cast<double *>(smem + O::atan), reinterpret_cast<float2 *>(smem + O::win), s_el, CAP};
  uint32_t nf = prep_stream<true>(P, B, M, scan, near_margin, el0, inv_step, PL);
  if (nf == FX_NONE) nf = prep_stream<false>(P, B, M, scan, near_margin, el0, inv_step, PL);  // (more survivors than the buffer keeps: once more, recycling it)
  FX_STAMP(1);
  for (uint32_t r = tid; r < R; r += NT) B.ring_cnt[(size_t)scan * R + r] = r_cnt[r];  // (k_front_redo's ring split starts from these)
  if (tid == 0) {
    B.n_filt[scan] = nf;
    B.flags[scan] = 0u;
  }
  if (scan == 0 && tid < FX_N_COUNTERS) B.counters[tid] = 0u;  // work-list counters of the batch (not the one k_front itself adds to)
  uint32_t n = 0, ring_max = 0;
  for (uint32_t r = 0; r < R; ++r) {
    const uint32_t c = r_cnt[r];
    n += c;
    ring_max = max(ring_max, c);
  }
  auto redo = [&]() {  // (workgroup-uniform) the general kernels' bodies take the scan from ~cloud: k_front_redo
    if (tid == 0) B.redo[atomicAdd(&B.counters[FX_CNT_REDO], 1u)] = scan;
    stamp_end();
  };
  if (n > CAP || nf > CAP || ring_max > P.max_ring_points || force_redo) {  // (force_redo: the test build's hook)
    redo();
    return;
  }
  if (n == 0) {
    no_keypoints();
    stamp_end();
    return;
  }
#if defined(FX_FRONT_STOP) && FX_FRONT_STOP == 1  // measurement build: the kernel's phases one at a time (tools/front_phases.sh)
  no_keypoints();
  stamp_end();
  return;
#endif
  // ---------------------------------------------------------------- B: ring split into LDS (ref: node.cpp:195-202)
  // A stable counting sort by ring with three barriers and nothing from HBM: the survivors are still in LDS (un-rotated, in
  // input order: the streaming pass's buffer) with their elevations.  Every wavefront owns a contiguous slice, ranks its
  // points ring by ring (a ballot per ring present in 64 points, running counts in registers), the counts of the wavefronts
  // before it make the slice's base in every ring; every lane then rotates its points (registers), and — once every lane has
  // read its points: the ring-major arrays take the buffer's place — the points go to their places.
  const float4 *f = B.filt + (size_t)scan * P.max_points;
  {
    const float2 *s_win = PL.win;
    uint32_t *cw = smem + O::cw, *cbase = cw + NW * FX_FRONT_RMAX;  // [NW][R] points of the wavefront's slice per ring; its first place per ring
    if (tid < 64) {
      const uint32_t c = tid < R ? r_cnt[tid] : 0u;
      uint32_t inc = c;
      inc = wave_incl_scan(inc);
      if (tid <= R) r_off[tid] = inc - c;  // (r_off[R] = n)
    }
    FX_STAMP(16);
    constexpr uint32_t kSub = (CAP + 64 * NW - 1) / (64 * NW);   // 64-point pieces of a wavefront's slice at most
    const uint32_t S = ((nf + 64u * NW - 1u) / (64u * NW)) * 64u;  // slice length
    // per point: its first ring (a point is in one ring, or — exactly on a window's edge — in that one and the next) and its
    // place among the wavefront's points of that ring; the second ring's place in a register of its own
    uint32_t ring_a[kSub], place_b[kSub];  // ring_a: ring | 0x100: also in ring + 1 | place in the ring << 16; FX_NONE: in no ring
    uint32_t run_cnt = 0;  // lane r: points of ring r in this wavefront's slice so far (no LDS round trip per ring in the loop below)
#pragma unroll
    for (uint32_t u = 0; u < kSub; ++u) {
      ring_a[u] = FX_NONE, place_b[u] = 0;
      if (u * 64u >= S) continue;  // (workgroup-uniform)
      const uint32_t i = wave * S + u * 64u + lane;
      const float el = i < nf ? s_el[i] : NAN;
      uint32_t mask = 0;
      int r_first = 0;
      if (isfinite(el)) mask = ring_membership(el, s_win, P.n_rings, el0, inv_step, r_first);
      // (windows of neighbouring rings share their edge only: at most two memberships, in consecutive rings)
      const int ra = mask ? r_first + (__ffs((int)mask) - 1) : -1;
      const bool two = (mask & (mask - 1u)) != 0u;
      if (mask) ring_a[u] = (uint32_t)ra | (two ? 0x100u : 0u);
#ifndef FX_SPLIT_BALLOTS
      // Ranks by a PACKED prefix sum: a lane's memberships as a one in the 8-bit field of its ring (four rings a word: at most
      // 64 points a piece, no field overflows), the words prefix-summed across the wavefront (six DPP steps each) — the rank
      // among the piece's points of every ring at once.  (A ballot per ring present: sixteen dependent trips a piece.)
      const uint32_t rb = (uint32_t)ra + 1u;
      const uint32_t one_a = mask ? 1u << (8u * ((uint32_t)ra & 3u)) : 0u, one_b = two ? 1u << (8u * (rb & 3u)) : 0u;
      const uint32_t ka = (uint32_t)ra >> 2, kb = rb >> 2;
      uint32_t mine_a = 0, mine_b = 0, tot = 0;  // the words of this lane's rings; lane r: the word of ring r's totals
#pragma nounroll
      for (uint32_t k = 0; 4u * k < R; ++k) {  // (a word of four rings a trip)
        uint32_t x = (ka == k ? one_a : 0u) + (kb == k ? one_b : 0u);
        x = wave_incl_scan(x);
        if (ka == k) mine_a = x;
        if (kb == k) mine_b = x;
        const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
        if ((lane >> 2) == k) tot = t;
      }
      // (this lane's points of ring r so far: lane r's run_cnt — fetched by the ring's number)
      const uint32_t before_a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)ra & 63u) << 2), (int)run_cnt);
      if (mask) ring_a[u] |= (before_a + ((mine_a >> (8u * ((uint32_t)ra & 3u))) & 0xffu) - 1u) << 16;
      if (__ballot(two)) {  // (wave-uniform; a point exactly on a window's edge: rare)
        const uint32_t before_b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((rb & 63u) << 2), (int)run_cnt);
        if (two) place_b[u] = before_b + ((mine_b >> (8u * (rb & 3u))) & 0xffu) - 1u;
      }
      if (lane < R) run_cnt += (tot >> (8u * (lane & 3u))) & 0xffu;
#else
      int lo = mask ? ra : 0x7fffffff, hi = mask ? ra + (two ? 2 : 1) : -1;  // rings present in these 64 points: [lo, hi)
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        lo = min(lo, __shfl_xor(lo, d, 64));
        hi = max(hi, __shfl_xor(hi, d, 64));
      }
      lo = __builtin_amdgcn_readfirstlane(max(lo, 0));
      hi = __builtin_amdgcn_readfirstlane(min(hi, (int)R));
      for (int r = lo; r < hi; ++r) {
        const bool in_a = r == ra, in_b = two && r == ra + 1;
        const unsigned long long m = __ballot(in_a || in_b);
        if (!m) continue;
        const uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)run_cnt, r) + lanes_below(m);
        if (in_a) ring_a[u] |= at << 16;
        if (in_b) place_b[u] = at;
        if ((int)lane == r) run_cnt += (uint32_t)__popcll(m);
      }
#endif
    }
    if (lane < R) cw[wave * R + lane] = run_cnt;
    FX_STAMP(17);
    // this lane's points, rotated as the sweep rotated them for ~cloud (pcl::transformPointCloud's order, ref: node.cpp:161-166)
    float rx[kSub], ry[kSub], rz[kSub];
#pragma unroll
    for (uint32_t u = 0; u < kSub; ++u) {
      const uint32_t i = min(wave * S + u * 64u + lane, nf - 1u);
      const float x = px[3 * i], y = px[3 * i + 1], z = px[3 * i + 2];  // (the buffer: x y z per survivor, where px / py / pz will be)
      rx[u] = ((M.R[0] * x + M.R[1] * y) + M.R[2] * z) + 0.0f;
      ry[u] = ((M.R[3] * x + M.R[4] * y) + M.R[5] * z) + 0.0f;
      rz[u] = ((M.R[6] * x + M.R[7] * y) + M.R[8] * z) + 0.0f;
    }
    FX_STAMP(18);
    __syncthreads();  // (every count is in; every lane has read its points)
    for (uint32_t t = tid; t < NW * R; t += NT) {
      const uint32_t w = t / R, r = t - w * R;
      uint32_t before = r_off[r];
      for (uint32_t x = 0; x < w; ++x) before += cw[x * R + r];
      cbase[t] = before;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < kSub; ++u) {
      if (ring_a[u] == FX_NONE) continue;
      const uint32_t ra = ring_a[u] & 0xffu;
      uint32_t pos = cbase[wave * R + ra] + (ring_a[u] >> 16);
      px[pos] = rx[u], py[pos] = ry[u], pz[pos] = rz[u];
      sidx[pos] = (uint16_t)(wave * S + u * 64u + lane);
      if (ring_a[u] & 0x100u) {
        pos = cbase[wave * R + ra + 1u] + place_b[u];
        px[pos] = rx[u], py[pos] = ry[u], pz[pos] = rz[u];
        sidx[pos] = (uint16_t)(wave * S + u * 64u + lane);
      }
    }
    wg_global_sync();  // (the ring-major points are in; ~cloud, written by the sweeps, is read below: cluster intensities, member copies)
    FX_STAMP(19);
  }
#if defined(FX_FRONT_STOP) && FX_FRONT_STOP == 2
  no_keypoints();
  stamp_end();
  return;
#endif
  front_cluster_merge<false>(P, B, scan, smem, n, clk_slot, merge_cap);
}

// ---- the fused kernel as TWO launches (batches that fill the chip: VERDICT r5 #1).  k_front's two workgroups own a CU — 2 x 79 KB of
// LDS, 16 wavefronts x 128 registers — for the kernel's whole 0.28 ms, and its halves want opposite things: A (+ B) is
// bandwidth-shaped, C + D are latency chains.  k_front_ab: the streaming pass and the ring split in k_prep's register
// budget, the ring-major records (rotated x y z, elevation: ~cloud's record) written to B.ring_pts as k_bucket writes them —
// 41 KB a scan, read back once by k_front_cd from the L2 / infinity cache —; k_front_cd: loads them into the image and runs
// phases C and D (front_cluster_merge<true>: a member's record is its ring-major one, no survivor index).  A scan's state
// between the two: B.front_n[scan] = its ring-major entries, 0 when it has none (k_front_ab wrote the empty results),
// FX_NONE when k_front_ab handed it to k_front_redo itself.
struct FrontAbOff {  // word offsets into k_front_ab's LDS image
  static constexpr uint32_t keep = 0;                                       // [3 CAP] the streaming pass's survivors (un-rotated x y z)
  static constexpr uint32_t el = 3 * FX_FRONT_CAP;                          // [CAP] their elevations
  static constexpr uint32_t atan = 4 * FX_FRONT_CAP;                        // doubles
  static constexpr uint32_t win = atan + a4(2 * (FX_ATAN_N + 1) * (FX_ATAN_DEG + 1));
  static constexpr uint32_t cnt = win + 2 * FX_FRONT_RMAX;
  static constexpr uint32_t r_cnt = cnt + 2 * FX_FRONT_NW;
  static constexpr uint32_t r_off = r_cnt + FX_FRONT_RMAX;
  static constexpr uint32_t cw = r_off + FX_FRONT_RMAX + 4;                 // [2][NW][RMAX]
  static constexpr uint32_t end = cw + 2 * FX_FRONT_NW * FX_FRONT_RMAX;
};
static_assert((FrontAbOff::atan % 2) == 0, "k_front_ab alignment");
__host__ __device__ inline size_t front_ab_lds_bytes() { return (size_t)FrontAbOff::end * 4; }
#ifndef FX_FRONT_AB_OCC
#define FX_FRONT_AB_OCC FX_PREP_OCC
#endif

extern "C" __global__ __launch_bounds__(FX_FRONT_T, FX_FRONT_AB_OCC) void k_front_ab(FxDevParams P, FxBuffers B, float near_margin, float el0, float inv_step,
                                                                                 uint32_t clk_slot, uint32_t force_redo) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  using O = FrontAbOff;
  constexpr uint32_t NT = FX_FRONT_T, NW = FX_FRONT_NW, CAP = FX_FRONT_CAP;
  const uint32_t scan = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t R = (uint32_t)P.n_rings;
  float *keep = reinterpret_cast<float *>(smem + O::keep), *s_el = reinterpret_cast<float *>(smem + O::el);
  uint32_t *r_cnt = smem + O::r_cnt, *r_off = smem + O::r_off;
  {  // the support-list counters of the batch are cleared here, a slice per scan (k_gather fills them)
    const uint32_t per = (P.max_total_kp + gridDim.x - 1) / gridDim.x;
    const uint32_t z0 = scan * per, z1 = min(z0 + per, P.max_total_kp);
    for (uint32_t t = z0 + tid; t < z1; t += NT) B.s_cnt[t] = 0u;
    if (tid == 0) B.ovf_cnt[scan] = 0u;  // entries in the scan's overflow region (k_gather)
  }
  const FxScanMeta M = B.meta[scan];
  if (tid == 0) atomicMin(&B.clk[2 * clk_slot], (unsigned long long)wall_clock64());
  auto stamp_end = [&]() {
    if (tid == 0) atomicMax(&B.clk[2 * clk_slot + 1], (unsigned long long)wall_clock64());
  };
  auto no_keypoints = [&]() {  // ref: node.cpp:209-210, 263-264
    if (tid == 0) {
      B.n_cand[scan] = 0u;
      B.n_kp[scan] = 0u;
      B.n_kpc[scan] = 0u;
      B.front_n[scan] = 0u;
    }
  };
  if (M.n == 0) {  // empty scan: its pointer may be null — nothing is loaded
    if (tid == 0) {
      B.n_filt[scan] = 0u;
      B.flags[scan] = 0u;
    }
    no_keypoints();
    for (uint32_t r = tid; r < R; r += NT) B.ring_cnt[(size_t)scan * R + r] = 0u;
    if (scan == 0 && tid < FX_N_COUNTERS) B.counters[tid] = 0u;
    return;
  }
  // ---------------------------------------------------------------- A: the streaming pass
  const PrepLds PL{smem + O::cnt, keep, r_cnt, reinterpret_cast<double *>(smem + O::atan), reinterpret_cast<float2 *>(smem + O::win), s_el, CAP};
  uint32_t nf = prep_stream<true>(P, B, M, scan, near_margin, el0, inv_step, PL);
  if (nf == FX_NONE) nf = prep_stream<false>(P, B, M, scan, near_margin, el0, inv_step, PL);  // (more survivors than the buffer keeps: once more, recycling it)
  for (uint32_t r = tid; r < R; r += NT) B.ring_cnt[(size_t)scan * R + r] = r_cnt[r];  // (k_front_redo's ring split starts from these)
  if (tid == 0) {
    B.n_filt[scan] = nf;
    B.flags[scan] = 0u;
  }
  if (scan == 0 && tid < FX_N_COUNTERS) B.counters[tid] = 0u;  // work-list counters of the batch (not the one the front kernels add to)
  uint32_t n = 0, ring_max = 0;
  for (uint32_t r = 0; r < R; ++r) {
    const uint32_t c = r_cnt[r];
    n += c;
    ring_max = max(ring_max, c);
  }
  if (n > CAP || nf > CAP || ring_max > P.max_ring_points || force_redo) {  // (force_redo: the test build's hook)
    if (tid == 0) {
      B.redo[atomicAdd(&B.counters[FX_CNT_REDO], 1u)] = scan;
      B.front_n[scan] = FX_NONE;
    }
    stamp_end();
    return;
  }
  if (n == 0) {
    no_keypoints();
    stamp_end();
    return;
  }
  // ---------------------------------------------------------------- B: ring split (ref: node.cpp:195-202), to B.ring_pts
  // k_front's stable counting sort — every wavefront ranks a contiguous slice of the survivors per ring by packed prefix
  // sums, the wavefronts' counts make each slice's base per ring — with the places in HBM instead of LDS.  (In two passes
  // over the slice — count, then rank and store, nothing per point kept in registers between them — it was slower:
  // k_front_ab alone 0.134 -> 0.146 ms, profiles/r06_experiments.md §1.)
  {
    const float2 *s_win = PL.win;
    uint32_t *cw = smem + O::cw, *cbase = cw + NW * FX_FRONT_RMAX;  // [NW][R] points of the wavefront's slice per ring; its first place per ring
    if (tid < 64) {
      const uint32_t c = tid < R ? r_cnt[tid] : 0u;
      uint32_t inc = c;
      inc = wave_incl_scan(inc);
      if (tid <= R) r_off[tid] = inc - c;  // (r_off[R] = n)
    }
    constexpr uint32_t kSub = (CAP + 64 * NW - 1) / (64 * NW);   // 64-point pieces of a wavefront's slice at most
    const uint32_t S = ((nf + 64u * NW - 1u) / (64u * NW)) * 64u;  // slice length
    uint32_t ring_a[kSub], place_b[kSub];  // ring_a: ring | 0x100: also in ring + 1 | place in the ring << 16; FX_NONE: in no ring
    uint32_t run_cnt = 0;  // lane r: points of ring r in this wavefront's slice so far
#pragma unroll
    for (uint32_t u = 0; u < kSub; ++u) {
      ring_a[u] = FX_NONE, place_b[u] = 0;
      if (u * 64u >= S) continue;  // (workgroup-uniform)
      const uint32_t i = wave * S + u * 64u + lane;
      const float el = i < nf ? s_el[i] : NAN;
      uint32_t mask = 0;
      int r_first = 0;
      if (isfinite(el)) mask = ring_membership(el, s_win, P.n_rings, el0, inv_step, r_first);
      // (windows of neighbouring rings share their edge only: at most two memberships, in consecutive rings)
      const int ra = mask ? r_first + (__ffs((int)mask) - 1) : -1;
      const bool two = (mask & (mask - 1u)) != 0u;
      if (mask) ring_a[u] = (uint32_t)ra | (two ? 0x100u : 0u);
      const uint32_t rb = (uint32_t)ra + 1u;
      const uint32_t one_a = mask ? 1u << (8u * ((uint32_t)ra & 3u)) : 0u, one_b = two ? 1u << (8u * (rb & 3u)) : 0u;
      const uint32_t ka = (uint32_t)ra >> 2, kb = rb >> 2;
      uint32_t mine_a = 0, mine_b = 0, tot = 0;  // the words of this lane's rings; lane r: the word of ring r's totals
#pragma nounroll
      for (uint32_t k = 0; 4u * k < R; ++k) {  // (a word of four rings a trip)
        uint32_t x = (ka == k ? one_a : 0u) + (kb == k ? one_b : 0u);
        x = wave_incl_scan(x);
        if (ka == k) mine_a = x;
        if (kb == k) mine_b = x;
        const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
        if ((lane >> 2) == k) tot = t;
      }
      const uint32_t before_a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)ra & 63u) << 2), (int)run_cnt);
      if (mask) ring_a[u] |= (before_a + ((mine_a >> (8u * ((uint32_t)ra & 3u))) & 0xffu) - 1u) << 16;
      if (__ballot(two)) {  // (wave-uniform; a point exactly on a window's edge: rare)
        const uint32_t before_b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((rb & 63u) << 2), (int)run_cnt);
        if (two) place_b[u] = before_b + ((mine_b >> (8u * (rb & 3u))) & 0xffu) - 1u;
      }
      if (lane < R) run_cnt += (tot >> (8u * (lane & 3u))) & 0xffu;
    }
    if (lane < R) cw[wave * R + lane] = run_cnt;
    __syncthreads();  // (every count is in)
    for (uint32_t t = tid; t < NW * R; t += NT) {
      const uint32_t w = t / R, r = t - w * R;
      uint32_t before = r_off[r];
      for (uint32_t x = 0; x < w; ++x) before += cw[x * R + r];
      cbase[t] = before;
    }
    if (tid < R) B.ring_off[(size_t)scan * R + tid] = r_off[tid];
    if (tid == 0) B.front_n[scan] = n;
    __syncthreads();
    float4 *dst = B.ring_pts + (size_t)scan * P.ring_slot_cap;
#pragma unroll
    for (uint32_t u = 0; u < kSub; ++u) {
      if (ring_a[u] == FX_NONE) continue;
      const uint32_t i = wave * S + u * 64u + lane;
      // rotated as the sweep rotated it for ~cloud (pcl::transformPointCloud's order, ref: node.cpp:161-166): the same record
      const float x = keep[3 * i], y = keep[3 * i + 1], z = keep[3 * i + 2];
      const float4 rec = make_float4(((M.R[0] * x + M.R[1] * y) + M.R[2] * z) + 0.0f, ((M.R[3] * x + M.R[4] * y) + M.R[5] * z) + 0.0f,
                                     ((M.R[6] * x + M.R[7] * y) + M.R[8] * z) + 0.0f, s_el[i]);
      const uint32_t ra = ring_a[u] & 0xffu;
      dst[cbase[wave * R + ra] + (ring_a[u] >> 16)] = rec;
      if (ring_a[u] & 0x100u) dst[cbase[wave * R + ra + 1u] + place_b[u]] = rec;
    }
  }
  stamp_end();
}

// (self_n: behind the SLICED streaming pass and ring split instead of k_front_ab — k_prep_count / k_prep_sliced / k_bucket_sliced,
//  several workgroups a scan: what a scan per call takes, where k_front's streaming pass is one workgroup on one
//  CU.  The kernel then does k_front_ab's bookkeeping itself: the scan's entries from the ring counts, the hand-over to
//  k_front_redo, the empty results, the support-list counters' slice.)
extern "C" __global__ __launch_bounds__(FX_FRONT_T, FX_PREP_OCC) void k_front_cd(FxDevParams P, FxBuffers B, uint32_t clk_slot, uint32_t merge_cap,
                                                                              uint32_t self_n, uint32_t force_redo) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  using O = FrontOff;
  constexpr uint32_t NT = FX_FRONT_T, CAP = FX_FRONT_CAP;
  const uint32_t scan = blockIdx.x, tid = threadIdx.x;
  const uint32_t R = (uint32_t)P.n_rings;
  uint32_t n;
  if (self_n) {
    {  // the support-list counters of the batch are cleared here, a slice per scan (k_gather fills them)
      const uint32_t per = (P.max_total_kp + gridDim.x - 1) / gridDim.x;
      const uint32_t z0 = scan * per, z1 = min(z0 + per, P.max_total_kp);
      for (uint32_t t = z0 + tid; t < z1; t += NT) B.s_cnt[t] = 0u;
      if (tid == 0) B.ovf_cnt[scan] = 0u;
    }
    if (B.meta[scan].n == 0u) {  // an empty scan (ref: node.cpp:209-210, 263-264): never handed on, as in k_front
      if (tid == 0) B.n_cand[scan] = 0u, B.n_kp[scan] = 0u, B.n_kpc[scan] = 0u;
      return;
    }
    uint32_t ring_max = 0;
    n = 0;
    for (uint32_t r = 0; r < R; ++r) {  // (uniform loads)
      const uint32_t c = B.ring_cnt[(size_t)scan * R + r];
      n += c;
      ring_max = max(ring_max, c);
    }
    if (n > CAP || B.n_filt[scan] > CAP || ring_max > P.max_ring_points || force_redo) {  // (k_front's conditions)
      if (tid == 0) B.redo[atomicAdd(&B.counters[FX_CNT_REDO], 1u)] = scan;
      return;
    }
    if (n == 0u) {  // ref: node.cpp:209-210, 263-264
      if (tid == 0) B.n_cand[scan] = 0u, B.n_kp[scan] = 0u, B.n_kpc[scan] = 0u;
      return;
    }
  } else {
    n = B.front_n[scan];
    if (n == 0u || n == FX_NONE) return;  // (k_front_ab wrote the empty results / handed the scan to k_front_redo)
  }
  float *px = reinterpret_cast<float *>(smem + O::px), *py = reinterpret_cast<float *>(smem + O::py), *pz = reinterpret_cast<float *>(smem + O::pz);
  uint32_t *r_off = smem + O::r_off, *r_cnt = smem + O::r_cnt;
  if (tid < R) {
    r_off[tid] = B.ring_off[(size_t)scan * R + tid];
    r_cnt[tid] = B.ring_cnt[(size_t)scan * R + tid];
  }
  if (tid == R) r_off[R] = n;
  const float4 *src = B.ring_pts + (size_t)scan * P.ring_slot_cap;
  constexpr uint32_t kSub = (CAP + NT - 1) / NT;
  float4 v[kSub];
#pragma unroll
  for (uint32_t u = 0; u < kSub; ++u) v[u] = src[min(u * NT + tid, n - 1u)];
#pragma unroll
  for (uint32_t u = 0; u < kSub; ++u) {
    const uint32_t i = u * NT + tid;
    if (i < n) px[i] = v[u].x, py[i] = v[u].y, pz[i] = v[u].z;
  }
  __syncthreads();
  front_cluster_merge<true>(P, B, scan, smem, n, clk_slot, merge_cap);
}

// k_front_cd with the lean image (FrontLeanOff): the points stay in B.ring_pts.
// (256 threads: held to 80 registers for three 512-thread workgroups a CU the kernel misses its occupancy target — the scalar
//  registers it spills take vector ones —; four wavefronts a workgroup at its natural 86 make four workgroups a CU)
#ifndef FX_FRONT_CDL_T
#define FX_FRONT_CDL_T 256
#endif
extern "C" __global__ __launch_bounds__(FX_FRONT_CDL_T) void k_front_cdl(FxDevParams P, FxBuffers B, uint32_t clk_slot, uint32_t merge_cap) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  using O = FrontLeanOff;
  const uint32_t scan = blockIdx.x, tid = threadIdx.x;
  const uint32_t R = (uint32_t)P.n_rings;
  const uint32_t n = B.front_n[scan];
  if (n == 0u || n == FX_NONE) return;  // (k_front_ab wrote the empty results / handed the scan to k_front_redo)
  uint32_t *r_off = smem + O::r_off, *r_cnt = smem + O::r_cnt;
  if (tid < R) {
    r_off[tid] = B.ring_off[(size_t)scan * R + tid];
    r_cnt[tid] = B.ring_cnt[(size_t)scan * R + tid];
  }
  if (tid == R) r_off[R] = n;
  __syncthreads();
  front_cluster_merge<true, FrontLeanOff, FX_FRONT_CDL_T>(P, B, scan, smem, n, clk_slot, merge_cap);
}

// The scans k_front could not take (k_front_redo, a workgroup of k_front's shape each — 512 threads, the same LDS image — so
// that a launch that finds nothing to do slips in beside the other batches' k_front workgroups instead of waiting for a whole
// free CU): the general kernels' bodies as far as that image holds them (the ring split to HBM, every ring through the
// workgroup tier, the merge tiers), starting from ~cloud and the ring counts.  What does not fit that image either (a ring of
// more than FX_FRONT_G_CAP1 points with more than FX_FRONT_G_CCAP2 size-admissible clusters, more candidates than either
// merge tier holds here) goes on to the slow tier (k_slow: the same bodies on scratch in HBM), ring by ring — launched with
// every batch, so the scan's result never depends on what earlier batches needed.
#define FX_FRONT_G_WORDS (FrontOff::end - SegCfg<FX_FRONT_T>::kWords)
#define FX_FRONT_G_CAP1 ((FX_FRONT_G_WORDS / (FX_RING_WORDS_PER_POINT + FX_RING_WORDS_PER_CLUSTER)) & ~3u)  // points = clusters
#define FX_FRONT_G_CCAP2 224u                                                                          // clusters of a larger ring
#define FX_FRONT_G_CAP2 (((FX_FRONT_G_WORDS - FX_RING_WORDS_PER_CLUSTER * FX_FRONT_G_CCAP2) / FX_RING_WORDS_PER_POINT) & ~3u)
__host__ __device__ constexpr uint32_t front_merge_cap_lds() {  // candidates the LDS merge tier holds in k_front's image
  uint32_t c = 512;
  while (merge_words(c + 64, c + 64, FX_FRONT_RMAX, true) <= FrontOff::end) c += 64;
  return c;
}
__device__ __forceinline__ void front_general(const FxDevParams &P, const FxBuffers &B, uint32_t scan, float el0, float inv_step, uint32_t huge_ccap,
                                              uint32_t *smem, bool force_slow) {
  constexpr int NT = FX_FRONT_T;
  const uint32_t R = (uint32_t)P.n_rings;
  if (R > 24)
    bucket_body<true, NT>(P, B, scan, el0, inv_step, smem);
  else
    bucket_body<false, NT>(P, B, scan, el0, inv_step, smem);
  wg_global_sync();
  bool pending = false;
  for (uint32_t ring = 0; ring < R; ++ring) {
    const uint32_t n = B.ring_cnt[(size_t)scan * R + ring];
    bool ok = false;
    if (force_slow)  // (the test build's hook: every ring through the slow tier)
      ok = false;
    else if (n > P.max_ring_points)
      ok = ring_body<NT>(P, B, scan, ring, P.max_ring_points, P.max_ring_points, smem, true);  // (flags the scan, touches no LDS)
    else if (n <= FX_FRONT_G_CAP1)
      ok = ring_body<NT>(P, B, scan, ring, FX_FRONT_G_CAP1, FX_FRONT_G_CAP1, smem, false);
    else if (n <= FX_FRONT_G_CAP2)
      ok = ring_body<NT>(P, B, scan, ring, FX_FRONT_G_CAP2, FX_FRONT_G_CCAP2, smem, false);
    __syncthreads();
    if (!ok) {  // more points, or more clusters, than this workgroup's LDS holds: the slow tier does the ring (and then the merge)
      if (threadIdx.x == 0) slow_ring(P, B, scan, ring);
      pending = true;
    }
  }
  if (pending) return;
  wg_global_sync();
  constexpr uint32_t kCapLds = front_merge_cap_lds();
  const uint32_t cap1 = min(kCapLds, P.max_candidates);
  if (merge_body<NT, true>(P, B, scan, cap1, cap1, smem, cap1 >= P.max_candidates)) return;
  __syncthreads();
  if (merge_words(P.max_candidates, huge_ccap, R, false) <= FrontOff::end) {
    merge_body<NT, false>(P, B, scan, P.max_candidates, huge_ccap, smem, true);
    return;
  }
  if (threadIdx.x == 0) slow_push(B, scan);  // (more candidates than either merge tier holds here)
}

extern "C" __global__ __launch_bounds__(FX_FRONT_T) void k_front_redo(FxDevParams P, FxBuffers B, float el0, float inv_step, uint32_t huge_ccap,
                                                                    uint32_t force_slow) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t n_redo = B.counters[FX_CNT_REDO];
  for (uint32_t w = blockIdx.x; w < n_redo; w += gridDim.x) {
    front_general(P, B, B.redo[w], el0, inv_step, huge_ccap, smem, force_slow != 0u);
    __syncthreads();
  }
}

// The slow tier behind every LDS-sized one: rings of more points (or clusters) than a workgroup's LDS holds and scans of more
// candidates than the merge tiers hold, up to the context's limits, by the SAME bodies with their per-point and per-cluster
// arrays in a scratch region of HBM (GS).  Always launched — 256 threads and 9 KB of LDS place anywhere, and an empty
// launch costs a few microseconds — so what a scan gets never depends on what earlier batches needed (ref: node.cpp:72-145
// never drops a scan).  A listed scan has its pending rings done (in ring order), then its merge (again, if a merge tier has
// already run on the rings that were ready).  Grid <= P.gs_slots: workgroup b owns scratch region b.
#define FX_SLOW_T 256
__host__ __device__ constexpr size_t slow_ring_words(uint32_t cap) { return (size_t)(FX_RING_WORDS_PER_POINT + FX_RING_WORDS_PER_CLUSTER) * cap; }
__host__ __device__ constexpr size_t slow_merge_words(uint32_t cap, uint32_t ccap) { return 4 * (size_t)cap + cap + merge_aux_words(cap) + 3 * (size_t)ccap; }
__device__ __forceinline__ void offsets_body(const FxDevParams &P, const FxBuffers &B, uint32_t batch, uint32_t clk_next, uint32_t *s_w);
// The batch's keypoint offsets (offsets_body: k_offsets' work until round 5, a launch of its own) are computed by whichever
// workgroup of this launch finishes LAST — a ticket: every workgroup makes its writes visible device-wide, draws a number,
// and the one that draws gridDim.x - 1 knows all the others are done.  So the grid need not be one workgroup for the launch
// to stand in for k_offsets (with one workgroup a batch that suddenly hands on many scans went through it one
// after the other), and every batch has one launch less in its chain.
extern "C" __global__ __launch_bounds__(FX_SLOW_T) void k_slow(FxDevParams P, FxBuffers B, uint32_t huge_ccap, uint32_t batch, uint32_t clk_next) {
  static_assert(FX_SLOW_T == FX_WG, "k_slow stands in for k_offsets");
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t n_slow = B.counters[FX_CNT_REDO + 1];
  const uint32_t R = (uint32_t)P.n_rings, PW = (R + 31u) / 32u;
  uint32_t *gs = B.gs_pool + (size_t)blockIdx.x * P.gs_words;
  bool did = false;  // (workgroup-uniform) this workgroup wrote results
  for (uint32_t w = blockIdx.x; w < n_slow; w += gridDim.x) {
    did = true;
    const uint32_t scan = B.slow[w];
    uint32_t *pend = B.ring_pending + (size_t)scan * PW;
    for (uint32_t ring = 0; ring < R; ++ring) {
      if (!((pend[ring >> 5] >> (ring & 31u)) & 1u)) continue;  // (workgroup-uniform)
      ring_body<FX_SLOW_T, true>(P, B, scan, ring, P.max_ring_points, P.max_ring_points, smem, true, gs);
      wg_global_sync();
    }
    for (uint32_t t = threadIdx.x; t < PW; t += FX_SLOW_T) pend[t] = 0u;  // (the bit map and the marker start every batch clear)
    if (threadIdx.x == 0) B.slow_state[scan] = 0u;
    wg_global_sync();
    merge_body<FX_SLOW_T, false, false, true>(P, B, scan, P.max_candidates, huge_ccap, smem, true, nullptr, gs);
    wg_global_sync();
  }
  // ---- the last workgroup of the launch does the offsets
  if (did && gridDim.x > 1u) __threadfence();  // (EVERY wavefront that stored results waits for its own stores and writes them back before the ticket is drawn)
  __syncthreads();  // (smem is free)
  if (threadIdx.x == 0) {
    uint32_t last = 1u;
    if (gridDim.x > 1u) {
      __threadfence();  // (release at device scope: what this workgroup's scans got — n_kp, flags — before its ticket)
      last = atomicAdd(&B.counters[FX_CNT_SLOW_TICKET], 1u) == gridDim.x - 1u ? 1u : 0u;
      if (last) B.counters[FX_CNT_SLOW_TICKET] = 0u;  // (for the next batch: launches of a context are ordered)
    }
    smem[0] = last;
  }
  __syncthreads();
  if (smem[0]) {  // (workgroup-uniform)
    __syncthreads();  // (smem[0] is read by all before offsets_body reuses the words)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the other workgroups' results, whatever this CU's L1 holds
    offsets_body(P, B, batch, clk_next, smem);
  }
}

// ====================================================================== stage 4: offsets
// (by one FX_WG-thread workgroup; s_w: FX_NWAVE words of LDS)
__device__ __forceinline__ void offsets_body(const FxDevParams &P, const FxBuffers &B, uint32_t batch, uint32_t clk_next, uint32_t *s_w) {
  uint32_t run = 0;
  for (uint32_t b0 = 0; b0 < batch; b0 += FX_WG) {
    const uint32_t b = b0 + threadIdx.x;
    const uint32_t c = b < batch ? B.n_kp[b] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan<FX_WG>(c, s_w, tot);
    if (b < batch) {
      const uint32_t off = run + ex;
      B.kp_offset[b] = off;
      if (off + c > P.max_total_kp) atomicOr(&B.flags[b], FX_FLAG_TOTAL_KP_OVERFLOW);
    }
    run += tot;
  }
  if (threadIdx.x == 0) {
    // what the rarely used tiers had to do in this batch (every kernel that adds to these lists has completed): the host
    // sizes the next batch's launches of those tiers by it — an empty 256-workgroup launch of a kernel that takes a whole
    // CU's LDS waits for 256 CUs to drain, in the way of the other batches in flight
    uint32_t l1 = 0, l2 = 0;
    for (uint32_t c = 0; c < 8u; ++c) l1 = max(l1, B.counters[FX_CNT_LARGE + c]), l2 = max(l2, B.counters[FX_CNT_LARGE2 + c]);
    B.tier_hint[0] = l1;
    B.tier_hint[1] = l2;
    B.tier_hint[2] = B.counters[1];
    B.tier_hint[3] = B.counters[9];
    B.tier_hint[6] = B.counters[FX_CNT_REDO];  // scans k_front handed to k_front_redo (which has read the count by now)
    B.tier_hint[7] = B.counters[FX_CNT_REDO + 1];  // scans handed to the slow tier (k_slow has read the count by now)
    B.counters[FX_CNT_REDO] = 0u;
    B.counters[FX_CNT_REDO + 1] = 0u;
    B.clk[2 * clk_next] = ~0ull;  // the clock slot the next batch's first kernel stamps
    B.clk[2 * clk_next + 1] = 0ull;
    B.kp_offset[batch] = run;
    B.seq[0] += 1ull;  // batch tag of the dense tier's density cache (device side: a replayed HIP graph advances it too)
  }
}

// ====================================================================== stage 5: descriptors
// pcl::ShapeContext3DEstimation (ref: node.cpp:329-355):
//   k_gather        one pass over each scan tests every (rotated) point against the keypoints of the
//                   scan (binned along x) and appends the points within R + R/5 of a keypoint to that
//                   keypoint's support list (a superset of the neighbour query and of every density query);
//                   (the rows themselves are cleared by k_desc_group, which sees every row once)
//   k_desc_group    4 keypoints per wavefront: support sets of <= 64 points (the bulk)
//   k_desc_mid      one launch: one wavefront per keypoint (65..192 support points: FX_WAVE_CAP) and one 256-thread
//                   workgroup per keypoint (lists of up to 1024 entries)
//                   -- all three: bins + density + weight per neighbour, sort by (bin, d2, index) ==
//                      PCL's accumulation order, sequential fp32 sum per bin; angles in fp32, exact next to a bin edge --
//   k_dense_*       larger support sets (dense many-ring scans) and overflowed lists: cell-sorted support sets in HBM
//                   pools, a per-scan density cache shared by the rows, many wavefronts per row
//   (3DSC's RNG ordinal rule — a keypoint without neighbours draws no x-axis — is applied by k_gather, which knows
//    which keypoints have one before any descriptor is computed; k_rng_ord when several workgroups share a scan)

__device__ __forceinline__ uint32_t dense_class(uint32_t nS);
__device__ __forceinline__ uint32_t dense_class_counter(uint32_t cls);

// Which scan does global keypoint row w belong to?  kp_offset is an exclusive prefix.
__device__ __forceinline__ uint32_t scan_of_row(const uint32_t *kp_offset, uint32_t batch, uint32_t w) {
  uint32_t lo = 0, hi = batch;  // invariant: kp_offset[lo] <= w < kp_offset[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (kp_offset[mid] <= w)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// (j, k, l) bin and lookup-table weight of one neighbour.
// kp = keypoint (origin), b = neighbour, xa = the keypoint's 3DSC x-axis (z component is -0).
// FAST = false: phi / theta through fp64 atan2 / acos rounded once to fp32 (the oracle's policy).
// FAST = true: fp32 atan2f / acosf; when an angle is within FX_FAST_EPS_DEG of a bin edge, i.e. when fp32 could put
// the neighbour into another bin than the exact evaluation, the exact evaluation is made for that neighbour
// (everywhere else the two agree on the bin, and the bin is all that is used downstream).
#define FX_FAST_EPS_DEG 5e-4f
__device__ __forceinline__ bool near_multiple(float v, float step, float inv_step) {
  const float t = v * inv_step;
  return fabsf(t - rintf(t)) * step < FX_FAST_EPS_DEG;
}
// the exact angles (fp64 atan2 / acos rounded once: the oracle's policy) for the rare neighbour whose fp32 angle
// falls next to a bin edge; kept out of line so that its registers and code stay out of the fast kernels' loops
__device__ __noinline__ void sc3d_angles_exact(float cn, float xd, float tc, float *phi, float *theta) {
  *phi = (float)atan2((double)cn, (double)xd) * 57.29578f;
  *theta = (float)acos((double)tc) * 57.29578f;
}
template <bool FAST>
__device__ __forceinline__ uint32_t sc3d_bin(const float4 kp, float bx, float by, float bz, float d2, const float2 xa,
                                             const FxScTables *T, float &lut, bool &amb) {
  const float nx = 0.0f, ny = 0.0f, nz = 1.0f;  // every normal is +z (ref: node.cpp:337-340)
  const float ax = xa.x, ay = xa.y, az = -0.0f;
  const float r = sqrtf(d2);
  // pcl::geometry::project(neighbour, origin, normal, proj); proj -= origin; proj.normalize()
  const float pox = bx - kp.x, poy = by - kp.y, poz = bz - kp.z;
  const float lambda = nx * pox + (ny * poy + nz * poz);
  float p0 = (bx - lambda * nx) - kp.x;
  float p1 = (by - lambda * ny) - kp.y;
  float p2 = (bz - lambda * nz) - kp.z;
  {
    const float zz = p0 * p0 + (p1 * p1 + p2 * p2);
    if (zz > 0.0f) {
      const float s = sqrtf(zz);
      p0 /= s;
      p1 /= s;
      p2 /= s;
    }
  }
  // cross = x_axis x proj; phi = atan2(|cross|, x_axis . proj) in degrees, mirrored by sign
  const float c0 = ay * p2 - az * p1;
  const float c1 = az * p0 - ax * p2;
  const float c2 = ax * p1 - ay * p0;
  const float cn = sqrtf(c0 * c0 + (c1 * c1 + c2 * c2));
  const float xd = ax * p0 + (ay * p1 + az * p2);
  float phi = (FAST ? atan2f(cn, xd) : (float)atan2((double)cn, (double)xd)) * 57.29578f;
  const float cdn = c0 * nx + (c1 * ny + c2 * nz);
  phi = cdn < 0.f ? (360.0f - phi) : phi;
  // theta = acos(clamp(normal . normalized(neighbour - origin))) in degrees
  float n0 = pox, n1 = poy, n2 = poz;
  {
    const float zz = n0 * n0 + (n1 * n1 + n2 * n2);
    if (zz > 0.0f) {
      const float s = sqrtf(zz);
      n0 /= s;
      n1 /= s;
      n2 /= s;
    }
  }
  float theta = nx * n0 + (ny * n1 + nz * n2);
  const float mx = (-1.0f < theta) ? theta : -1.0f;  // std::max(-1.0f, theta)
  const float tc = (mx < 1.0f) ? mx : 1.0f;          // std::min(1.0f, .)
  theta = (FAST ? acosf(tc) : (float)acos((double)tc)) * 57.29578f;
#ifndef FX_TRIG_LITERAL_F32  // (diagnostic build: PCL's literal atan2f / acosf — this device's — with no exact re-evaluation;
                             //  tools/trig_policy.py counts the bins that moves)
  if (FAST) {
    // edges: phi_div = 30 l, theta_div = l * (180/11) (A.8-2); non-finite angles go the exact way too
    if (!(phi == phi) || !(theta == theta) || near_multiple(phi, 30.0f, 1.0f / 30.0f) ||
        near_multiple(theta, 180.0f / 11.0f, 11.0f / 180.0f)) {
      float pe, te;
      sc3d_angles_exact(cn, xd, tc, &pe, &te);
      phi = cdn < 0.f ? (360.0f - pe) : pe;
      theta = te;
    }
  }
#endif

  // PCL scans each edge table for the first edge >= the value and falls back to bin 0 when there is
  // none (A.8-9).  The edges ascend, so "first edge >= v" is the number of edges below v: counted
  // without branches; a count that runs off the table (or a NaN, which counts nothing) is the fallback.
  uint32_t j = 0, kk = 0, l = 0;
#pragma unroll
  for (uint32_t rad = 1; rad < 16; ++rad) j += (r > T->radii[rad]) ? 1u : 0u;
#pragma unroll
  for (uint32_t ang = 1; ang < 12; ++ang) kk += (theta > T->theta[ang]) ? 1u : 0u;
#pragma unroll
  for (uint32_t ang = 1; ang < 13; ++ang) l += (phi > T->phi[ang]) ? 1u : 0u;
  j = j == 15u ? 0u : j;
  kk = kk == 11u ? 0u : kk;
  l = l == 12u ? 0u : l;
  lut = T->lut[kk * 15 + j];
  return (l * 11 + kk) * 15 + j;
}
// 3DSC skips a neighbour whose squared distance "equals" zero: pcl::utils::equal(nn_dists[ne], 0.0f) with its
// default tolerance std::numeric_limits<float>::min() (pcl/common/utils.h), i.e. |d2 - 0| < FLT_MIN — the point
// the keypoint sits on (or one a subnormal d2 away), nothing farther.
// (-DFX_SKIP_EPSILON: the other reading of PCL's skip rule — SURVEY.md A.8-6's `< FLT_EPSILON` instead of pcl::utils::equal's
//  default tolerance — as ONE constant, so that whichever a PCL run confirms, the device and the oracle's switch
//  FXO_POLICY_SKIP_EPSILON are already proven to flip together: lib/libfx_hip_skipeps.so, tests/test_gpu_skip_policy.py)
#ifdef FX_SKIP_EPSILON
__device__ __forceinline__ bool sc3d_is_origin(float d2) { return fabsf(d2 - 0.0f) < FLT_EPSILON; }
#else
__device__ __forceinline__ bool sc3d_is_origin(float d2) { return fabsf(d2 - 0.0f) < FLT_MIN; }
#endif
__device__ __forceinline__ unsigned long long sc3d_key(uint32_t bin, float d2, uint32_t idx) {
  return ((unsigned long long)bin << 52) | ((unsigned long long)__float_as_uint(d2) << 20) | (unsigned long long)idx;
}

// ---------------------------------------------------------------- k_gather
// Keypoints of the scan are binned into xy cells at least one support radius wide, and every cell lists the
// keypoints of the 3 x 3 cells around it: a point looks its cell up once and tests exactly the keypoints that can
// be within reach.  Points with candidates are parked in a per-wavefront LDS queue and tested with all lanes busy
// (in firing order only some lanes of a wavefront hold such points).  Groups of 64 points that k_prep found
// wholly out of the filter box's reach are not even loaded.
#ifndef FX_GATHER_T
#define FX_GATHER_T 256  // (384 and 512 threads measured: see DESIGN.md)
#endif
#define FX_GATHER_WIDE_T 1024  // one workgroup per scan for batches that do not fill the GPU with 256-thread ones
#define FX_GATHER_G 32  // cells per axis at most
#define FX_GATHER_CELLS (FX_GATHER_G * FX_GATHER_G)
#ifndef FX_GATHER_STAGE
#define FX_GATHER_STAGE 384  // hits one workgroup stages between flushes (a larger stage costs more in occupancy than it saves in flushes)
#endif
#ifndef FX_GATHER_DIRECT
#define FX_GATHER_DIRECT 1   // one workgroup per scan: hits go straight to their list slots (positions are LDS counters)
#endif
#ifndef FX_GATHER_QUEUE
#define FX_GATHER_QUEUE 128  // points one wavefront of k_gather parks before it drains them (drained once more than 64 are waiting:
                             // measured 72 .. 192 — 96 to 160 are level, 1.2 % above 80; k_gather_wide keeps 80: sixteen queues)
#endif
#define FX_GATHER_QUEUE_WIDE 80
#define FX_GATHER_KCAP 2048  // keypoints the LDS tables hold (148 KB with the wide workgroup's queues); scans with more are gathered in passes
__host__ __device__ constexpr uint32_t gather_queue(uint32_t nt) { return nt <= 256u ? (uint32_t)FX_GATHER_QUEUE : (uint32_t)FX_GATHER_QUEUE_WIDE; }
__host__ __device__ inline uint32_t gather_words(uint32_t mk, uint32_t nt) {
  uint32_t w = 32 + 4 * mk;                                   // scratch, keypoints
  w += FX_GATHER_CELLS + 4;                                   // cell table (the fill cursors borrow the staging area)
  w += ((9 * mk + 1) / 2 + 3) & ~3u;                          // cell lists (uint16)
  w += (5 * mk + 3) & ~3u;                                    // staged hits per keypoint, reserved list positions, list lengths, has-a-neighbour flags, overflow slots
  w += FX_GATHER_STAGE + 4 * FX_GATHER_STAGE;                 // staged hits: meta, points
  w += (nt / 64) * gather_queue(nt) * 5;                      // per-wavefront queues: points, cell info
  return w;
}
// PASSES: contexts whose max_keypoints exceeds FX_GATHER_KCAP (the instance without the pass loop keeps its registers)
// MODE (batches of few big scans: 64 scans of config 5 are 64 workgroups on 256 CUs): several workgroups a scan WITHOUT the
// staged appends and their global atomics (which serialise on a dense scan's hot keypoints: 5 x slower, see drain below) — two
// launches over the same slices.  1: the counting pass — every hit adds one to its keypoint's LDS counter, nothing is stored;
// the slice's counts go to B.gather_cnt.  2: the scatter pass — a keypoint's first list position for this slice is the sum
// of the earlier slices' counts, so its hits go straight to their slots as with one workgroup a scan; the overflow region's
// positions (entries beyond list_cap) follow from the same counts.  The near sectors are read and tested twice, by several
// times as many workgroups.  0: as before (one workgroup a scan, or the streaming mode's staged appends).
template <int NT, bool PASSES, int MODE = 0>
__device__ __forceinline__ void gather_body(const FxDevParams &P, const FxBuffers &B, float box_margin, uint32_t *smem) {
  static_assert(MODE == 0 || !PASSES, "the counted slices hold a scan's keypoints in one pass");
  static_assert(MODE == 0 || FX_GATHER_DIRECT, "the scatter pass writes its hits straight to their slots");
  constexpr uint32_t FX_GATHER_NW = NT / 64;
  // keypoints the LDS tables hold at a time: a scan with more (a forest under the launch preset) is gathered in passes —
  // the scan's near sectors are streamed once per FX_GATHER_KCAP keypoints
  const uint32_t MK = PASSES ? (uint32_t)FX_GATHER_KCAP : P.max_keypoints;  // (the host launches the PASSES instance beyond FX_GATHER_KCAP)
  uint32_t *s_w = smem;                                          // 0..5 keypoint box, 8 staged hits
  float4 *s_kp = reinterpret_cast<float4 *>(smem + 32);          // 16-byte aligned
  uint32_t *s_cell = smem + 32 + 4 * MK;                         // [CELLS + 4] list start | entries << 20
  uint16_t *s_flat = reinterpret_cast<uint16_t *>(s_cell + FX_GATHER_CELLS + 4);  // [9 MK] keypoint ids, cell by cell
  // hits are staged in LDS and appended to the keypoints' lists in bulk: one global atomic per
  // (keypoint, flush) reserves the slots instead of one returning atomic per hit
  uint32_t *s_kcnt = reinterpret_cast<uint32_t *>(s_flat) + (((9 * MK + 1) / 2 + 3) & ~3u);  // [MK] staged hits per keypoint
  uint32_t *s_kbase = s_kcnt + MK;                               // [MK] reserved list position
  uint32_t *s_kpos = s_kbase + MK;                               // [MK] list length so far (one workgroup per scan)
  uint32_t *s_knbr = s_kpos + MK;                                // [MK] 1: some point lies within the search radius (one workgroup per scan)
  uint32_t *s_kovf = s_knbr + MK;                                // [MK] overflow-region slot of the keypoint's staged ordinal 0 (entries beyond list_cap)
  uint32_t *s_smeta = s_kcnt + ((5 * MK + 3) & ~3u);             // [STAGE] keypoint << 16 | staged ordinal
  const bool solo = gridDim.x == 1 && MODE == 0;
  const bool own = solo || MODE == 2;  // list positions are this workgroup's own LDS counters
  float4 *s_spt = reinterpret_cast<float4 *>(s_smeta + FX_GATHER_STAGE);
  uint32_t *s_cur = reinterpret_cast<uint32_t *>(s_spt);        // [CELLS] fill cursors while the lists are built (4 STAGE >= CELLS words)
  const uint32_t scan = blockIdx.y, slice = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr uint32_t kQueue = gather_queue(NT);
  float4 *q_pt = s_spt + FX_GATHER_STAGE + wave * kQueue;                                                  // this wavefront's queue
  uint32_t *q_info = reinterpret_cast<uint32_t *>(s_spt + FX_GATHER_STAGE + FX_GATHER_NW * kQueue) + wave * kQueue;
  uint32_t K_all = B.n_kp[scan];
  if (K_all == 0) return;
  const uint32_t row_first = B.kp_offset[scan];
  if (row_first >= P.max_total_kp) return;
  if (row_first + K_all > P.max_total_kp) K_all = P.max_total_kp - row_first;
  const FxScanMeta M = B.meta[scan];
  const bool passes = PASSES && K_all > MK;  // (then the has-a-neighbour flags of all keypoints meet in HBM, as with several workgroups a scan)
  if (tid == 0) s_w[9] = 0;  // entries in the scan's overflow region (one workgroup per scan)
  uint32_t kb = 0;
  do {
  const uint32_t K = PASSES ? min(MK, K_all - kb) : K_all, row0 = row_first + kb;
  __syncthreads();  // (the previous pass is done with the tables)
  // row -> (scan, keypoint) map for the per-keypoint kernels (one load instead of a binary search)
  if (slice == 0 && MODE != 1)
    for (uint32_t k = tid; k < K; k += NT) B.row_map[row0 + k] = make_uint2(scan, kb + k);
  // keypoints of the scan -> LDS; their bounding box (ordered-uint atomics) for a cheap reject
  if (tid < 6) s_w[tid] = (tid & 1) ? f2ord(-INFINITY) : f2ord(INFINITY);
  if (tid == 0) s_w[8] = 0;  // staged hits
  for (uint32_t c = tid; c < FX_GATHER_CELLS; c += NT) s_cell[c] = 0;
  __syncthreads();
  for (uint32_t k = tid; k < K; k += NT) {
    const float4 kp = B.keypoints[(size_t)scan * P.max_keypoints + kb + k];
    s_kp[k] = kp;
    s_kcnt[k] = 0;
    s_kpos[k] = 0;
    s_knbr[k] = 0;
    if (slice == 0 && MODE != 1) B.row_kp[row0 + k] = kp;
    atomicMin(&s_w[0], f2ord(kp.x));
    atomicMax(&s_w[1], f2ord(kp.x));
    atomicMin(&s_w[2], f2ord(kp.y));
    atomicMax(&s_w[3], f2ord(kp.y));
    atomicMin(&s_w[4], f2ord(kp.z));
    atomicMax(&s_w[5], f2ord(kp.z));
  }
  __syncthreads();
  const float kx0 = ord2f(s_w[0]), kx1 = ord2f(s_w[1]), ky0 = ord2f(s_w[2]), ky1 = ord2f(s_w[3]);
  const float bx0 = kx0 - box_margin, bx1 = kx1 + box_margin;
  const float by0 = ky0 - box_margin, by1 = ky1 + box_margin;
  const float bz0 = ord2f(s_w[4]) - box_margin, bz1 = ord2f(s_w[5]) + box_margin;
  // cell width >= the support radius (+ margin): a point within reach of a keypoint is in the keypoint's cell or
  // in one next to it; wider when the keypoints spread over more than G cells of that width
  const float wx = fmaxf(box_margin, (bx1 - bx0) / (float)(FX_GATHER_G - 1) * 1.0001f + 1e-6f);
  const float wy = fmaxf(box_margin, (by1 - by0) / (float)(FX_GATHER_G - 1) * 1.0001f + 1e-6f);
  const float inv_wx = 1.0f / wx, inv_wy = 1.0f / wy;
  const int ncx = min((int)((bx1 - bx0) * inv_wx) + 1, FX_GATHER_G), ncy = min((int)((by1 - by0) * inv_wy) + 1, FX_GATHER_G);
  auto cell_x = [&](float x) { return min(max((int)floorf((x - bx0) * inv_wx), 0), ncx - 1); };
  auto cell_y = [&](float y) { return min(max((int)floorf((y - by0) * inv_wy), 0), ncy - 1); };
  for (uint32_t k = tid; k < K; k += NT) {
    const int cx = cell_x(s_kp[k].x), cy = cell_y(s_kp[k].y);
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx)
        if (cx + dx >= 0 && cx + dx < ncx && cy + dy >= 0 && cy + dy < ncy) atomicAdd(&s_cell[(cy + dy) * ncx + cx + dx], 1u);
  }
  __syncthreads();
  if (tid < 64) {  // counts -> start | count << 20, and the fill cursors, by one wavefront
    const uint32_t n_cells = (uint32_t)(ncx * ncy), per = (n_cells + 63) / 64;
    uint32_t sum = 0;
    for (uint32_t u = 0; u < per; ++u) {
      const uint32_t ci = tid * per + u;
      sum += ci < n_cells ? s_cell[ci] : 0u;
    }
    uint32_t incl = sum;
    incl = wave_incl_scan(incl);
    uint32_t run = incl - sum;
    for (uint32_t u = 0; u < per; ++u) {
      const uint32_t ci = tid * per + u;
      if (ci < n_cells) {
        const uint32_t c = s_cell[ci];
        s_cell[ci] = run | (c << 20);
        s_cur[ci] = run;
        run += c;
      }
    }
  }
  __syncthreads();
  for (uint32_t k = tid; k < K; k += NT) {
    const int cx = cell_x(s_kp[k].x), cy = cell_y(s_kp[k].y);
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx)
        if (cx + dx >= 0 && cx + dx < ncx && cy + dy >= 0 && cy + dy < ncy)
          s_flat[atomicAdd(&s_cur[(cy + dy) * ncx + cx + dx], 1u)] = (uint16_t)k;
  }
  __syncthreads();

  if (MODE == 2) {  // this slice's first position in every list, and in the scan's overflow region, from the counting pass
    const uint32_t S = gridDim.x;
    const uint32_t *cnt = B.gather_cnt + (size_t)scan * S * P.max_keypoints;
    uint32_t ovf_before = 0, ovf_all = 0;
    for (uint32_t k = tid; k < K; k += NT) {
      uint32_t run = 0, mine = 0;
      for (uint32_t sl = 0; sl < S; ++sl) {
        const uint32_t c = cnt[(size_t)sl * P.max_keypoints + k];
        const uint32_t o = run + c > P.list_cap ? run + c - max(run, P.list_cap) : 0u;  // this slice's entries beyond the list
        if (sl == slice) mine = run;
        if (sl < slice) ovf_before += o;
        ovf_all += o;
        run += c;
      }
      s_kpos[k] = mine;
      if (slice == 0) B.s_cnt[row0 + k] = run;  // (the TRUE support size, as with one workgroup a scan)
    }
    if (ovf_before) atomicAdd(&s_w[9], ovf_before);
    if (slice == 0 && ovf_all) atomicAdd(&B.ovf_cnt[scan], ovf_all);  // (cleared by the front kernels)
    __syncthreads();
  }
  constexpr uint32_t kTile = NT * 4;  // 1024 points: wave w owns [256 w, 256 w + 256), 64 consecutive points per load
  const uint32_t n = M.n;
  uint32_t chunk = (n + gridDim.x - 1) / gridDim.x;
  chunk = (chunk + kTile - 1) / kTile * kTile;
  // (the counted slices take the scan's tiles in turn — tile t is slice t mod S's —: a pole's hits lie in a few consecutive
  //  tiles, and contiguous ranges gave them all to one slice)
  const uint32_t tstep = MODE != 0 ? gridDim.x * kTile : kTile;
  const uint32_t lo = MODE != 0 ? slice * kTile : slice * chunk;
  const uint32_t hi = MODE != 0 ? n : (lo + chunk < n ? lo + chunk : n);
  const uint32_t *near_bits = B.near_bits + (size_t)scan * P.near_words;
  // called by the whole workgroup, after a barrier: reserve list positions, write the staged hits out
  // Entries beyond a list's list_cap slots go to the scan's overflow region (B.ovf_pts / B.ovf_kp, ovf_cap entries a
  // scan, unordered): the dense tier (k_dense_sort) collects a row's entries from there, so no support set is truncated.
  float4 *ovf_pts = B.ovf_pts + (size_t)scan * P.ovf_cap;
  uint32_t *ovf_kp = B.ovf_kp + (size_t)scan * P.ovf_cap;
  auto ovf_reserve = [&](uint32_t n) -> uint32_t { return own ? atomicAdd(&s_w[9], n) : atomicAdd(&B.ovf_cnt[scan], n); };
  auto flush = [&](uint32_t staged) {
    staged = min(staged, (uint32_t)FX_GATHER_STAGE);
    for (uint32_t k = tid; k < K; k += NT) {
      const uint32_t c = s_kcnt[k];
      if (c) {
        uint32_t base;
        if (solo) {  // this workgroup is the only writer of the scan's lists: positions are its own running counts
          base = s_kpos[k];
          s_kpos[k] = base + c;
        } else {
          base = atomicAdd(&B.s_cnt[row0 + k], c);
        }
        s_kbase[k] = base;
        if (base + c > P.list_cap) {  // ordinals from `first` on overflow the list
          const uint32_t first = base < P.list_cap ? P.list_cap - base : 0u;
          s_kovf[k] = ovf_reserve(c - first) - first;  // (wraps below zero for ordinals that stay in the list: unused there)
        }
      }
      s_kcnt[k] = 0;
    }
    __syncthreads();
    for (uint32_t e = tid; e < staged; e += NT) {
      const uint32_t meta = s_smeta[e], k = meta >> 16;
      const uint32_t pos = s_kbase[k] + (meta & 0xffffu);
      if (pos < P.list_cap) {
        B.s_pts[(size_t)(row0 + k) * P.list_cap + pos] = s_spt[e];
      } else {
        const uint32_t slot = s_kovf[k] + (meta & 0xffffu);
        if (slot < P.ovf_cap) {
          ovf_pts[slot] = s_spt[e];
          ovf_kp[slot] = k;
        }
      }
    }
    if (tid == 0) s_w[8] = 0;
    __syncthreads();
  };
  // this wavefront's parked points against the keypoints of their cells' lists, one point per lane
  uint32_t qn = 0;
  auto drain = [&]() {
    wave_sync_lds();
    for (uint32_t t = lane; t < qn; t += 64) {
      const float4 pq = q_pt[t];
      const uint32_t info = q_info[t], st = info & 0xfffffu, cnt = info >> 20;
      for (uint32_t e = 0; e < cnt; ++e) {
        const uint32_t k = s_flat[st + e];
        const float4 kp = s_kp[k];
        const float d2 = dist2(kp.x, kp.y, kp.z, pq.x, pq.y, pq.z);
        if (d2 < P.r2_support) {
          // 3DSC draws its x-axis only for keypoints that have a neighbour (d2 < R^2: the descriptor kernels' count)
          if (d2 < P.r2_search && MODE != 2) {
            if (solo && !passes)
              s_knbr[k] = 1u;
            else
              B.kp_nbrs[(size_t)scan * P.max_keypoints + kb + k] = 1u;  // (cleared by the merge stage; k_rng_ord reads it)
          }
          if (MODE == 1) {  // the counting pass: one more entry for this keypoint from this slice
            atomicAdd(&s_kpos[k], 1u);
            continue;
          }
          // One workgroup per scan: the list position is the workgroup's own LDS counter, so the hit goes straight to
          // its slot (no staging, no flush, no workgroup barrier in the tile loop).  Several workgroups per scan
          // (small batches): hits are staged and a flush reserves positions with one global atomic per keypoint —
          // fine for a handful of sparse scans, hopeless for dense ones (hundreds of flushes: BASELINE config 3 with four
          // workgroups per scan 5.4 ms against 0.66 with one; a global atomic per hit: 7.3 ms, the hot keypoints' counters
          // serialise), which is why dense scans get one WIDE workgroup instead (k_gather_wide).
          const uint32_t slot = (FX_GATHER_DIRECT && own) ? FX_GATHER_STAGE : atomicAdd(&s_w[8], 1u);
          if (slot < FX_GATHER_STAGE) {
            s_spt[slot] = pq;
            s_smeta[slot] = (k << 16) | atomicAdd(&s_kcnt[k], 1u);
          } else {  // (also: stage full — a burst of hits within one tile)
            const uint32_t pos = own ? atomicAdd(&s_kpos[k], 1u) : atomicAdd(&B.s_cnt[row0 + k], 1u);
            if (pos < P.list_cap) {
              B.s_pts[(size_t)(row0 + k) * P.list_cap + pos] = pq;
            } else {
              const uint32_t os = ovf_reserve(1u);
              if (os < P.ovf_cap) {
                ovf_pts[os] = pq;
                ovf_kp[os] = k;
              }
            }
          }
        }
      }
    }
    wave_sync_lds();
    qn = 0;
  };
  // the 64-byte sectors (four points) of a tile that hold a point within reach of the filter box (k_prep): this
  // wavefront's 256 points = the 64 sector bits one wavefront of k_prep wrote (its tiles are two of these)
  // (The wavefront's sector words of 64 tiles at a time, lane t those of tile t — ONE load, then a readlane per tile: fetched
  //  tile by tile, the two words were a global round trip in front of every tile's loads, which they predicate.)
  uint2 nb_words = make_uint2(0u, 0u);
  uint32_t nb_first = lo;  // the tile lane 0 holds
  auto nb_fill = [&](uint32_t i_first) {
    nb_first = i_first;
    const uint32_t i = i_first + lane * tstep;
    nb_words = make_uint2(0u, 0u);
    if (i < hi) {
      const uint32_t w = (i + wave * 256u) / 128u;  // (sector order: this wavefront's 256 points are 64 sectors = two words; w is even)
      nb_words = *reinterpret_cast<const uint2 *>(near_bits + w);
    }
  };
  nb_fill(lo);
  auto near_sectors = [&](uint32_t i0) -> unsigned long long {
    if (i0 >= hi) return 0ull;
    uint32_t t = (i0 - nb_first) / tstep;  // (workgroup-uniform)
    if (t >= 64u) nb_fill(i0), t = 0u;
    const uint32_t ts = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)nb_words.x, (int)ts) |
           ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)nb_words.y, (int)ts) << 32);
  };
  auto load_tile = [&](uint32_t i0, unsigned long long sect, float4 (&v)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t i = i0 + wave * 256 + u * 64 + lane;
      v[u] = make_float4(NAN, NAN, NAN, 0);
      if (((sect >> (16 * u + (lane >> 2))) & 1ull) && i < hi) v[u] = *reinterpret_cast<const float4 *>(M.pts + (size_t)i * M.stride_f);
    }
  };
  float4 v[4], nv[4];
  unsigned long long nib = near_sectors(lo), nnib = 0;
  load_tile(lo, nib, v);
  for (uint32_t i0 = lo; i0 < hi; i0 += tstep) {
    nnib = near_sectors(i0 + tstep);
    load_tile(i0 + tstep, nnib, nv);  // the next tile's loads are in flight while this one is tested
    if (MODE == 0 && i0 != lo && !(FX_GATHER_DIRECT && solo)) {  // flush when the next tile might not fit any more (workgroup-uniform decision)
      __syncthreads();
      const uint32_t staged = s_w[8];
      __syncthreads();
      if (staged > FX_GATHER_STAGE / 2) flush(staged);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!((nib >> (16 * u)) & 0xffffull)) continue;  // (wave-uniform: none of the load's sixteen sectors was loaded)
      const float x = v[u].x, y = v[u].y, z = v[u].z;
      const float rx = ((M.R[0] * x + M.R[1] * y) + M.R[2] * z) + 0.0f;
      const float ry = ((M.R[3] * x + M.R[4] * y) + M.R[5] * z) + 0.0f;
      const float rz = ((M.R[6] * x + M.R[7] * y) + M.R[8] * z) + 0.0f;
      // non-finite points are not part of the search surface; NaN fails every comparison
      const bool near = rx >= bx0 && rx <= bx1 && ry >= by0 && ry <= by1 && rz >= bz0 && rz <= bz1 &&
                        isfinite(rx) && isfinite(ry) && isfinite(rz);
      uint32_t info = 0;
      if (near) info = s_cell[cell_y(ry) * ncx + cell_x(rx)];
      const bool has = (info >> 20) != 0u;
      const unsigned long long m = __ballot(has);
      if (m) {  // wave-uniform
        if (has) {
          const uint32_t slot = qn + lanes_below(m);
          q_pt[slot] = make_float4(rx, ry, rz, __uint_as_float(i0 + wave * 256 + u * 64 + lane));
          q_info[slot] = info;
        }
        qn += (uint32_t)__popcll(m);
        if (qn > kQueue - 64) drain();
      }
    }
    nib = nnib;
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = nv[u];
  }
  drain();
  __syncthreads();
  if (MODE == 0) flush(s_w[8]);
  if (MODE == 1) {  // this slice's entries per keypoint
    uint32_t *cnt = B.gather_cnt + ((size_t)scan * gridDim.x + slice) * P.max_keypoints;
    for (uint32_t k = tid; k < K; k += NT) cnt[k] = s_kpos[k];
  }
  if (solo) {
    for (uint32_t k = tid; k < K; k += NT) B.s_cnt[row0 + k] = s_kpos[k];
    // RNG ordinals: keypoint k takes the x-axis number (keypoints before it that have a neighbour)
    __syncthreads();
    if (wave == 0 && !passes) {
      uint32_t base = 0;
      for (uint32_t k0 = 0; k0 < K; k0 += 64) {
        const uint32_t k = k0 + lane;
        const unsigned long long m = __ballot(k < K && s_knbr[k] != 0u);
        if (k < K) B.row_xa[row0 + k] = B.xaxis[base + lanes_below(m)];
        base += (uint32_t)__popcll(m);
      }
    }
  }
  } while (PASSES && (kb += MK) < K_all);
  if (solo) {
    if (tid == 0) B.ovf_cnt[scan] = s_w[9];
    if (passes) {  // the ordinals over all passes' flags
      wg_global_sync();
      if (wave == 0) {
        uint32_t base = 0;
        for (uint32_t k0 = 0; k0 < K_all; k0 += 64) {
          const uint32_t k = k0 + lane;
          const unsigned long long m = __ballot(k < K_all && B.kp_nbrs[(size_t)scan * P.max_keypoints + k] != 0u);
          if (k < K_all) B.row_xa[row_first + k] = B.xaxis[base + lanes_below(m)];
          base += (uint32_t)__popcll(m);
        }
      }
    }
  }
}
extern "C" __global__ __launch_bounds__(FX_GATHER_T) void k_gather(FxDevParams P, FxBuffers B, float box_margin) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  gather_body<FX_GATHER_T, false>(P, B, box_margin, smem);
}
extern "C" __global__ __launch_bounds__(FX_GATHER_WIDE_T) void k_gather_wide(FxDevParams P, FxBuffers B, float box_margin) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  gather_body<FX_GATHER_WIDE_T, false>(P, B, box_margin, smem);
}
// contexts of more than FX_GATHER_KCAP keypoints a scan: the wide workgroup, whatever the batch (its tables take most of a CU's LDS)
// batches of few big scans: the counted slices (gather_body's MODE), two launches
extern "C" __global__ __launch_bounds__(FX_GATHER_T) void k_gather_count(FxDevParams P, FxBuffers B, float box_margin) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  gather_body<FX_GATHER_T, false, 1>(P, B, box_margin, smem);
}
extern "C" __global__ __launch_bounds__(FX_GATHER_T) void k_gather_scatter(FxDevParams P, FxBuffers B, float box_margin) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  gather_body<FX_GATHER_T, false, 2>(P, B, box_margin, smem);
}
extern "C" __global__ __launch_bounds__(FX_GATHER_WIDE_T) void k_gather_passes(FxDevParams P, FxBuffers B, float box_margin) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  gather_body<FX_GATHER_WIDE_T, true>(P, B, box_margin, smem);
}

// ---------------------------------------------------------------- wavefront tier (runs inside k_desc_mid)
#ifndef FX_WAVE_CAP
#define FX_WAVE_CAP 192  // support points a wave row holds (multiples of 64; measured 128 / 192 / 256 / 320: k_desc_mid 0.118 / 0.099 / 0.119 / 0.153 ms)
#endif
#define FX_WAVE_WORDS (FX_WAVE_CAP * 8)
// Clears one descriptor row (1989 floats, 4-byte aligned) with `stride` lanes: 16-byte stores over its aligned body.
__device__ __forceinline__ void desc_zero_row(float *out, uint32_t lane, uint32_t stride) {
  const uint32_t head = (uint32_t)((16u - ((uintptr_t)out & 15u)) & 15u) / 4u;  // floats before the first 16-byte boundary
  const uint32_t n4 = (FX_DESC_FLOATS - head) / 4u, tail0 = head + 4u * n4;
  if (lane < head) out[lane] = 0.0f;
  if (lane < FX_DESC_FLOATS - tail0) out[tail0 + lane] = 0.0f;
  float4 *body = reinterpret_cast<float4 *>(out + head);
  for (uint32_t t = lane; t < n4; t += stride) body[t] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void desc_fill_nan(float *out, uint32_t lane, uint32_t stride) {
  for (uint32_t t = lane; t < FX_DESC_FLOATS; t += stride) out[t] = t < FX_DESC_BINS ? NAN : 0.0f;
}

// The 3DSC edge tables and weight table (206 floats) are copied to LDS once per workgroup: the bin
// search and the weight lookup of every neighbour then stay on chip.  Contains a barrier.
#define FX_TABLE_WORDS 208
__device__ __forceinline__ const FxScTables *tables_to_lds(const FxBuffers &B, uint32_t *dst) {
  const float *src = reinterpret_cast<const float *>(B.tables);
  for (uint32_t t = threadIdx.x; t < sizeof(FxScTables) / 4; t += blockDim.x) dst[t] = __float_as_uint(src[t]);
  __syncthreads();
  return reinterpret_cast<const FxScTables *>(dst);
}

// One wavefront per keypoint, for the rows in B.wave_desc (support sets of 65..FX_WAVE_CAP = 192 points).
// Angles in fp32; a keypoint with any neighbour whose angle lies within FX_FAST_EPS_DEG of a bin
// edge is evaluated exactly in place (sc3d_bin), so
// the result is the exact one either way.
template <bool FAST>
__device__ __forceinline__ void desc_wave_body(const FxDevParams &P, const FxBuffers &B, uint32_t batch,
                                               uint32_t *smem, uint32_t bid, uint32_t nblk) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const WithinR2 within(P.r2_density);
  uint32_t *base = smem + wave * FX_WAVE_WORDS;
  // per-wave LDS: support set as float4 (x, y, z, d2) + point index, unsorted (key, weight); the
  // sorted (key, weight) arrays reuse the support-set storage once the density counts are done
  float4 *sp = reinterpret_cast<float4 *>(base);                                                  // 4 * CAP words
  unsigned long long *nkey = reinterpret_cast<unsigned long long *>(base + 4 * FX_WAVE_CAP);     // 2 * CAP words
  float *nw = reinterpret_cast<float *>(base + 6 * FX_WAVE_CAP);
  uint32_t *sidx = base + 7 * FX_WAVE_CAP;
  unsigned long long *skey = reinterpret_cast<unsigned long long *>(base);  // aliases sp
  float *sw = reinterpret_cast<float *>(base + 2 * FX_WAVE_CAP);            // aliases sp
  const FxScTables *T = tables_to_lds(B, smem + FX_NWAVE * FX_WAVE_WORDS);

  uint32_t total = B.kp_offset[batch];
  if (total > P.max_total_kp) total = P.max_total_kp;
  const uint32_t n_items = B.counters[8];
  for (uint32_t it = bid * FX_NWAVE + wave; it < n_items; it += nblk * FX_NWAVE) {
    const uint32_t row = B.wave_desc[it];
    const uint2 rm = B.row_map[row];
    const uint32_t scan = rm.x, k = rm.y;
    // everything the row needs is fetched in one round trip: the list entries are loaded before the
    // list length is known (slots past the length hold stale data and are ignored)
    const uint32_t nS = B.s_cnt[row];
    const float4 kp = B.row_kp[row];
    const float2 xa = B.row_xa[row];
    float4 lv[FX_WAVE_CAP / 64];
#pragma unroll
    for (uint32_t u = 0; u < FX_WAVE_CAP / 64; ++u)
      if (lane + u * 64 < P.list_cap) lv[u] = B.s_pts[(size_t)row * P.list_cap + lane + u * 64];
    if (nS > FX_WAVE_CAP || nS > P.list_cap) continue;  // (never listed here: k_desc_group sends those rows to their tiers)
    float *out = B.desc + (size_t)row * FX_DESC_FLOATS;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (uint32_t u = 0; u < FX_WAVE_CAP / 64; ++u) {
      const uint32_t e = lane + u * 64;
      if (e < nS) {
        const float4 v = lv[u];
        sp[e] = make_float4(v.x, v.y, v.z, dist2(kp.x, kp.y, kp.z, v.x, v.y, v.z));
        sidx[e] = __float_as_uint(v.w);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    uint32_t nAll = 0, nM = 0;
    bool amb = false;
    for (uint32_t e0 = 0; e0 < nS; e0 += 64) {
      const uint32_t e = e0 + lane;
      const float4 b = e < nS ? sp[e] : make_float4(0, 0, 0, INFINITY);
      const float d2 = b.w;
      const bool nb = d2 < P.r2_search;
      nAll += (uint32_t)__popcll(__ballot(nb));
      const bool use = nb && !sc3d_is_origin(d2);
      unsigned long long key = 0;
      float w = 0.f;
      if (use) {
        float lut;
        const uint32_t bin = sc3d_bin<FAST>(kp, b.x, b.y, b.z, d2, xa, T, lut, amb);
        const uint32_t dens = within.count(sp, 0u, nS, b.x, b.y, b.z);  // support points within R/5 of this neighbour (itself included)
        w = (1.0f / (float)dens) * lut;
        key = sc3d_key(bin, d2, sidx[e]);
      }
      const unsigned long long um = __ballot(use);
      if (use) {
        const uint32_t pos = nM + lanes_below(um);
        nkey[pos] = key;
        nw[pos] = w;
      }
      nM += (uint32_t)__popcll(um);
    }
    if (lane == 0) B.kp_nbrs[(size_t)scan * P.max_keypoints + k] = nAll;
    if (nAll == 0) {  // no neighbours: NaN descriptor, no RNG draw (A.8-3)
      desc_fill_nan(out, lane, 64);
      continue;
    }
    // the output rows were cleared by k_desc_group; only the non-empty bins are written here.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // rank sort (keys are unique: they end in the point index)
    for (uint32_t e0 = 0; e0 < nM; e0 += 64) {
      const uint32_t e = e0 + lane;
      if (e < nM) {
        const unsigned long long key = nkey[e];
        uint32_t rank = 0;
#pragma unroll 8
        for (uint32_t q = 0; q < nM; ++q) rank += (nkey[q] < key) ? 1u : 0u;
        skey[rank] = key;
        sw[rank] = nw[e];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t e0 = 0; e0 < nM; e0 += 64) {
      const uint32_t e = e0 + lane;
      if (e >= nM) continue;
      const uint32_t bin = (uint32_t)(skey[e] >> 52);
      if (e > 0 && (uint32_t)(skey[e - 1] >> 52) == bin) continue;
      float acc = 0.0f;
      uint32_t q = e;
      do {
        acc += sw[q];
        ++q;
      } while (q < nM && (uint32_t)(skey[q] >> 52) == bin);
      out[bin] = acc;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------- k_desc_group
// Most keypoints have a small support set (median < 10 points on the VLP-16 scenes, 90 % < 64):
// a whole wavefront per keypoint leaves its lanes idle and pays the per-keypoint latency 64-wide.
// Here a wavefront works on FX_GROUPS keypoints at once, FX_GLANES lanes each (fp32 angles, same
// exactness contract as the wavefront tier).  Rows with more than FX_GROUP_CAP support points go to the
// wave list, rows with an angle near a bin edge to the exact list.
#ifndef FX_GLANES
#define FX_GLANES 16
#endif
#define FX_GROUPS (64 / FX_GLANES)
#ifndef FX_GROUP_CAP
#define FX_GROUP_CAP 64
#endif
#define FX_GROUP_WORDS (FX_GROUP_CAP * 8 + 8)  // per group: support float4, keys, weights, indices + 8 counters
#ifndef FX_GROUP_UNROLL
#define FX_GROUP_UNROLL 8
#endif
#ifndef FX_GROUP_OCC
#define FX_GROUP_OCC 4
#endif
extern "C" __global__ __launch_bounds__(FX_WG, FX_GROUP_OCC) void k_desc_group(FxDevParams P, FxBuffers B, uint32_t batch) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t gl = lane % FX_GLANES, g = lane / FX_GLANES;
  const WithinR2 within(P.r2_density);
  uint32_t *base = smem + (wave * FX_GROUPS + g) * FX_GROUP_WORDS;
  float4 *sp = reinterpret_cast<float4 *>(base);                                                   // 4 * CAP words
  unsigned long long *nkey = reinterpret_cast<unsigned long long *>(base + 4 * FX_GROUP_CAP);      // 2 * CAP
  float *nw = reinterpret_cast<float *>(base + 6 * FX_GROUP_CAP);
  uint32_t *sidx = base + 7 * FX_GROUP_CAP;
  uint32_t *cnt = base + 8 * FX_GROUP_CAP;  // [0] neighbours, [1] binned neighbours, [2] ambiguous
  unsigned long long *skey = reinterpret_cast<unsigned long long *>(base);  // aliases sp
  float *sw = reinterpret_cast<float *>(base + 2 * FX_GROUP_CAP);           // aliases sp
  const FxScTables *T = tables_to_lds(B, smem + FX_NWAVE * FX_GROUPS * FX_GROUP_WORDS);

  uint32_t total = B.kp_offset[batch];
  if (total > P.max_total_kp) total = P.max_total_kp;
  const uint32_t stride = gridDim.x * FX_NWAVE * FX_GROUPS;
  // Everything a row needs is fetched in one round trip — keypoint, x-axis, the list entries before the list length is known
  // (slots past the length hold stale data and are ignored), and what the row holds from last time (its bin count and all 64
  // slots of its bin list, four per lane) — and a TRIP AHEAD: the next trip's loads go out before this trip's last step (few
  // registers are live there) and are taken at the top of the next trip.  (With one workgroup a CU a wavefront has a SIMD to
  // itself, and every trip began with that round trip.)
  static_assert(FX_GROUP_CAP == 4 * FX_GLANES, "a lane fetches four slots of the row's bin list");
  uint2 f_rm = make_uint2(0u, 0u), f_pb = make_uint2(0u, 0u);
  uint32_t f_nS = 0, f_nb = 0;
  float4 f_kp = make_float4(0, 0, 0, 0), f_lv[FX_GROUP_CAP / FX_GLANES];
  float2 f_xa = make_float2(1.f, 0.f);
  auto fetch = [&](uint32_t row) {
    // (every value is defined anew on both paths: nothing of the previous fetch stays live across a trip)
    f_rm = make_uint2(0u, 0u), f_pb = make_uint2(0u, 0u), f_nS = 0, f_nb = 0, f_kp = make_float4(0, 0, 0, 0), f_xa = make_float2(1.f, 0.f);
#pragma unroll
    for (uint32_t u = 0; u < FX_GROUP_CAP / FX_GLANES; ++u) f_lv[u] = make_float4(0, 0, 0, 0);
    if (row < total) {
      f_rm = B.row_map[row];
      f_nS = B.s_cnt[row];
      f_kp = B.row_kp[row];
      f_xa = B.row_xa[row];
#pragma unroll
      for (uint32_t u = 0; u < FX_GROUP_CAP / FX_GLANES; ++u)
        if (gl + u * FX_GLANES < P.list_cap) f_lv[u] = B.s_pts[(size_t)row * P.list_cap + gl + u * FX_GLANES];
      f_nb = B.desc_nbins[row];
      f_pb = *reinterpret_cast<const uint2 *>(B.desc_bins + (size_t)row * FX_GROUP_CAP + 4u * gl);
    }
  };
  fetch((blockIdx.x * FX_NWAVE + wave) * FX_GROUPS + g);
  // all groups of a wavefront make the same number of trips (wave-level fences inside)
  for (uint32_t r0 = (blockIdx.x * FX_NWAVE + wave) * FX_GROUPS; r0 < total; r0 += stride) {
    const uint32_t row = r0 + g;
    bool live = row < total;
    uint32_t scan = 0, k = 0, nS = 0;
    float4 kp = make_float4(0, 0, 0, 0);
    float2 xa = make_float2(1.f, 0.f);
    float4 lv[FX_GROUP_CAP / FX_GLANES];
    uint32_t nb_prev = 0;
    uint2 pb = make_uint2(0u, 0u);
    if (live) {
      scan = f_rm.x, k = f_rm.y, nS = f_nS, kp = f_kp, xa = f_xa, nb_prev = f_nb, pb = f_pb;
#pragma unroll
      for (uint32_t u = 0; u < FX_GROUP_CAP / FX_GLANES; ++u) lv[u] = f_lv[u];
    }
    float *out = B.desc + (size_t)row * FX_DESC_FLOATS;
    // Every descriptor row is cleared here, whichever tier ends up computing it (the keypoint kernels write
    // non-empty bins only): this kernel sees every row once and waits on latencies with its memory pipe idle.
    // A row this kernel itself filled last time — nine in ten — is cleared by zeroing the bins it wrote then (recorded in
    // desc_bins): some tens of 4-byte stores instead of 7956 bytes.  (The descriptor rows are half of the batch's
    // algorithmic bytes and all of them writes; the dirty lines they leave in L2 / MALL are written back under the next
    // kernels — k_prep ran 0.17 ms behind a batch's row clears and 0.13 ms without.)
    uint16_t *my_bins = B.desc_bins + (size_t)row * FX_GROUP_CAP;
    if (live) {
      if (nb_prev > FX_GROUP_CAP) {
        desc_zero_row(out, gl, FX_GLANES);
      } else {
        const uint32_t t0 = 4u * gl;
        if (t0 + 0u < nb_prev) out[pb.x & 0xffffu] = 0.0f;
        if (t0 + 1u < nb_prev) out[pb.x >> 16] = 0.0f;
        if (t0 + 2u < nb_prev) out[pb.y & 0xffffu] = 0.0f;
        if (t0 + 3u < nb_prev) out[pb.y >> 16] = 0.0f;
      }
    }
    // too long for a group: wavefront rows (<= FX_WAVE_CAP support points), list rows (<= dense_min), and the dense tier beyond
    // that — also every row whose list overflowed its list_cap slots (the rest of it sits in the scan's overflow region)
    const bool too_long = live && (nS > FX_GROUP_CAP || nS > P.list_cap);
    uint32_t dense_fail = 0;
    {
      // The rows a wavefront hands on draw their list positions together: one atomic per list and wavefront instead of one
      // (dense rows: three) per row — on config 3 half of the 40 000 rows of a batch are handed on, and their atomics, all on
      // two cache lines of counters, went through the L2 one after the other.
      const bool lead = too_long && gl == 0;
      const bool to_dense = lead && (nS > P.dense_min || nS > P.list_cap);
      const bool to_list = lead && !to_dense && nS > FX_WAVE_CAP, to_wave = lead && !to_dense && !to_list;
      // positions [base, base + n) of counter c for the lanes of m (wave-uniform call); returns this lane's
      auto draw = [&](unsigned long long m, bool mine, uint32_t c, uint32_t amount, uint32_t before) -> uint32_t {
        if (!m) return 0u;
        const uint32_t first = (uint32_t)__ffsll((long long)m) - 1u;
        uint32_t b0 = 0;
        if (lane == first) b0 = atomicAdd(&B.counters[c], amount);
        b0 = (uint32_t)__shfl((int)b0, (int)first, 64);
        return mine ? b0 + before : 0u;
      };
      const unsigned long long m_list = __ballot(to_list), m_wave = __ballot(to_wave), m_dense = __ballot(to_dense);
      const uint32_t p_list = draw(m_list, to_list, 4u, (uint32_t)__popcll(m_list), lanes_below(m_list));
      const uint32_t p_wave = draw(m_wave, to_wave, 8u, (uint32_t)__popcll(m_wave), lanes_below(m_wave));
      if (to_list) B.list_desc[p_list] = row;
      if (to_wave) B.wave_desc[p_wave] = row;
      if (m_dense) {  // (wave-uniform)
        // a slot in the dense-row list and nS entries of the sorted pool (k_dense_sort fills them)
        uint32_t pre = 0, tot = 0;
#pragma unroll
        for (uint32_t g2 = 0; g2 < FX_GROUPS; ++g2) {
          const uint32_t v = (uint32_t)__shfl((int)(to_dense ? nS : 0u), (int)(g2 * FX_GLANES), 64);
          pre += g2 < g ? v : 0u;
          tot += v;
        }
        const uint32_t slot = draw(m_dense, to_dense, 6u, (uint32_t)__popcll(m_dense), lanes_below(m_dense));
        const uint32_t off = draw(m_dense, to_dense, 13u, tot, pre);
        // the tier's kernels take the rows largest first (four size classes), so that the big ones do not end up alone at the tail
        const uint32_t cls = dense_class(nS);
        uint32_t p_cls = 0;
#pragma unroll
        for (uint32_t c2 = 0; c2 < 4u; ++c2) {
          const bool in = to_dense && slot < P.max_dense_rows && cls == c2;  // (a row without a slot is in no class list)
          const unsigned long long m_c = __ballot(in);
          const uint32_t v = draw(m_c, in, dense_class_counter(c2), (uint32_t)__popcll(m_c), lanes_below(m_c));
          p_cls = in ? v : p_cls;
        }
        if (to_dense) {
          const bool ok = slot < P.max_dense_rows && off <= P.dense_cap && nS <= P.dense_cap - off;
          if (slot < P.max_dense_rows) {
            B.dense_rows[slot] = ok ? row : FX_NONE;
            B.dense_off[slot] = off;
            B.dense_order[cls * P.max_dense_rows + p_cls] = slot;
          }
          if (!ok) {  // pools exhausted (limits.max_dense_points): flagged, never silent
            atomicOr(&B.flags[scan], FX_FLAG_NBR_OVERFLOW);
            B.kp_nbrs[(size_t)scan * P.max_keypoints + k] = FX_NONE;
            dense_fail = 1;
          }
        }
      }
    }
    dense_fail = (uint32_t)__shfl((int)dense_fail, (int)(g * FX_GLANES), 64);
    if (dense_fail) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (the clearing stores first)
      desc_fill_nan(out, gl, FX_GLANES);
    }
    if (too_long && gl == 0) B.desc_nbins[row] = FX_ROW_DIRTY;  // written by another tier: cleared whole next time
    if (too_long) live = false;
    if (!live) nS = 0;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (uint32_t u = 0; u < FX_GROUP_CAP / FX_GLANES; ++u) {
      const uint32_t e = gl + u * FX_GLANES;
      if (e < nS) {
        const float4 v = lv[u];
        sp[e] = make_float4(v.x, v.y, v.z, dist2(kp.x, kp.y, kp.z, v.x, v.y, v.z));
        sidx[e] = __float_as_uint(v.w);
      }
    }
    if (gl < 3) cnt[gl] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t e = gl; e < nS; e += FX_GLANES) {
      const float4 b = sp[e];
      const float d2 = b.w;
      if (!(d2 < P.r2_search)) continue;
      atomicAdd(&cnt[0], 1u);
      if (sc3d_is_origin(d2)) continue;
      float lut;
      bool amb = false;
      const uint32_t bin = sc3d_bin<true>(kp, b.x, b.y, b.z, d2, xa, T, lut, amb);
      const uint32_t dens = within.count(sp, 0u, nS, b.x, b.y, b.z);  // support points within R/5 of this neighbour (itself included)
      const uint32_t pos = atomicAdd(&cnt[1], 1u);
      nkey[pos] = sc3d_key(bin, d2, sidx[e]);
      nw[pos] = (1.0f / (float)dens) * lut;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t nAll = live ? cnt[0] : 0u;
    uint32_t nM = live ? cnt[1] : 0u;
    // (s_waitcnt vmcnt(0): the clearing stores are acknowledged before anything else is written to the row; a wider
    //  scope would write the whole L2 back)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (live) {
      if (gl == 0) B.kp_nbrs[(size_t)scan * P.max_keypoints + k] = nAll;
      if (nAll == 0) {  // no neighbours: NaN descriptor, no RNG draw (A.8-3)
        desc_fill_nan(out, gl, FX_GLANES);
        nM = 0;
        if (gl == 0) B.desc_nbins[row] = FX_ROW_DIRTY;
      }
    }
    if (gl == 0) cnt[2] = 0;  // bins written to the row
    // rank sort (keys are unique: they end in the point index); the sorted arrays reuse the support storage
    unsigned long long my_key[FX_GROUP_CAP / FX_GLANES];
    float my_w[FX_GROUP_CAP / FX_GLANES];
    uint32_t my_rank[FX_GROUP_CAP / FX_GLANES];
#pragma unroll
    for (uint32_t u = 0; u < FX_GROUP_CAP / FX_GLANES; ++u) {
      const uint32_t e = gl + u * FX_GLANES;
      my_rank[u] = FX_NONE;
      if (e < nM) {
        my_key[u] = nkey[e];
        my_w[u] = nw[e];
        uint32_t rank = 0;
#pragma unroll FX_GROUP_UNROLL
        for (uint32_t q = 0; q < nM; ++q) rank += (nkey[q] < my_key[u]) ? 1u : 0u;
        my_rank[u] = rank;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (uint32_t u = 0; u < FX_GROUP_CAP / FX_GLANES; ++u)
      if (my_rank[u] != FX_NONE) {
        skey[my_rank[u]] = my_key[u];
        sw[my_rank[u]] = my_w[u];
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    fetch(r0 + stride + g);  // (the next trip's rows: see the top)
    // one lane per bin run adds its weights in sorted order (the row was cleared at the top of this trip)
    for (uint32_t e = gl; e < nM; e += FX_GLANES) {
      const uint32_t bin = (uint32_t)(skey[e] >> 52);
      if (e > 0 && (uint32_t)(skey[e - 1] >> 52) == bin) continue;
      float acc = 0.0f;
      uint32_t q = e;
      do {
        acc += sw[q];
        ++q;
      } while (q < nM && (uint32_t)(skey[q] >> 52) == bin);
      out[bin] = acc;
      my_bins[atomicAdd(&cnt[2], 1u)] = (uint16_t)bin;  // (at most nM <= FX_GROUP_CAP of them)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (live && nAll != 0 && gl == 0) B.desc_nbins[row] = cnt[2];
  }
}

// ---------------------------------------------------------------- workgroup tiers
struct DescLds {
  unsigned long long *nkey;
  float4 *sp;  // support set (x, y, z, d2)
  float *nw, *img;
  uint32_t *sidx, *s_w;
};
#define FX_DESC_WORDS_PER_POINT 8
#define FX_DGRID3 12  // the list tiers' xyz density grid (12^3 cell words + the 3DSC tables fit the image)
__device__ __forceinline__ DescLds desc_carve(uint32_t *smem, uint32_t cap) {
  DescLds L;
  L.s_w = smem;  // 16 words
  uint32_t *p = smem + 16;
  L.sp = (float4 *)p, p += 4 * cap;                // 16-byte aligned: 16 words in front
  L.nkey = (unsigned long long *)p, p += 2 * cap;
  L.nw = (float *)p, p += cap;
  L.sidx = p, p += cap;
  L.img = (float *)p;  // FX_DESC_BINS floats
  return L;
}

__device__ __forceinline__ DescLds desc_carve_gs(uint32_t *smem, uint32_t *gs, uint32_t cap) {
  DescLds L;
  L.s_w = smem;  // 16 words
  L.img = (float *)(smem + 16);  // FX_DESC_BINS floats
  uint32_t *p = gs;  // (16-byte aligned)
  L.sp = (float4 *)p, p += 4 * (size_t)cap;
  L.nkey = (unsigned long long *)p, p += 2 * (size_t)cap;
  L.nw = (float *)p, p += cap;
  L.sidx = p;
  return L;
}

// One keypoint by a whole workgroup.  from_list: the support set comes from k_gather's list;
// otherwise it is re-gathered from the scan (lists that overflowed P.list_cap).
// Returns false if the support set does not fit `cap` (only possible when !from_list).
// GS (dense_slow_loop): the per-point arrays live in the scratch region `gs` of HBM (8 words a point: any support set), the
// scratch words and the image stay in LDS.
template <bool FAST, int NT, bool GS = false>
__device__ __forceinline__ bool desc_body(const FxDevParams &P, const FxBuffers &B, uint32_t row, uint32_t scan, uint32_t k,
                          uint32_t cap, uint32_t *smem, bool from_list, uint32_t *gs = nullptr) {
  DescLds L = GS ? desc_carve_gs(smem, gs, cap) : desc_carve(smem, cap);
  FX_STAMP_INIT(B.stamps && FAST ? B.stamps + 48 : nullptr);
  const uint32_t tid = threadIdx.x;
  const WithinR2 within(P.r2_density);
  const FxScanMeta M = B.meta[scan];
  const float4 kp = B.keypoints[(size_t)scan * P.max_keypoints + k];
  float *out = B.desc + (size_t)row * FX_DESC_FLOATS;
  // A list row's entries (up to four a thread: lists of up to 4 NT entries), its length and its x-axis are fetched HERE, with
  // the keypoint, before the length is known (slots past it hold stale data and are ignored): fetched where they are used
  // they were three more global round trips in a row's chain — keypoint, list (counting pass), list (fill pass), x-axis.
  constexpr uint32_t kPre = 4;
  const bool pre = from_list && !GS;
  float4 lv[kPre];
  uint32_t nS_pre = 0;
  float2 xa_pre = make_float2(1.f, 0.f);
  if (pre) {
    nS_pre = B.s_cnt[row];
    xa_pre = B.row_xa[row];
#pragma unroll
    for (uint32_t u = 0; u < kPre; ++u)
      if (tid + u * NT < P.list_cap) lv[u] = B.s_pts[(size_t)row * P.list_cap + tid + u * NT];
  }

  if (tid < 4) L.s_w[tid] = 0;  // 0: support count, 1: binned neighbours, 2: all neighbours
  // the 3DSC tables ride in the image until the bins are known (the image is cleared after that)
  uint32_t *tl = reinterpret_cast<uint32_t *>(L.img) + FX_DGRID3 * FX_DGRID3 * FX_DGRID3;  // behind the cell table
  for (uint32_t t = tid; t < sizeof(FxScTables) / 4; t += NT) tl[t] = __float_as_uint(reinterpret_cast<const float *>(B.tables)[t]);
  wg_sync<GS>();
  uint32_t nS;
  // Density grid: a list-fed support set is stored sorted by cell (12 x 12 x 12 cells of width >= R/5
  // over the support sphere's box), so the density query of a neighbour only scans the nine rows of
  // three x-cells around it instead of the whole set.  One word per cell — first its count, then its
  // start, then (after the fill) its end, which is the next cell's start — borrows the image.
  constexpr uint32_t G = FX_DGRID3, kCells = G * G * G;
  uint32_t *cell_end = reinterpret_cast<uint32_t *>(L.img);  // [kCells]
  const float r_sup = sqrtf(P.r2_support);
  const float cell_w = fmaxf(sqrtf(P.r2_density) * 1.001f, 2.0f * r_sup / (float)(G - 1) * 1.0001f);
  const float inv_cw = 1.0f / cell_w, gx0 = kp.x - r_sup, gy0 = kp.y - r_sup, gz0 = kp.z - r_sup;
  auto cell_1d = [&](float v, float v0) { return (uint32_t)min(max((int)floorf((v - v0) * inv_cw), 0), (int)G - 1); };
  auto cell_of = [&](float x, float y, float z) { return (cell_1d(z, gz0) * G + cell_1d(y, gy0)) * G + cell_1d(x, gx0); };
  bool grid = false;
  if (from_list) {
    nS = pre ? nS_pre : B.s_cnt[row];
    grid = nS <= cap;
    const bool in_regs = pre && nS <= kPre * NT;  // (workgroup-uniform)
    if (grid) {
      const float4 *lst = B.s_pts + (size_t)row * P.list_cap;
      for (uint32_t t = tid; t < kCells; t += NT) cell_end[t] = 0;
      wg_sync<GS>();
      if (in_regs) {
#pragma unroll
        for (uint32_t u = 0; u < kPre; ++u)
          if (tid + u * NT < nS) atomicAdd(&cell_end[cell_of(lv[u].x, lv[u].y, lv[u].z)], 1u);
      } else {
        for (uint32_t e = tid; e < nS; e += NT) {
          const float4 v = lst[e];
          atomicAdd(&cell_end[cell_of(v.x, v.y, v.z)], 1u);
        }
      }
      wg_sync<GS>();
      if (tid < 64) {  // counts -> exclusive starts, in place, by one wavefront
        constexpr uint32_t per = (kCells + 63) / 64;
        uint32_t sum = 0;
        for (uint32_t u = 0; u < per; ++u) {
          const uint32_t ci = tid * per + u;
          sum += ci < kCells ? cell_end[ci] : 0u;
        }
        uint32_t incl = sum;
        incl = wave_incl_scan(incl);
        uint32_t run = incl - sum;
        for (uint32_t u = 0; u < per; ++u) {
          const uint32_t ci = tid * per + u;
          if (ci < kCells) {
            const uint32_t c = cell_end[ci];
            cell_end[ci] = run;
            run += c;
          }
        }
      }
      wg_sync<GS>();
      if (in_regs) {
#pragma unroll
        for (uint32_t u = 0; u < kPre; ++u)
          if (tid + u * NT < nS) {
            const float4 v = lv[u];
            const uint32_t slot = atomicAdd(&cell_end[cell_of(v.x, v.y, v.z)], 1u);
            L.sp[slot] = make_float4(v.x, v.y, v.z, dist2(kp.x, kp.y, kp.z, v.x, v.y, v.z));
            L.sidx[slot] = __float_as_uint(v.w);
          }
      } else {
        for (uint32_t e = tid; e < nS; e += NT) {  // second read of the list (cache-resident): each entry to its cell
          const float4 v = lst[e];
          const uint32_t slot = atomicAdd(&cell_end[cell_of(v.x, v.y, v.z)], 1u);
          L.sp[slot] = make_float4(v.x, v.y, v.z, dist2(kp.x, kp.y, kp.z, v.x, v.y, v.z));
          L.sidx[slot] = __float_as_uint(v.w);
        }
      }
    } else {
      for (uint32_t e = tid; e < nS; e += NT) {
        const float4 v = B.s_pts[(size_t)row * P.list_cap + e];
        L.sp[e] = make_float4(v.x, v.y, v.z, dist2(kp.x, kp.y, kp.z, v.x, v.y, v.z));
        L.sidx[e] = __float_as_uint(v.w);
      }
    }
    wg_sync<GS>();
  } else {
    const uint32_t n = M.n;
    for (uint32_t i0 = 0; i0 < n; i0 += NT * 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t i = i0 + u * NT + tid;
        v[u] = i < n ? *reinterpret_cast<const float4 *>(M.pts + (size_t)i * M.stride_f) : make_float4(NAN, NAN, NAN, 0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float x = v[u].x, y = v[u].y, z = v[u].z;
        const float rx = ((M.R[0] * x + M.R[1] * y) + M.R[2] * z) + 0.0f;
        const float ry = ((M.R[3] * x + M.R[4] * y) + M.R[5] * z) + 0.0f;
        const float rz = ((M.R[6] * x + M.R[7] * y) + M.R[8] * z) + 0.0f;
        const float d = dist2(kp.x, kp.y, kp.z, rx, ry, rz);
        if (d < P.r2_support && isfinite(rx) && isfinite(ry) && isfinite(rz)) {
          const uint32_t pos = atomicAdd(&L.s_w[0], 1u);
          if (pos < cap) {
            L.sp[pos] = make_float4(rx, ry, rz, d);
            L.sidx[pos] = i0 + u * NT + tid;
          }
        }
      }
    }
    wg_sync<GS>();
    nS = L.s_w[0];
    if (nS > cap) return false;  // the caller hands the keypoint to the spill tier
  }
  FX_STAMP(1);

  const FxScTables *T = reinterpret_cast<const FxScTables *>(tl);
  const float2 xa = pre ? xa_pre : B.row_xa[row];
  // ---- neighbours (d2 < R^2, not the keypoint itself) packed densely: nlist[m] = support position.
  //      The list lives in the weight array and the density counters in the key array until the
  //      per-neighbour pass below overwrites slot m with the real key and weight.
  uint32_t *nlist = reinterpret_cast<uint32_t *>(L.nw);
  uint32_t *dens = reinterpret_cast<uint32_t *>(L.nkey);  // dens[2 * m]
  // (a stable compaction: the list keeps the support set's order, which for a list-fed set is its cell order —
  //  the lanes of a wavefront then hold neighbours of the same few cells, whose density queries scan the same rows:
  //  equal trip counts and LDS broadcast reads instead of a wavefront paying every row's longest lane)
  {
    uint32_t *scratch = reinterpret_cast<uint32_t *>(L.img) + FX_DESC_BINS - 32;  // (behind the cell table and the 3DSC tables)
    uint32_t n_use = 0;
    for (uint32_t e0 = 0; e0 < nS; e0 += NT) {
      const uint32_t e = e0 + tid;
      const float d2 = e < nS ? L.sp[e].w : INFINITY;
      const bool nb = d2 < P.r2_search, use = nb && !sc3d_is_origin(d2);
      const unsigned long long m_nb = __ballot(nb);
      if ((tid & 63u) == 0 && m_nb) atomicAdd(&L.s_w[2], (uint32_t)__popcll(m_nb));
      uint32_t tot;
      const uint32_t r = block_rank<NT>(use, scratch, tot);
      if (use) {
        nlist[n_use + r] = e;
        dens[2 * (n_use + r)] = 0u;
      }
      n_use += tot;
    }
    if (tid == 0) L.s_w[1] = n_use;
  }
  wg_sync<GS>();
  FX_STAMP(6);
  {
    // ---- local point density = support points within R/5 of the neighbour (itself included): every
    //      neighbour's count is split over `parts` lanes so that the whole workgroup is busy
    const uint32_t nMq = L.s_w[1];
    uint32_t parts = 1;
    while (parts < (grid ? 8u : 32u) && nMq * parts * 2 <= (uint32_t)NT) parts <<= 1;
    const uint32_t chunk = (nS + parts - 1) / parts;
    for (uint32_t t = tid; t < nMq * parts; t += NT) {
      const uint32_t m = t / parts, part = t % parts;
      const float4 bq = L.sp[nlist[m]];
      uint32_t c = 0;
      if (grid) {
        const uint32_t cx = cell_1d(bq.x, gx0), cy = cell_1d(bq.y, gy0), cz = cell_1d(bq.z, gz0);
        const uint32_t xa0 = cx > 0 ? cx - 1 : 0, xa1 = min(cx + 1, G - 1);
        for (uint32_t rowi = part; rowi < 9; rowi += parts) {  // the cells of an x-row are stored back to back
          const uint32_t yy = cy + rowi % 3u - 1u, zz = cz + rowi / 3u - 1u;
          if (yy >= G || zz >= G) continue;  // (also the wrapped -1)
          const uint32_t c0 = (zz * G + yy) * G + xa0, c1 = (zz * G + yy) * G + xa1;
          const uint32_t q0 = c0 ? cell_end[c0 - 1] : 0u, q1 = cell_end[c1];
          c += within.count(L.sp, q0, q1, bq.x, bq.y, bq.z);
        }
      } else {
        const uint32_t q0 = part * chunk, q1 = min(q0 + chunk, nS);
        c += within.count(L.sp, q0, q1, bq.x, bq.y, bq.z);
      }
      if (c) atomicAdd(&dens[2 * m], c);
    }
    wg_sync<GS>();
    FX_STAMP(7);
    // ---- bins and weights, one neighbour per lane
    for (uint32_t m = tid; m < nMq; m += NT) {
      const uint32_t e = nlist[m];
      const float4 bq = L.sp[e];
      float lut;
      bool amb = false;
      const uint32_t bin = sc3d_bin<FAST>(kp, bq.x, bq.y, bq.z, bq.w, xa, T, lut, amb);
      const uint32_t d = dens[2 * m];
      L.nkey[m] = sc3d_key(bin, bq.w, L.sidx[e]);
      L.nw[m] = (1.0f / (float)d) * lut;
    }
  }
  wg_sync<GS>();
  FX_STAMP(2);
  for (uint32_t t = tid; t < FX_DESC_BINS; t += NT) L.img[t] = 0.0f;  // (cell table and 3DSC tables are done with)
  const uint32_t nM = L.s_w[1], nAll = L.s_w[2];
  if (tid == 0) {
    FX_COUNT(12, 1);
    FX_COUNT(13, nS);
    FX_COUNT(14, nM);
  }
  if (tid == 0) B.kp_nbrs[(size_t)scan * P.max_keypoints + k] = nAll;
  if (nAll == 0) {  // no neighbours: NaN descriptor, no RNG draw (A.8-3)
    desc_fill_nan(out, tid, NT);
    wg_sync<GS>();
    return true;
  }
  // ---- PCL adds a bin's contributions in the order of its sorted radius search, (d2, index) ascending: counting sort by
  //      bin (the image doubles as the bin table), ranking inside the bin (keys are unique: they end in the point index),
  //      then one lane per bin adds the bin's weights in order.  (A ranking of all keys against all keys — 160 000
  //      comparisons for 400 neighbours — was most of a list row's instructions; a bitonic network beyond 512.)
  uint32_t *bin_end = reinterpret_cast<uint32_t *>(L.img);  // [1980], zero (cleared above)
  unsigned long long *skey = reinterpret_cast<unsigned long long *>(L.sp);  // the support set is done with: [cap] keys by bin,
  float *sw = reinterpret_cast<float *>(L.sp) + 2 * cap;                    // [cap] their weights,
  float *sorted_w = reinterpret_cast<float *>(L.sp) + 3 * cap;              // [cap] the weights in (bin, d2, index) order
  wg_sync<GS>();
  for (uint32_t m = tid; m < nM; m += NT) atomicAdd(&bin_end[(uint32_t)(L.nkey[m] >> 52)], 1u);
  wg_sync<GS>();
  if (tid < 64) {  // counts -> exclusive starts, in place, by one wavefront
    constexpr uint32_t per = (FX_DESC_BINS + 63) / 64;
    uint32_t sum = 0;
    for (uint32_t u = 0; u < per; ++u) {
      const uint32_t bi = tid * per + u;
      sum += bi < FX_DESC_BINS ? bin_end[bi] : 0u;
    }
    uint32_t incl = sum;
    incl = wave_incl_scan(incl);
    uint32_t run = incl - sum;
    for (uint32_t u = 0; u < per; ++u) {
      const uint32_t bi = tid * per + u;
      if (bi < FX_DESC_BINS) {
        const uint32_t c = bin_end[bi];
        bin_end[bi] = run;
        run += c;
      }
    }
  }
  wg_sync<GS>();
  for (uint32_t m = tid; m < nM; m += NT) {  // (the fill turns a bin's start into its end = the next bin's start)
    const unsigned long long key = L.nkey[m];
    const uint32_t pos = atomicAdd(&bin_end[(uint32_t)(key >> 52)], 1u);
    skey[pos] = key;
    sw[pos] = L.nw[m];
  }
  wg_sync<GS>();
  for (uint32_t p = tid; p < nM; p += NT) {
    const unsigned long long key = skey[p];
    const uint32_t bin = (uint32_t)(key >> 52);
    const uint32_t s0 = bin ? bin_end[bin - 1u] : 0u, s1 = bin_end[bin];
    uint32_t rank = 0;
    for (uint32_t q = s0; q < s1; ++q) rank += skey[q] < key ? 1u : 0u;
    sorted_w[s0 + rank] = sw[p];
  }
  wg_sync<GS>();
  FX_STAMP(3);
  // ---- sequential fp32 accumulation per bin, in sorted order; the row was cleared by k_desc_group (rf stays zero)
  for (uint32_t bin = tid; bin < FX_DESC_BINS; bin += NT) {
    const uint32_t s0 = bin ? bin_end[bin - 1u] : 0u, s1 = bin_end[bin];
    if (s0 == s1) continue;
    float acc = 0.0f;
    for (uint32_t q = s0; q < s1; ++q) acc += sorted_w[q];
    out[bin] = acc;
  }
  wg_sync<GS>();
  FX_STAMP(4);
  FX_STAMP(5);
  return true;
}

// List rows (FX_WAVE_CAP + 1 .. cap support points): one keypoint per workgroup at a time, from k_gather's list, fp32 angles
// (exact in place next to a bin edge).
template <bool FAST, int NT>
__device__ __forceinline__ void desc_wg_loop(const FxDevParams &P, const FxBuffers &B, uint32_t batch, uint32_t cap,
                                             uint32_t *smem, uint32_t bid, uint32_t nblk) {
  const uint32_t n_items = B.counters[4];
  for (uint32_t i = bid; i < n_items; i += nblk) {
    const uint32_t row = B.list_desc[i];
    const uint2 rm = B.row_map[row];
    const uint32_t nS = B.s_cnt[row];
    if (nS > P.list_cap || nS > cap) continue;  // (never listed here: k_desc_group sends those rows to the dense tier)
    desc_body<FAST, NT>(P, B, row, rm.x, rm.y, cap, smem, true);
    __syncthreads();
  }
}
// The fp32 pass runs 256-thread workgroups, four keypoints per CU at a time (most phases of a keypoint are
// latency chains that leave lanes idle, so concurrency beats width).
#define FX_DESC_WG_FAST_T 256
// The dense tier's rows WITHOUT its launches (four in round 5, three since k_dense_finish takes every row): a batch whose predecessors had no dense row (every VLP-16-class batch) gets
// a handful of workgroups of k_desc_mid's launch instead — 256 threads and 8 KB of LDS place anywhere, where the tier's own kernels'
// workgroups each wait for a large LDS slot behind the other batches' kernels (+3.7 % on the headline with them gone,
// profiles/r05_experiments.md).  A row that does turn up is computed here by the list tier's body on scratch in HBM, its
// support set re-gathered from the scan: slower, the same result (each row counts its neighbours' densities itself: what the
// tier's per-scan cache shares).  Which of the two runs never changes a result — the host's memory of earlier batches chooses
// speed only.
#define FX_DSLOW_T 256
__device__ __forceinline__ void dense_slow_loop(const FxDevParams &P, const FxBuffers &B, uint32_t *smem, uint32_t bid, uint32_t nblk) {
  const uint32_t n_rows = min(B.counters[6], P.max_dense_rows);
  uint32_t *gs = B.gsd_pool + (size_t)bid * P.gsd_words;
  for (uint32_t slot = bid; slot < n_rows; slot += nblk) {
    const uint32_t row = B.dense_rows[slot];
    if (row == FX_NONE) continue;  // (no room in the pools: flagged by k_desc_group, as for the fast kernels)
    const uint2 rm = B.row_map[row];
    desc_body<true, FX_DSLOW_T, true>(P, B, row, rm.x, rm.y, P.max_points, smem, false, gs);
    wg_global_sync();
  }
}
// Both middle tiers in one launch: the first n_wg workgroups take list rows (193..cap support points, one keypoint per
// workgroup at a time), the others take wave rows (65..192, one keypoint per wavefront).  Neither tier fills the chip
// alone (1600 and 900 of the 8192 wave slots on the VLP-16 bench) and neither depends on the other: one after the
// other they cost 0.135 + 0.075 ms, together about the longer of the two.  The last n_dslow workgroups (when the dense
// tier's own kernels are not launched) take the dense tier's rows: k_desc_group, which lists them, has completed.
static_assert(FX_DSLOW_T == FX_WG && FX_DESC_WG_FAST_T == FX_WG, "k_desc_mid's three kinds of workgroups");
extern "C" __global__ __launch_bounds__(FX_WG) void k_desc_mid(FxDevParams P, FxBuffers B, uint32_t batch, uint32_t cap,
                                                                uint32_t n_wg, uint32_t n_dslow) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // (k_desc_group, which fills the dense tier's row list, has completed)
    B.tier_hint[4] = B.counters[6];
    B.tier_hint[5] = B.counters[13];
  }
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t n_mid = gridDim.x - n_dslow;
  if (blockIdx.x >= n_mid)
    dense_slow_loop(P, B, smem, blockIdx.x - n_mid, n_dslow);
  else if (blockIdx.x < n_wg)
    desc_wg_loop<true, FX_DESC_WG_FAST_T>(P, B, batch, cap, smem, blockIdx.x, n_wg);
  else
    desc_wave_body<true>(P, B, batch, smem, blockIdx.x - n_wg, n_mid - n_wg);
}

// ---------------------------------------------------------------- dense tier
// Rows with more than P.dense_min support points (dense many-ring scans: thousands of ground returns around a pole
// next to the sensor), and every row whose list overflowed its slots.  3DSC's local point density — the number of
// cloud points within R/5 of a neighbour — does not depend on the keypoint: it is a property of
// the point.  So the tier is built around a per-scan density cache indexed by point:
//   k_dense_sort     one workgroup per row: the support set (list + the row's entries of the scan's overflow region),
//                    counting-sorted by a 25 x 25 x 7 cell grid over the support sphere's box (cells half a density
//                    radius wide in x and y) into the row's region of the sorted pool; the row's cell table; and the
//                    row's QUERY list: its neighbours whose density no other row of the batch has claimed yet
//                    (atomic compare-and-swap on the cache), in cell order;
//   k_dense_density  one workgroup per 1024 consecutive queries, four per lane: the cell rows within reach pass through
//                    LDS, every lane walks the part its queries can reach (per row a contiguous run, narrowed to the
//                    sphere's chord) and tests every target against its four queries with packed fp32 arithmetic in
//                    FLANN's operation order; counts -> the cache;
//   k_dense_finish   one workgroup per row: (bin, d2, index) keys of the neighbours sorted in LDS — counting sort by
//                    bin, ranking inside the bin (a bitonic network in the row's region of the key pool beyond 14336
//                    keys) — weights from the cache, sequential fp32 sum per bin.
// Many wavefronts share a row, rows share densities, nothing is streamed per row from the scan, and no support set is
// too large: the pools bound the batch, not the row (FX_FLAG_NBR_OVERFLOW when they are exhausted).
#define FX_DG 25                    // cells per axis in x and y: 2 (R + R/5) / (R/10) = 24, + 1
#define FX_DGZ 7                    // layers (at least one density radius high)
#define FX_DCELLS (FX_DG * FX_DG * FX_DGZ)
#define FX_DENS_BITS 21             // density count (max_points <= 2^20) in the low bits of a cache word, batch tag above
#define FX_DENS_MASK ((1ull << FX_DENS_BITS) - 1ull)
#ifndef FX_DSORT_T
#define FX_DSORT_T 512
#endif
#ifndef FX_DSORT_PER_CU
#define FX_DSORT_PER_CU (1024 / FX_DSORT_T)  // workgroups a CU of the full grid
#endif
#define FX_DSORT_WONW 2048     // words of k_dense_sort's winners bit map
#define FX_DQ_WON 0x80000000u  // sorted region, index word: the row computes this point's density (set by k_dense_sort)
#ifndef FX_DFIN_K
#define FX_DFIN_K 14336    // binned neighbours the finishing kernel sorts in LDS (keys + order + bin table: 152 KB)
#endif
#ifndef FX_DFIN_T
#define FX_DFIN_T 1024
#endif
#ifndef FX_DFIN_OCC
#define FX_DFIN_OCC 1
#endif
// size classes of the tier's rows and the batch counters that count them (k_desc_group fills the class lists)
__device__ __forceinline__ uint32_t dense_class(uint32_t nS) { return nS > 8192u ? 0u : (nS > 4096u ? 1u : (nS > 2048u ? 2u : 3u)); }
__device__ __forceinline__ uint32_t dense_class_counter(uint32_t cls) { return cls == 0u ? 2u : (cls == 1u ? 3u : (cls == 2u ? 7u : 10u)); }
// i-th row of the tier, largest class first (FX_NONE past the end)
__device__ __forceinline__ uint32_t dense_nth(const FxDevParams &P, const FxBuffers &B, uint32_t i) {
#pragma unroll
  for (uint32_t cls = 0; cls < 4; ++cls) {
    const uint32_t n = min(B.counters[dense_class_counter(cls)], P.max_dense_rows);
    if (i < n) return B.dense_order[cls * P.max_dense_rows + i];
    i -= n;
  }
  return FX_NONE;
}
// next row of the tier for this workgroup: a ticket from a batch counter, broadcast through LDS (contains two barriers)
__device__ __forceinline__ uint32_t dense_next(const FxDevParams &P, const FxBuffers &B, uint32_t ticket_counter, uint32_t *s_slot) {
  __syncthreads();
  if (threadIdx.x == 0) *s_slot = dense_nth(P, B, atomicAdd(&B.counters[ticket_counter], 1u));
  __syncthreads();
  return *s_slot;
}
struct DenseGrid {
  float gx0, gy0, gz0, inv_cw, inv_ch;
  __device__ __forceinline__ uint32_t cx(float x) const { return (uint32_t)min(max((int)floorf((x - gx0) * inv_cw), 0), FX_DG - 1); }
  __device__ __forceinline__ uint32_t cy(float y) const { return (uint32_t)min(max((int)floorf((y - gy0) * inv_cw), 0), FX_DG - 1); }
  __device__ __forceinline__ uint32_t cz(float z) const { return (uint32_t)min(max((int)floorf((z - gz0) * inv_ch), 0), FX_DGZ - 1); }
  __device__ __forceinline__ uint32_t cell(float x, float y, float z) const { return (cz(z) * FX_DG + cy(y)) * FX_DG + cx(x); }
};
// Cells slightly wider than half a density radius: two points closer than the density radius are at most two cells
// apart in x and in y, and at most one layer apart in z.
__device__ __forceinline__ DenseGrid dense_grid(const FxDevParams &P, const float4 kp) {
  const float r_sup = sqrtf(P.r2_support), r_d = sqrtf(P.r2_density);
  const float cw = fmaxf(0.5f * r_d * 1.001f, 2.0f * r_sup / (float)(FX_DG - 1) * 1.0001f);
  const float ch = fmaxf(r_d * 1.001f, 2.0f * r_sup / (float)(FX_DGZ - 1) * 1.0001f);
  DenseGrid g;
  g.gx0 = kp.x - r_sup, g.gy0 = kp.y - r_sup, g.gz0 = kp.z - r_sup;
  g.inv_cw = 1.0f / cw, g.inv_ch = 1.0f / ch;
  return g;
}
__device__ __forceinline__ void dense_row_failed(const FxDevParams &P, const FxBuffers &B, uint32_t slot, uint32_t row, uint32_t scan,
                                                 uint32_t k, uint32_t nt) {
  if (threadIdx.x == 0) {
    atomicOr(&B.flags[scan], FX_FLAG_NBR_OVERFLOW);
    B.kp_nbrs[(size_t)scan * P.max_keypoints + k] = FX_NONE;
    B.dense_nq[slot] = 0u;
    B.dense_nm[slot] = FX_NONE;
  }
  desc_fill_nan(B.desc + (size_t)row * FX_DESC_FLOATS, threadIdx.x, nt);
}

// In-place exclusive prefix over the FX_DCELLS entries of an LDS table by all NT threads of the workgroup (PAD: every entry
// rounded up to a multiple of four first); returns the total.  tmp: NT / 64 words.  Contains barriers; the caller adds one
// before the table is read.
template <int NT, bool PAD>
__device__ __forceinline__ uint32_t dense_cells_prefix(uint32_t *cells, uint32_t *tmp) {
  constexpr uint32_t per = (FX_DCELLS + NT - 1) / NT;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  uint32_t sum = 0;
  for (uint32_t u = 0; u < per; ++u) {
    const uint32_t ci = tid * per + u;
    const uint32_t c = ci < FX_DCELLS ? cells[ci] : 0u;
    sum += PAD ? (c + 3u) & ~3u : c;
  }
  uint32_t incl = sum;
  incl = wave_incl_scan(incl);
  if (lane == 63u) tmp[wave] = incl;
  __syncthreads();
  uint32_t run = incl - sum, total = 0;
#pragma unroll
  for (uint32_t w = 0; w < NT / 64; ++w) {
    const uint32_t t = tmp[w];
    run += w < wave ? t : 0u;
    total += t;
  }
  for (uint32_t u = 0; u < per; ++u) {
    const uint32_t ci = tid * per + u;
    if (ci < FX_DCELLS) {
      const uint32_t c = PAD ? (cells[ci] + 3u) & ~3u : cells[ci];
      cells[ci] = run;
      run += c;
    }
  }
  return total;
}

extern "C" __global__ __launch_bounds__(FX_DSORT_T) void k_dense_sort(FxDevParams P, FxBuffers B) {
  __shared__ uint32_t s_scan[FX_DSORT_T / 64];
  __shared__ uint32_t cell_end[FX_DCELLS + 1];
  __shared__ uint32_t s_w[16];
  const uint32_t tid = threadIdx.x;
  __shared__ uint32_t s_slot;
  __shared__ uint32_t won_bits[FX_DSORT_WONW];
  __shared__ uint32_t cell_q[FX_DCELLS + 1];  // queries per cell, then the cell's part of the query list
  if (B.counters[6] == 0u) return;  // no dense rows in this batch (sparse scans): not even a ticket is drawn
  const unsigned long long seq = B.seq[0];
  FX_STAMP_INIT(B.stamps ? B.stamps + 16 : nullptr);
  while (true) {
    const uint32_t slot = dense_next(P, B, 11u, &s_slot);
    if (slot == FX_NONE) break;
    const uint32_t row = B.dense_rows[slot];
    if (row == FX_NONE) {  // (no room in the pools: k_desc_group flagged it)
      if (tid == 0) B.dense_nq[slot] = 0u, B.dense_nm[slot] = FX_NONE;
      continue;
    }
    const uint2 rm = B.row_map[row];
    const uint32_t scan = rm.x, k = rm.y;
    const float4 kp = B.row_kp[row];
    const uint32_t nS = B.s_cnt[row], off = B.dense_off[slot];
    const uint32_t n_list = min(nS, P.list_cap);
    const uint32_t n_ovf = nS > P.list_cap ? min(B.ovf_cnt[scan], P.ovf_cap) : 0u;  // entries of the scan's overflow region (any row's)
    const float4 *lst = B.s_pts + (size_t)row * P.list_cap;
    const float4 *ovf = B.ovf_pts + (size_t)scan * P.ovf_cap;
    const uint32_t *ovf_kp = B.ovf_kp + (size_t)scan * P.ovf_cap;
    const DenseGrid G = dense_grid(P, kp);
    float4 *dst = B.dense_pts + off;
    uint32_t *table = B.dense_cells + (size_t)slot * FX_DCELLS;
    __syncthreads();
#ifdef FX_STAMPS
    stamp_prev_ = __builtin_amdgcn_s_memtime();
#endif
    const bool in_lds = nS <= min(32u * FX_DSORT_WONW, P.dense_won_points);
    for (uint32_t t = tid; t < FX_DCELLS + 1; t += FX_DSORT_T) cell_end[t] = 0u, cell_q[t] = 0u;
    if (in_lds)
      for (uint32_t t = tid; t < (nS + 31u) / 32u; t += FX_DSORT_T) won_bits[t] = 0u;
    if (tid < 16) s_w[tid] = 0;
    __syncthreads();
    // every support point of the row: its list, then its entries of the scan's overflow region; four loads in flight
    // per lane (a pass is a chain of L2 round trips otherwise).  fn(v, u) sees the four points of a lane one after the
    // other, then done() once: what it starts for a point (an atomic whose result it needs) it finishes there.
    uint32_t mine = 0;
    auto each_point = [&](auto &&fn, auto &&done) {
      for (uint32_t e0 = 0; e0 < n_list; e0 += 4u * FX_DSORT_T) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t e = e0 + (uint32_t)u * FX_DSORT_T + tid;
          if (e < n_list) v[u] = lst[e];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) fn(v[u], u, e0 + (uint32_t)u * FX_DSORT_T + tid < n_list);
        done();
      }
      for (uint32_t e0 = 0; e0 < n_ovf; e0 += 4u * FX_DSORT_T) {
        uint32_t kk[4];
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t e = e0 + (uint32_t)u * FX_DSORT_T + tid;
          kk[u] = e < n_ovf ? ovf_kp[e] : FX_NONE;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (kk[u] == k) v[u] = ovf[e0 + (uint32_t)u * FX_DSORT_T + tid];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          fn(v[u], u, kk[u] == k);
          mine += kk[u] == k ? 1u : 0u;
        }
        done();
      }
    };
    // ---- pass 1: support points per cell
    each_point([&](const float4 &v, int, bool valid) { if (valid) atomicAdd(&cell_end[G.cell(v.x, v.y, v.z)], 1u); }, [] {});
    if (mine) atomicAdd(&s_w[8], mine);
    __syncthreads();
    FX_STAMP(1);
    if (n_list + s_w[8] != nS) {  // the scan's overflow region overflowed (workgroup-uniform): entries were lost
      dense_row_failed(P, B, slot, row, scan, k, FX_DSORT_T);
      continue;
    }
    dense_cells_prefix<FX_DSORT_T, false>(cell_end, s_scan);  // counts -> exclusive starts, in place
    __syncthreads();
    FX_STAMP(2);
    // ---- pass 2: every point to its cell (the fill turns a cell's start into its end = the next cell's start), and on
    //      the way: neighbours (d2 < R^2; the count 3DSC reports), binned neighbours (not the keypoint's own point), and the
    //      queries: binned neighbours whose density this row is the first to claim — claimed here, counted per cell and
    //      remembered (a bit map in LDS over the sorted positions; the top bit of the point's index word in the sorted region
    //      for rows beyond 65536 support points).  Batch tags only grow and a claim is the largest word of its batch: one
    //      atomic max both tests and claims; a lane's four are in flight together.
    unsigned long long *cache = B.dens_cache + (size_t)scan * P.max_points;
    const unsigned long long claim = (seq << FX_DENS_BITS) | FX_DENS_MASK;
    uint32_t n_nb = 0, n_use = 0;
    {
      float4 pv[4];
      uint32_t pp[4];
      unsigned long long old[4];
      bool use[4], val[4];
      each_point(
          [&](const float4 &v, int u, bool valid) {
            val[u] = valid, use[u] = false, pv[u] = v;
            if (!valid) return;
            pp[u] = atomicAdd(&cell_end[G.cell(v.x, v.y, v.z)], 1u);
            const float d2 = dist2(kp.x, kp.y, kp.z, v.x, v.y, v.z);
            const bool nb = d2 < P.r2_search;
            use[u] = nb && !sc3d_is_origin(d2);
            n_nb += nb ? 1u : 0u;
            n_use += use[u] ? 1u : 0u;
#ifdef FX_NO_DEDUPE  // (diagnostic: every row computes all its neighbours' densities itself)
            old[u] = 0ull;
            if (use[u]) atomicMax(cache + __float_as_uint(v.w), claim);
#else
            old[u] = use[u] ? atomicMax(cache + __float_as_uint(v.w), claim) : claim;
#endif
          },
          [&] {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              if (!val[u]) continue;
              const bool won = use[u] && (old[u] >> FX_DENS_BITS) != seq;  // this row computes the point's density
              float4 v = pv[u];
              if (won) {
                atomicAdd(&cell_q[G.cell(v.x, v.y, v.z)], 1u);
                if (in_lds)
                  atomicOr(&won_bits[pp[u] >> 5], 1u << (pp[u] & 31u));
                else
                  v.w = __uint_as_float(__float_as_uint(v.w) | FX_DQ_WON);
              }
              dst[pp[u]] = v;
            }
          });
    }
    if (n_nb) atomicAdd(&s_w[9], n_nb);
    if (n_use) atomicAdd(&s_w[10], n_use);
    // (the sorted region is re-read below by other waves of this workgroup: they share the CU's L1, and the barrier's
    //  workgroup-scope fences order the stores — an agent-scope __threadfence() would write the L2 back each time)
    wg_global_sync();
    FX_STAMP(3);
    for (uint32_t t = tid; t < FX_DCELLS; t += FX_DSORT_T) table[t] = cell_end[t];  // cell c = [c ? end[c - 1] : 0, end[c])
    // ---- the query list, cell by cell, every cell's part padded to a multiple of four entries: k_dense_density takes the
    // queries in quads, and four queries of ONE cell have a box no larger than the cell.  (Unpadded, 4 % of the quads held
    // queries of three or more cells — the sparse stretches of a row, the edges of what other rows had claimed — and
    // walked twice as far as the others.)  A padded prefix over the cells and the row's share of the query pool; then
    // every winner to its cell's part of the list.
    {
      const uint32_t total = dense_cells_prefix<FX_DSORT_T, true>(cell_q, s_scan);
      if (tid == 0) {  // the row's share of the query pool
        s_w[12] = total;
        const uint32_t qo = atomicAdd(&B.counters[FX_CNT_QPOOL], total);
        s_w[13] = qo;
        s_w[14] = (qo <= P.dense_qcap && total <= P.dense_qcap - qo) ? 1u : 0u;
      }
    }
    __syncthreads();
    const uint32_t n_q = s_w[12], qoff = s_w[13];  // (with the padding)
    if (!s_w[14]) {  // query pool exhausted (workgroup-uniform).  What the row claimed is never computed: rows that share
                     // those points fail with it — flagged below by k_dense_finish (a density of zero is not a weight).
      dense_row_failed(P, B, slot, row, scan, k, FX_DSORT_T);
      continue;
    }
    uint32_t *qlist = B.dense_q + qoff;
    // (one lane per cell walking the cell's stretch of the bit map instead — no loads of the sorted region — was measured:
    //  46 000 cycles a row against 30 000, a dense cell is one lane's serial loop)
    for (uint32_t p0 = 0; p0 < nS; p0 += 4u * FX_DSORT_T) {
      float4 v[4];
      bool won[4];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t p = p0 + u * FX_DSORT_T + tid;
        won[u] = false;
        if (p < nS && in_lds && !((won_bits[p >> 5] >> (p & 31u)) & 1u)) continue;  // (most points: no load)
        if (p < nS) {
          v[u] = dst[p];
          won[u] = in_lds || (__float_as_uint(v[u].w) & FX_DQ_WON) != 0u;
        }
      }
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u)
        if (won[u]) qlist[atomicAdd(&cell_q[G.cell(v[u].x, v[u].y, v[u].z)], 1u)] = p0 + u * FX_DSORT_T + tid;
    }
    __syncthreads();
    // the padding behind every cell's queries (cell_q[c] is now the end of what cell c wrote; its part ends at the next multiple of four)
    for (uint32_t c = tid; c < FX_DCELLS; c += FX_DSORT_T) {
      const uint32_t e = cell_q[c];
      for (uint32_t t = e; t < ((e + 3u) & ~3u); ++t) qlist[t] = FX_NONE;  // (no query: k_dense_density skips it)
    }
    __syncthreads();
    FX_STAMP(4);
    if (tid == 0) {
      FX_COUNT(6, 1);
      FX_COUNT(7, nS);
      FX_COUNT(8, n_ovf);
      B.kp_nbrs[(size_t)scan * P.max_keypoints + k] = s_w[9];
      B.dense_nq[slot] = n_q;
      B.dense_qoff[slot] = qoff;
      B.dense_nm[slot] = s_w[10];
      if (s_w[10] > min(P.dense_lds_keys, (uint32_t)FX_DFIN_K)) {  // keys beyond the finishing kernel's LDS array: a region of the key pool
        uint32_t p2 = 1;
        while (p2 < s_w[10]) p2 <<= 1;
        const uint32_t koff = atomicAdd(&B.counters[12], p2);
        if (koff <= P.dense_cap && p2 <= P.dense_cap - koff) {
          B.dense_koff[slot] = koff;
        } else {  // key pool exhausted
          atomicOr(&B.flags[scan], FX_FLAG_NBR_OVERFLOW);
          B.kp_nbrs[(size_t)scan * P.max_keypoints + k] = FX_NONE;
          B.dense_nm[slot] = FX_NONE;
          s_w[11] = 1u;
        }
      }
      // one work item per 1024 queries (FX_DDENS_Q of k_dense_density)
      const uint32_t n_items = (n_q + 1023u) / 1024u;
      const uint32_t base = n_items ? atomicAdd(&B.counters[14], n_items) : 0u;
      for (uint32_t i = 0; i < n_items; ++i) B.dense_items[base + i] = make_uint2(slot, i * 1024u);
    }
    __syncthreads();
    FX_STAMP(5);
    if (s_w[11]) desc_fill_nan(B.desc + (size_t)row * FX_DESC_FLOATS, tid, FX_DSORT_T);  // (the claimed densities are still computed: other rows read them)
  }
}

// Density of up to 1024 consecutive queries of a row by one workgroup.  The queries are taken in QUADS (four consecutive
// in cell order, so almost always in one cell): a quad shares one walk over the targets, four tests per target with packed
// fp32 arithmetic in FLANN's operation order.  The cell rows within reach of any of the workgroup's queries — per row
// one contiguous run of the sorted region — pass through LDS in windows of FX_DDENS_C targets; per window, every lane lists
// for ITS quad the part of each row that the quad's density spheres can reach (the row narrowed to the sphere's chord):
// (quad, first, length) UNITS.  The units of the workgroup's 256 quads are then counting-sorted by length and its wavefronts
// take them 64 at a time, longest first, from a shared counter — a wavefront's trip costs its longest lane, and the quads
// differ a lot: with every lane walking its own quad's rows in step a wavefront made 853 trips a quad where the mean lane
// needed 474 (config 3; 1159 / 632 on config 5: profiles/r04_front_experiments.md), because ring arcs cross the cell rows at
// different places for lanes a few cells apart and queries — the points no earlier row has claimed — lie in neighbourhoods
// of different density.  Counts go to LDS accumulators per query.
// (Per-lane walks straight from global memory were measured: 64 scattered 16-byte loads per instruction keep the
//  texture unit busier than the 24 arithmetic instructions they feed — 3.8 ms against 2.1 ms for a wave-uniform
//  stream of the whole box, which tests 2-3 times as many targets as a lane needs.)
#define FX_DDENS_T 256
#define FX_DDENS_Q (4 * FX_DDENS_T)  // queries per work item
#ifndef FX_DDENS_C
#define FX_DDENS_C 2048              // targets per window
#endif
#ifndef FX_DDENS_RUNS
#define FX_DDENS_RUNS 8              // units a lane lists per pass
#endif
#ifndef FX_DDENS_PER_CU
#define FX_DDENS_PER_CU 3             // workgroups a CU of the full grid (52 KB of LDS each)
#endif
#define FX_DDENS_BINS 96             // length classes of the units (exponent, three mantissa bits)
static_assert(FX_DDENS_C <= 2048 && FX_DDENS_T == 256, "k_dense_density packs a unit as quad (8 bits) | first (11) | length - 1 (11)");
static_assert(FX_DDENS_BINS <= FX_DDENS_T, "one bin a thread in the prefix");
// length class, longest first: bin 0 holds the longest units
__device__ __forceinline__ uint32_t ddens_bin(uint32_t len) {  // len in [1, 2048]
  const uint32_t e = 31u - (uint32_t)__clz((int)len);           // 0 .. 11
  const uint32_t m = e >= 3u ? (len >> (e - 3u)) & 7u : (len << (3u - e)) & 7u;
  return (FX_DDENS_BINS - 1u) - (e * 8u + m);
}
// exclusive prefix of one value a thread over a 256-thread workgroup (tmp: four words; contains one barrier; total: the sum)
__device__ __forceinline__ uint32_t wg256_prefix(uint32_t v, uint32_t *tmp, uint32_t &total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t incl = v;
  incl = wave_incl_scan(incl);
  if (lane == 63u) tmp[wave] = incl;
  __syncthreads();
  uint32_t before = 0;
  total = 0;
#pragma unroll
  for (uint32_t w = 0; w < 4u; ++w) {
    const uint32_t t = tmp[w];
    before += w < wave ? t : 0u;
    total += t;
  }
  return before + incl - v;
}
extern "C" __global__ __launch_bounds__(FX_DDENS_T) void k_dense_density(FxDevParams P, FxBuffers B) {
  __shared__ float s_t[3 * FX_DDENS_C];                    // the window: x | y | z
  __shared__ float4 s_q[4 * FX_DDENS_T];                   // per quad: x of its four queries, y, z, their counts (as uint32)
  __shared__ uint32_t s_unit[FX_DDENS_RUNS * FX_DDENS_T];  // the units of a pass, sorted by length
  __shared__ uint32_t s_hist[FX_DDENS_BINS];
  __shared__ uint32_t s_cat[FX_DDENS_T];    // start of every row of the box (<= 176) in the concatenation of the rows' runs
  __shared__ uint32_t s_row0[FX_DDENS_T];   // sorted-region position of the first target of every row of the box
  __shared__ uint32_t s_w[16];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t n_items = B.counters[14];
  if (n_items == 0u) return;
  const unsigned long long tag = B.seq[0] << FX_DENS_BITS;
  const float r2d = P.r2_density;
  const float r_d = sqrtf(r2d);
  // Counting d2 < r2 without a compare and an add-with-carry per test: with S a power of two, fma(d2, -S, r2 S) is the
  // exactly scaled difference rounded once — positive, zero or negative as r2 - d2 is (-inf when d2 S overflows) — at
  // least 2^76 in magnitude unless zero, so the instruction's clamp to [0, 1] turns it into 1.0f or 0.0f; the counts add
  // up exactly in fp32 (below 2^21).  Two tests per packed instruction, two instructions instead of four.
  const float kS = __uint_as_float(min(354u - ((__float_as_uint(r2d) >> 23) & 0xffu), 254u) << 23);  // r2 S in [2^100, 2^101)
                                                                                        // (fx_create: r2 >= 1e-30, so ulp(r2) S >= 1)
  const fx_f2 nS2 = {-kS, -kS}, rS2 = {r2d * kS, r2d * kS};
  while (true) {
    __syncthreads();
    if (tid == 0) s_w[0] = atomicAdd(&B.counters[15], 1u);
    if (tid >= 4 && tid < 10) s_w[tid] = (tid & 1u) ? 0u : FX_NONE;  // the box: 4 ylo 5 yhi 6 zlo 7 zhi 8 xlo 9 xhi
    __syncthreads();
    const uint32_t it = s_w[0];
    if (it >= n_items) break;
    const uint2 item = B.dense_items[it];
    const uint32_t slot = item.x, qb = item.y;
    const uint32_t row = B.dense_rows[slot], off = B.dense_off[slot], n_q = B.dense_nq[slot];
    const uint32_t scan = B.row_map[row].x;
    const DenseGrid G = dense_grid(P, B.row_kp[row]);
    const float cw = 1.0f / G.inv_cw, ch = 1.0f / G.inv_ch;
    const float4 *pts = B.dense_pts + off;
    const uint32_t *qlist = B.dense_q + B.dense_qoff[slot];
    const uint32_t *table = B.dense_cells + (size_t)slot * FX_DCELLS;  // table[c] = end of cell c in the sorted region
    auto cell_start = [&](uint32_t c) -> uint32_t { return c ? table[c - 1u] : 0u; };
    // ---- this lane's quad: its four queries and their box
    float4 q[4];
    uint32_t qid[4];
    float bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY, bz0 = INFINITY, bz1 = -INFINITY;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t qi = qb + 4u * tid + (uint32_t)u;
      q[u] = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 0.0f);  // (no query: its differences overflow to infinity, never below the radius)
      qid[u] = FX_NONE;
      const uint32_t qp = qi < n_q ? qlist[qi] : FX_NONE;  // (FX_NONE: padding at the end of a cell row)
      if (qp != FX_NONE) {
        q[u] = pts[qp];
        qid[u] = __float_as_uint(q[u].w) & ~FX_DQ_WON;
        bx0 = fminf(bx0, q[u].x), bx1 = fmaxf(bx1, q[u].x);
        by0 = fminf(by0, q[u].y), by1 = fmaxf(by1, q[u].y);
        bz0 = fminf(bz0, q[u].z), bz1 = fmaxf(bz1, q[u].z);
      }
    }
    s_q[4u * tid] = make_float4(q[0].x, q[1].x, q[2].x, q[3].x);
    s_q[4u * tid + 1u] = make_float4(q[0].y, q[1].y, q[2].y, q[3].y);
    s_q[4u * tid + 2u] = make_float4(q[0].z, q[1].z, q[2].z, q[3].z);
    s_q[4u * tid + 3u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // (four uint32 zeros)
    const bool any_q = bx0 <= bx1;
    // Margins: a point's cell comes from a rounded product, so cell borders are taken a thousandth of a cell wide.
    const float eps_w = 1e-3f * cw, eps_h = 1e-3f * ch;
    const float rr = r_d * 1.0001f;
    // rows (cy, cz) the quad reaches: two cells either side in y, one layer either side in z (dense_grid)
    uint32_t ylo = 0, ny = 0, zlo = 0, n_rows = 0;
    if (any_q) {
      const uint32_t cy0 = G.cy(by0), cy1 = G.cy(by1), cz0 = G.cz(bz0), cz1 = G.cz(bz1);
      ylo = cy0 >= 2u ? cy0 - 2u : 0u;
      const uint32_t yhi = min(cy1 + 2u, (uint32_t)FX_DG - 1u);
      zlo = cz0 >= 1u ? cz0 - 1u : 0u;
      const uint32_t zhi = min(cz1 + 1u, (uint32_t)FX_DGZ - 1u);
      ny = yhi - ylo + 1u;
      n_rows = ny * (zhi - zlo + 1u);
      const float hmax = rr * 1.0001f + eps_w;
      atomicMin(&s_w[4], ylo), atomicMax(&s_w[5], yhi), atomicMin(&s_w[6], zlo), atomicMax(&s_w[7], zhi);
      atomicMin(&s_w[8], G.cx(bx0 - hmax)), atomicMax(&s_w[9], G.cx(bx1 + hmax));
    }
    __syncthreads();
    const uint32_t UY0 = s_w[4], UY1 = s_w[5], UZ0 = s_w[6], UZ1 = s_w[7], UX0 = s_w[8], UX1 = s_w[9];
    if (UY0 > UY1) continue;  // (an item without queries: never emitted)
    const uint32_t uny = UY1 - UY0 + 1u, unr = uny * (UZ1 - UZ0 + 1u);
    // ---- the rows of the box laid end to end: s_cat[r] = start of row r in that concatenation, s_row0[r] in the sorted region
    {
      uint32_t len = 0, total;
      if (tid < unr) {
        const uint32_t base = ((UZ0 + tid / uny) * FX_DG + UY0 + tid % uny) * FX_DG;
        const uint32_t r0 = cell_start(base + UX0);
        s_row0[tid] = r0;
        len = table[base + UX1] - r0;
      }
      const uint32_t excl = wg256_prefix(len, &s_w[10], total);  // (unr <= 175 < 256: one row a thread)
      if (tid <= unr) s_cat[tid] = excl;                          // (s_cat[unr] = the total)
    }
    __syncthreads();
    const uint32_t t_tot = s_cat[unr];
#ifdef FX_STAMPS
    unsigned long long n_tests = 0, n_wave = 0;
#endif
    uint32_t r0 = 0, cy0 = ylo, cz0 = zlo;  // the lane's first row with targets the windows so far have not covered
    for (uint32_t w = 0; w < t_tot; w += FX_DDENS_C) {
      const uint32_t wn = min((uint32_t)FX_DDENS_C, t_tot - w);
      // ---- the window: concatenation position -> row (binary search) -> sorted-region position
      if (tid < wn) {
        uint32_t lo = 0, hi = unr;  // invariant: s_cat[lo] <= g < s_cat[hi]; a binary search for the thread's first target ...
        while (hi - lo > 1u) {
          const uint32_t mid = (lo + hi) >> 1;
          if (s_cat[mid] <= w + tid)
            lo = mid;
          else
            hi = mid;
        }
        for (uint32_t f = tid; f < wn; f += FX_DDENS_T) {
          const uint32_t g = w + f;
          while (s_cat[lo + 1u] <= g) ++lo;  // ... a step or two for the next ones (s_cat[unr] = t_tot > g)
          const float4 t = pts[s_row0[lo] + (g - s_cat[lo])];
          s_t[f] = t.x, s_t[FX_DDENS_C + f] = t.y, s_t[2 * FX_DDENS_C + f] = t.z;
        }
      }
      __syncthreads();
#if defined(FX_DDENS_STOP) && FX_DDENS_STOP == 1  // (measurement build: windows loaded, nothing walked)
      __syncthreads();
      continue;
#endif
      // ---- passes: every lane lists up to FX_DDENS_RUNS units of its quad, the workgroup sorts them, its wavefronts walk them
      // (the rows come in the order of the concatenation, so a lane's cursor r0 only moves forward: rows that end before the
      //  window are behind it, the row loop stops at the first row that starts after the window)
      uint32_t r = r0, cy = cy0, cz = cz0, lim = n_rows;
      while (true) {
        uint32_t n_run = 0;
        if (tid < FX_DDENS_BINS) s_hist[tid] = 0u;
        if (tid == 0) s_w[1] = 0u;  // the slot counter
        while (r < lim && n_run < FX_DDENS_RUNS) {
          const uint32_t rcy = cy, rcz = cz;
          ++r, cz += (cy + 1u == ylo + ny) ? 1u : 0u, cy = (cy + 1u == ylo + ny) ? ylo : cy + 1u;  // (the next row)
          const uint32_t ur = (rcz - UZ0) * uny + (rcy - UY0);
          const uint32_t rb = s_cat[ur], re = s_cat[ur + 1u];
          if (rb >= w + wn) {  // this row and the ones after it: later windows
            --r, cy = rcy, cz = rcz;
            lim = r;
            break;
          }
          if (re <= w + wn) r0 = r, cy0 = cy, cz0 = cz;  // nothing of this row beyond the window (nor of the rows before it)
          if (re <= w) continue;
          const float y0 = G.gy0 + (float)rcy * cw - eps_w, y1 = G.gy0 + (float)(rcy + 1u) * cw + eps_w;
          const float z0 = G.gz0 + (float)rcz * ch - eps_h, z1 = G.gz0 + (float)(rcz + 1u) * ch + eps_h;
          // (edge cells also hold what the clamp put there: they extend outwards without limit)
          const float dy = fmaxf(fmaxf(rcy == 0u ? 0.0f : y0 - by1, rcy == FX_DG - 1u ? 0.0f : by0 - y1), 0.0f);
          const float dz = fmaxf(fmaxf(rcz == 0u ? 0.0f : z0 - bz1, rcz == FX_DGZ - 1u ? 0.0f : bz0 - z1), 0.0f);
          const float h2 = rr * rr - (dy * dy + dz * dz);
          if (!(h2 > 0.0f)) continue;  // the row is out of reach
          const float h = sqrtf(h2) * 1.0001f + eps_w;
          const uint32_t base = (rcz * FX_DG + rcy) * FX_DG;
          // window-relative positions of the run [start(cxl), end(cxh)) of this row
          const uint32_t shift = rb - s_row0[ur];  // sorted-region position -> concatenation position (mod 2^32)
          const uint32_t g0 = cell_start(base + G.cx(bx0 - h)) + shift, g1 = table[base + G.cx(bx1 + h)] + shift;
          const uint32_t i0 = g0 > w ? g0 - w : 0u, i1 = g1 > w ? min(g1 - w, wn) : 0u;
          if (i1 > i0) s_unit[n_run++ * FX_DDENS_T + tid] = tid | i0 << 8 | (i1 - i0 - 1u) << 19;  // (the lane's own column for now)
        }
        if (!__syncthreads_or((int)n_run)) break;  // (no lane listed anything: every lane is through its rows; the histogram is clear)
        uint32_t run[FX_DDENS_RUNS];
#pragma unroll
        for (int k = 0; k < FX_DDENS_RUNS; ++k)
          if ((uint32_t)k < n_run) {
            run[k] = s_unit[k * FX_DDENS_T + tid];
            atomicAdd(&s_hist[ddens_bin((run[k] >> 19) + 1u)], 1u);
          }
        __syncthreads();  // (every lane has its column in registers: the sorted list may overwrite it)
        uint32_t n_units;
        {  // exclusive prefix over the bins (longest first)
          const uint32_t h = tid < FX_DDENS_BINS ? s_hist[tid] : 0u;
          const uint32_t excl = wg256_prefix(h, &s_w[10], n_units);
          if (tid < FX_DDENS_BINS) s_hist[tid] = excl;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < FX_DDENS_RUNS; ++k)
          if ((uint32_t)k < n_run) s_unit[atomicAdd(&s_hist[ddens_bin((run[k] >> 19) + 1u)], 1u)] = run[k];
        __syncthreads();
#if defined(FX_DDENS_STOP) && FX_DDENS_STOP == 2  // (measurement build: units listed and sorted, nothing walked)
        __syncthreads();
        continue;
#endif
        while (true) {  // slots of 64 units, longest first, to whichever wavefront is free
          uint32_t j = 0;
          if (lane == 0) j = atomicAdd(&s_w[1], 64u);
          j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
          if (j >= n_units) break;
          if (j + lane < n_units) {
            const uint32_t u = s_unit[j + lane];
            const uint32_t quad = u & 255u, i0 = (u >> 8) & 2047u, i1 = i0 + (u >> 19) + 1u;
            const float4 qx = s_q[4u * quad], qy = s_q[4u * quad + 1u], qz = s_q[4u * quad + 2u];
            const fx_f2 ax = {qx.x, qx.y}, ay = {qy.x, qy.y}, az = {qz.x, qz.y};
            const fx_f2 bx = {qx.z, qx.w}, by = {qy.z, qy.w}, bz = {qz.z, qz.w};
            fx_f2 ca = {0.0f, 0.0f}, cb = {0.0f, 0.0f};
            auto test = [&](const float tx, const float ty, const float tz) {
              // FLANN L2_Simple, query - point, ((dx dx) + dy dy) + dz dz: two queries per packed instruction
              const fx_f2 dxa = ax - tx, dya = ay - ty, dza = az - tz;
              const fx_f2 dxb = bx - tx, dyb = by - ty, dzb = bz - tz;
              fx_f2 ra = dxa * dxa, rb2 = dxb * dxb;
              ra = ra + dya * dya, rb2 = rb2 + dyb * dyb;
              ra = ra + dza * dza, rb2 = rb2 + dzb * dzb;
              fx_f2 ia, ib;
              asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(ia) : "v"(ra), "s"(nS2), "v"(rS2));
              asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(ib) : "v"(rb2), "s"(nS2), "v"(rS2));
              ca = ca + ia, cb = cb + ib;
            };
            uint32_t i = i0;
            for (; i + 1u < i1; i += 2u) {  // (unrolled by hand: the pragma gives up on a loop with inline assembly)
              const float tx0 = s_t[i], ty0 = s_t[FX_DDENS_C + i], tz0 = s_t[2 * FX_DDENS_C + i];
              const float tx1 = s_t[i + 1u], ty1 = s_t[FX_DDENS_C + i + 1u], tz1 = s_t[2 * FX_DDENS_C + i + 1u];
              test(tx0, ty0, tz0);
              test(tx1, ty1, tz1);
            }
            if (i < i1) test(s_t[i], s_t[FX_DDENS_C + i], s_t[2 * FX_DDENS_C + i]);
            uint32_t *cnt = reinterpret_cast<uint32_t *>(&s_q[4u * quad + 3u]);
            atomicAdd(&cnt[0], (uint32_t)ca.x), atomicAdd(&cnt[1], (uint32_t)ca.y);
            atomicAdd(&cnt[2], (uint32_t)cb.x), atomicAdd(&cnt[3], (uint32_t)cb.y);
#ifdef FX_STAMPS
            n_tests += i1 - i0;
#endif
          }
#ifdef FX_STAMPS
          n_wave += (s_unit[j] >> 19) + 1u;  // (sorted: the slot's first unit is its longest)
#endif
        }
        __syncthreads();  // (the units are read: the next pass may overwrite them)
      }
      // (the pass loop leaves through a barrier: the window may be overwritten)
    }
    __syncthreads();  // (every wavefront's counts are in)
    // ---- counts -> the scan's density cache (k_dense_finish of every row that has the point as a neighbour reads them)
    const uint32_t *cnt = reinterpret_cast<const uint32_t *>(&s_q[4u * tid + 3u]);
#ifdef FX_STAMPS
    if (B.stamps) {  // diagnostic: targets walked per quad, per wavefront (slot by slot: the longest unit), true densities, queries
      unsigned long long nq = 0, dsum = 0;
      for (int u = 0; u < 4; ++u)
        if (qid[u] != FX_NONE) nq += 1, dsum += cnt[u];
      atomicAdd(&B.stamps[44 + 0], n_tests);
      atomicAdd(&B.stamps[44 + 1], dsum);
      atomicAdd(&B.stamps[44 + 2], nq);
      if (lane == 0) atomicAdd(&B.stamps[44 + 3], n_wave);
      if (any_q) atomicAdd(&B.stamps[44 + 4], 1ull);
    }
#endif
    unsigned long long *cache = B.dens_cache + (size_t)scan * P.max_points;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (qid[u] != FX_NONE) cache[qid[u]] = tag | (unsigned long long)cnt[u];
  }
}

// Bitonic sort of p2 (a power of two) keys in global memory by NT threads (rows whose keys do not fit LDS).
template <int NT>
__device__ __forceinline__ void dense_bitonic_global(unsigned long long *sk, uint32_t p2) {
  for (uint32_t kb = 2; kb <= p2; kb <<= 1) {
    for (uint32_t jb = kb >> 1; jb > 0; jb >>= 1) {
      for (uint32_t t = threadIdx.x; t < p2; t += NT) {
        const uint32_t x = t ^ jb;
        if (x > t) {
          const unsigned long long a = sk[t], c = sk[x];
          const bool up = (t & kb) == 0;
          if ((a > c) == up) {
            sk[t] = c;
            sk[x] = a;
          }
        }
      }
      wg_global_sync();
    }
  }
}
// The (bin, d2, index) key of every binned neighbour of the row -> sk[0 .. nM), in any order; HIST: also a histogram of
// the bins (LDS, cleared by the caller).
template <int NT, bool HIST>
__device__ __forceinline__ void dense_keys(const FxDevParams &P, const float4 *pts, uint32_t nS, const float4 kp, const float2 xa,
                                           const FxScTables *T, unsigned long long *sk, uint32_t *hist, uint32_t *s_w) {
  const uint32_t tid = threadIdx.x;
  for (uint32_t p0 = 0; p0 < nS; p0 += NT) {
    const uint32_t p = p0 + tid;
    bool use = false;
    unsigned long long key = 0;
    if (p < nS) {
      const float4 v = pts[p];
      const float d2 = dist2(kp.x, kp.y, kp.z, v.x, v.y, v.z);
      use = d2 < P.r2_search && !sc3d_is_origin(d2);
      if (use) {
        float lut;
        bool amb = false;
        const uint32_t bin = sc3d_bin<true>(kp, v.x, v.y, v.z, d2, xa, T, lut, amb);
        key = sc3d_key(bin, d2, __float_as_uint(v.w) & ~FX_DQ_WON);
        if (HIST) atomicAdd(&hist[bin], 1u);
      }
    }
    const unsigned long long m = __ballot(use);
    if (m) {  // (wave-uniform)
      uint32_t base = 0;
      if ((tid & 63u) == 0) base = atomicAdd(&s_w[0], (uint32_t)__popcll(m));
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      if (use) sk[base + lanes_below(m)] = key;
    }
  }
}
// weight of one binned neighbour: (1 / local point density) x the bin's volume normalisation.  The
// density comes from the scan's cache; a word of another batch or a bare claim means the row that claimed the point did
// not get to compute it (its share of the query pool was refused): *missing is set and the row fails, flagged.
__device__ __forceinline__ float dense_weight(unsigned long long key, const unsigned long long *cache, unsigned long long seq,
                                              const FxScTables *T, uint32_t *missing) {
  const unsigned long long cw = cache[(uint32_t)(key & 0xfffffull)];
  const uint32_t dens = (cw >> FX_DENS_BITS) == seq ? (uint32_t)(cw & FX_DENS_MASK) : 0u;
  if (dens == 0u || dens == (uint32_t)FX_DENS_MASK) *missing = 1u;
  return (1.0f / (float)dens) * T->lut[(uint32_t)(key >> 52) % 165u];
}
// One row.  PCL adds a bin's contributions in the order of its sorted radius search, (d2, index) ascending, so the keys
// are sorted by (bin, d2, index) — here as a counting sort by bin (1980 of them) followed by a ranking inside each bin
// (a bitonic network over 16384 keys took 0.28 ms of a CU) — and one lane per bin then adds the bin's weights in order.
// LDS: keys [KMAX] u64 (later the sorted weights), ord [KMAX] u16, bin table [1984].
template <int KMAX, int NT>
__device__ __forceinline__ void dense_finish_row(const FxDevParams &P, const FxBuffers &B, uint32_t slot, uint32_t *smem,
                                                 const FxScTables *T, unsigned long long seq) {
  constexpr int PER = (KMAX + NT - 1) / NT;
  unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);
  uint16_t *ord = reinterpret_cast<uint16_t *>(smem + 2 * KMAX);
  uint32_t *bin_end = smem + 2 * KMAX + KMAX / 2;  // [1984]
  uint32_t *s_w = bin_end + 1984 + FX_TABLE_WORDS;
  const uint32_t tid = threadIdx.x;
  const uint32_t row = B.dense_rows[slot], off = B.dense_off[slot], nM = B.dense_nm[slot];
  const uint32_t scan = B.row_map[row].x;
  const uint32_t nS = B.s_cnt[row];
  const float4 kp = B.row_kp[row];
  const float2 xa = B.row_xa[row];
  float *out = B.desc + (size_t)row * FX_DESC_FLOATS;
  const float4 *pts = B.dense_pts + off;
  const unsigned long long *cache = B.dens_cache + (size_t)scan * P.max_points;
#ifdef FX_STAMPS_FINISH  // (-DFX_STAMPS -DFX_STAMPS_FINISH: the columns are the list tier's otherwise)
  FX_STAMP_INIT(B.stamps);
#else
  FX_STAMP_INIT((unsigned long long *)nullptr);
#endif
  __syncthreads();
  if (tid == 0) s_w[0] = 0, s_w[1] = 0;
  if (nM > (uint32_t)KMAX || nM > P.dense_lds_keys) {
    // ---- more keys than LDS holds: a bitonic network over the row's region of the key pool
    unsigned long long *sk = B.dense_key + B.dense_koff[slot];
    uint32_t p2 = 1;
    while (p2 < nM) p2 <<= 1;
    __syncthreads();
    dense_keys<NT, false>(P, pts, nS, kp, xa, T, sk, nullptr, s_w);
    for (uint32_t t = nM + tid; t < p2; t += NT) sk[t] = ~0ull;
    wg_global_sync();
    dense_bitonic_global<NT>(sk, p2);
    for (uint32_t t = tid; t < nM; t += NT) {  // every sorted key becomes (bin, weight) in place
      const unsigned long long key = sk[t];
      sk[t] = ((key >> 52) << 32) | (unsigned long long)__float_as_uint(dense_weight(key, cache, seq, T, &s_w[1]));
    }
    wg_global_sync();
    if (s_w[1]) {  // a density this row needs was never computed (workgroup-uniform)
      if (tid == 0) {
        atomicOr(&B.flags[scan], FX_FLAG_NBR_OVERFLOW);
        B.kp_nbrs[(size_t)scan * P.max_keypoints + B.row_map[row].y] = FX_NONE;
      }
      desc_fill_nan(out, tid, NT);
      __syncthreads();
      return;
    }
    for (uint32_t t = tid; t < nM; t += NT) {  // one lane per bin run
      const uint32_t bin = (uint32_t)(sk[t] >> 32);
      if (t > 0 && (uint32_t)(sk[t - 1] >> 32) == bin) continue;
      float acc = 0.0f;
      uint32_t e = t;
      do {
        acc += __uint_as_float((uint32_t)sk[e]);
        ++e;
      } while (e < nM && (uint32_t)(sk[e] >> 32) == bin);
      out[bin] = acc;
    }
    __syncthreads();
    return;
  }
  for (uint32_t t = tid; t < 1984; t += NT) bin_end[t] = 0;
  __syncthreads();
  dense_keys<NT, true>(P, pts, nS, kp, xa, T, keys, bin_end, s_w);
  __syncthreads();
  FX_STAMP(49);
  if (tid < 64) {  // counts -> exclusive starts, in place, by one wavefront
    constexpr uint32_t per = 1984 / 64;
    uint32_t sum = 0;
    for (uint32_t u = 0; u < per; ++u) sum += bin_end[tid * per + u];
    uint32_t incl = sum;
    incl = wave_incl_scan(incl);
    uint32_t run = incl - sum;
    for (uint32_t u = 0; u < per; ++u) {
      const uint32_t c = bin_end[tid * per + u];
      bin_end[tid * per + u] = run;
      run += c;
    }
  }
  __syncthreads();
  // every key's index to its bin's segment (the fill turns a bin's start into its end = the next bin's start)
  FX_STAMP(50);
  for (uint32_t e = tid; e < nM; e += NT) ord[atomicAdd(&bin_end[(uint32_t)(keys[e] >> 52)], 1u)] = (uint16_t)e;
  __syncthreads();
  FX_STAMP(51);
  // the keys into bin order, in place (through registers), so that the ranking below walks a bin's keys at consecutive
  // addresses instead of through their indices: two dependent LDS reads a step were most of this kernel's ranking time
  unsigned long long mykey[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const uint32_t p = tid + (uint32_t)u * NT;
    mykey[u] = p < nM ? keys[ord[p]] : 0ull;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const uint32_t p = tid + (uint32_t)u * NT;
    if (p < nM) keys[p] = mykey[u];
  }
  __syncthreads();
  // rank inside the bin (keys are unique: they end in the point index), and the weight
  uint32_t dst[PER];
  float wgt[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const uint32_t p = tid + (uint32_t)u * NT;
    dst[u] = FX_NONE;
    if (p < nM) {
      const unsigned long long key = mykey[u];
      const uint32_t bin = (uint32_t)(key >> 52);
      const uint32_t s0 = bin ? bin_end[bin - 1u] : 0u, s1 = bin_end[bin];
      uint32_t rank = 0;
#pragma unroll 4
      for (uint32_t qq = s0; qq < s1; ++qq) rank += keys[qq] < key ? 1u : 0u;
      dst[u] = s0 + rank;
    }
  }
#ifdef FX_STAMPS
  __syncthreads();
  FX_STAMP(57);
#endif
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const uint32_t p = tid + (uint32_t)u * NT;
    if (p < nM) wgt[u] = dense_weight(mykey[u], cache, seq, T, &s_w[1]);
  }
  __syncthreads();  // (the keys are done with: their storage takes the weights, in sorted order)
  FX_STAMP(52);
  if (s_w[1]) {  // a density this row needs was never computed (workgroup-uniform)
    if (tid == 0) {
      atomicOr(&B.flags[scan], FX_FLAG_NBR_OVERFLOW);
      B.kp_nbrs[(size_t)scan * P.max_keypoints + B.row_map[row].y] = FX_NONE;
    }
    desc_fill_nan(out, tid, NT);
    __syncthreads();
    return;
  }
  float *sw = reinterpret_cast<float *>(keys);
#pragma unroll
  for (int u = 0; u < PER; ++u)
    if (dst[u] != FX_NONE) sw[dst[u]] = wgt[u];
  __syncthreads();
  // one lane per bin adds the bin's weights in order (the row was cleared by k_desc_group)
  for (uint32_t bin = tid; bin < FX_DESC_BINS; bin += NT) {
    const uint32_t s0 = bin ? bin_end[bin - 1u] : 0u, s1 = bin_end[bin];
    if (s0 == s1) continue;
    float acc = 0.0f;
#pragma unroll 4
    for (uint32_t qq = s0; qq < s1; ++qq) acc += sw[qq];
    out[bin] = acc;
  }
  __syncthreads();
  FX_STAMP(53);
  if (tid == 0) {
    FX_COUNT(54, 1);
    FX_COUNT(55, nM);
    FX_COUNT(56, nS);
  }
}
// Every row of the tier by ONE kernel shape, the largest first (1024 threads, the LDS of a CU).  Until round 6 rows of up to 4096
// binned neighbours had a kernel of their own (256 threads, three workgroups a CU) launched before this one: a row is a chain of a
// dozen barrier-separated steps whose length hardly depends on the row — a quarter of the lanes made it longer, and the second
// launch waited for the first: one launch is 3 % (config 3, 2528 rows) to 10 % (10 240 small rows) shorter a batch
// (profiles/r06_experiments.md §11).
template <int KMAX, int NT>
__device__ __forceinline__ void dense_finish_loop(const FxDevParams &P, const FxBuffers &B) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  if (B.counters[6] == 0u) return;  // no dense rows in this batch
  uint32_t *tl = smem + 2 * KMAX + KMAX / 2 + 1984;
  const FxScTables *T = tables_to_lds(B, tl);
  uint32_t *s_slot = tl + FX_TABLE_WORDS + 8;
  const unsigned long long seq = B.seq[0];
  while (true) {
    const uint32_t slot = dense_next(P, B, 0u, s_slot);
    if (slot == FX_NONE) break;
    const uint32_t nM = B.dense_nm[slot];
    if (nM == FX_NONE) continue;  // failed row (flagged)
    const uint32_t row = B.dense_rows[slot];
    const uint2 rm = B.row_map[row];
    if (B.kp_nbrs[(size_t)rm.x * P.max_keypoints + rm.y] == 0u) {  // no point within R: NaN descriptor, no RNG draw (A.8-3)
      desc_fill_nan(B.desc + (size_t)row * FX_DESC_FLOATS, threadIdx.x, NT);
      continue;
    }
    if (nM == 0) continue;  // (only the keypoint's own point: the cleared row is the descriptor)
    dense_finish_row<KMAX, NT>(P, B, slot, smem, T, seq);  // (more keys than KMAX or P.dense_lds_keys — tests lower it —: sorted in the key pool)
  }
}
extern "C" __global__ __launch_bounds__(FX_DFIN_T, FX_DFIN_OCC) void k_dense_finish(FxDevParams P, FxBuffers B) {
  dense_finish_loop<FX_DFIN_K, FX_DFIN_T>(P, B);
}

// RNG ordinals when several workgroups of k_gather shared a scan (small batches): 3DSC draws its three numbers only for
// keypoints that have neighbours; k_gather left a flag per keypoint in kp_nbrs (the descriptor kernels then store the counts).
extern "C" __global__ __launch_bounds__(FX_WG) void k_rng_ord(FxDevParams P, FxBuffers B, uint32_t batch) {
  // one wavefront per scan: ordinal of keypoint k = number of earlier keypoints that have neighbours
  const uint32_t b = (blockIdx.x * FX_WG + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (b >= batch) return;
  uint32_t K = B.n_kp[b];
  const uint32_t row0 = B.kp_offset[b];
  if (row0 >= P.max_total_kp) return;
  if (row0 + K > P.max_total_kp) K = P.max_total_kp - row0;
  uint32_t base = 0;
  for (uint32_t k0 = 0; k0 < K; k0 += 64) {
    const uint32_t k = k0 + lane;
    const unsigned long long m = __ballot(k < K && B.kp_nbrs[(size_t)b * P.max_keypoints + k] != 0u);
    if (k < K) B.row_xa[row0 + k] = B.xaxis[base + lanes_below(m)];
    base += (uint32_t)__popcll(m);
  }
}

// pcl::concatenateFields(keypoints, descriptors) -> pcl::PointDescriptor records (ref: node.cpp:119).
extern "C" __global__ __launch_bounds__(FX_WG) void k_pack_features(FxDevParams P, FxBuffers B, uint32_t batch,
                                                                     uint8_t *dst, uint32_t capacity) {
  uint32_t total = B.kp_offset[batch];
  if (total > P.max_total_kp) total = P.max_total_kp;
  if (total > capacity) total = capacity;
  for (uint32_t w = blockIdx.x; w < total; w += gridDim.x) {
    const uint32_t scan = scan_of_row(B.kp_offset, batch, w);
    const uint32_t k = w - B.kp_offset[scan];
    const float4 kp = B.keypoints[(size_t)scan * P.max_keypoints + k];
    float *rec = reinterpret_cast<float *>(dst + (size_t)w * FX_FEATURE_RECORD_BYTES);
    const float *src = B.desc + (size_t)w * FX_DESC_FLOATS;
    if (threadIdx.x == 0) {
      rec[0] = kp.x;
      rec[1] = kp.y;
      rec[2] = kp.z;
      rec[3] = 1.0f;  // PCL_ADD_POINT4D padding word
      rec[4] = kp.w;
    }
    for (uint32_t t = threadIdx.x; t < FX_DESC_FLOATS; t += FX_WG) rec[5 + t] = src[t];
    if (threadIdx.x < 2) rec[5 + FX_DESC_FLOATS + threadIdx.x] = 0.0f;  // tail padding to 7984 B
  }
}

// Fixed-stride keypoint records for the cross-GPU gather: {n_kp, flags, 0, 0, kp[rec_kp] float4}.
extern "C" __global__ __launch_bounds__(FX_WG) void k_pack_kp_records(FxDevParams P, FxBuffers B, uint32_t batch,
                                                                       float4 *dst, uint32_t rec_kp) {
  const uint32_t scan = blockIdx.x;
  if (scan >= batch) return;
  float4 *rec = dst + (size_t)scan * (rec_kp + 1);
  const uint32_t K = B.n_kp[scan];
  if (threadIdx.x == 0) {
    uint4 h = make_uint4(K < rec_kp ? K : rec_kp, B.flags[scan] | (K > rec_kp ? FX_FLAG_KP_OVERFLOW : 0u), 0u, 0u);
    rec[0] = *reinterpret_cast<float4 *>(&h);
  }
  const float4 *kp = B.keypoints + (size_t)scan * P.max_keypoints;
  for (uint32_t k = threadIdx.x; k < rec_kp; k += FX_WG) rec[1 + k] = k < K ? kp[k] : make_float4(0, 0, 0, 0);
}

// The same keypoints as ONE compact block per batch — what crosses GPUs: a batch of VLP-16 scans has 54
// keypoints a scan where the fixed-stride records above reserve max_keypoints (256): 4.2 MB a rank and step where 0.9 are
// keypoints.  Block = float4 rows: row 0 {scans, keypoints stored, OR of the flags, max_total} (u32); then kp_offset[max_scans + 1]
// (u32, four a row: scan b's keypoints are rows [kp_offset[b], kp_offset[b + 1]) of the keypoint area; entries beyond the
// batch repeat the total); then flags[max_scans] (u32, four a row); then max_total keypoint rows (x, y, z, elevation), packed in
// scan order, zero beyond the total.  A batch with more keypoints than max_total is cut there: the scans that lose keypoints
// carry FX_FLAG_KP_OVERFLOW.  The size is fixed by (max_scans, max_total): every rank hands the collective the same count.
__host__ __device__ inline uint32_t kp_block_off_rows(uint32_t max_scans) { return (max_scans + 1u + 3u) / 4u; }
__host__ __device__ inline uint32_t kp_block_flag_rows(uint32_t max_scans) { return (max_scans + 3u) / 4u; }
extern "C" __global__ __launch_bounds__(FX_WG) void k_pack_kp_block(FxDevParams P, FxBuffers B, uint32_t batch, uint32_t *dst, uint32_t max_scans,
                                                                     uint32_t max_total) {
  __shared__ uint32_t s_or[FX_NWAVE];
  uint32_t *off = dst + 4, *flg = off + 4u * kp_block_off_rows(max_scans);
  float4 *kp = reinterpret_cast<float4 *>(flg + 4u * kp_block_flag_rows(max_scans));
  const uint32_t tid = threadIdx.x, nb = min(batch, max_scans);
  const uint32_t total = min(B.kp_offset[nb], max_total);
  if (blockIdx.x == 0) {  // the header rows
    uint32_t acc = 0;
    for (uint32_t i = tid; i < 4u * kp_block_off_rows(max_scans); i += FX_WG) off[i] = min(B.kp_offset[min(i, nb)], max_total);
    for (uint32_t i = tid; i < 4u * kp_block_flag_rows(max_scans); i += FX_WG) {
      uint32_t f = 0;
      if (i < nb) {  // (cut: the scan keeps fewer keypoints in the block than it has)
        const uint32_t o0 = B.kp_offset[i], o1 = B.kp_offset[i + 1u];
        f = B.flags[i] | (min(o1, max_total) - min(o0, max_total) < o1 - o0 ? FX_FLAG_KP_OVERFLOW : 0u);
      }
      flg[i] = f;
      acc |= f;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc |= (uint32_t)__shfl_xor((int)acc, d, 64);
    if ((tid & 63u) == 0u) s_or[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      uint32_t o = batch > max_scans ? FX_FLAG_KP_OVERFLOW : 0u;  // (a batch beyond the block's scan capacity: cut, and said so)
      for (uint32_t w = 0; w < FX_NWAVE; ++w) o |= s_or[w];
      dst[0] = nb, dst[1] = total, dst[2] = o, dst[3] = max_total;
    }
  }
  for (uint32_t scan = blockIdx.x; scan < nb; scan += gridDim.x) {
    const uint32_t o0 = min(B.kp_offset[scan], max_total), o1 = min(B.kp_offset[scan + 1u], max_total);
    const float4 *src = B.keypoints + (size_t)scan * P.max_keypoints;
    for (uint32_t k = tid; k < o1 - o0; k += FX_WG) kp[o0 + k] = src[k];
  }
  for (uint32_t k = total + blockIdx.x * FX_WG + tid; k < max_total; k += gridDim.x * FX_WG) kp[k] = make_float4(0, 0, 0, 0);
}

// ====================================================================== PointCloud2 wire formats
// Ingress: pcl::fromPCLPointCloud2 (ref: node.cpp:79-81) picks the float32 fields x, y, z by name
// out of point_step-byte records (velodyne driver clouds carry extra fields such as `ring`, and
// point_step need not be a multiple of 4) — here as a byte-offset gather into packed float4.
extern "C" __global__ __launch_bounds__(FX_WG) void k_unpack_pc2(const uint8_t *src, uint32_t n, uint32_t point_step,
                                                                  uint32_t off_x, uint32_t off_y, uint32_t off_z,
                                                                  uint32_t off_i, uint32_t big_endian, float4 *dst) {
  for (uint32_t i = blockIdx.x * FX_WG + threadIdx.x; i < n; i += gridDim.x * FX_WG) {
    const uint8_t *rec = src + (size_t)i * point_step;
    auto rd = [&](uint32_t off) -> float {
      uint32_t u = (uint32_t)rec[off] | ((uint32_t)rec[off + 1] << 8) | ((uint32_t)rec[off + 2] << 16) |
                   ((uint32_t)rec[off + 3] << 24);
      if (big_endian) u = __builtin_bswap32(u);
      return __uint_as_float(u);
    };
    dst[i] = make_float4(rd(off_x), rd(off_y), rd(off_z), off_i != 0xffffffffu ? rd(off_i) : 0.0f);
  }
}
// Egress: a pcl::PointCloud<pcl::PointXYZI> as pcl_ros serialises it — 32-byte records, x@0 y@4 z@8
// (pad 1.0f @12) intensity@16 — from the context's float4 (x, y, z, intensity) arrays
// (ref: node.cpp:129-139 publishes keypoints, keypoint_cloud and cloud this way).
extern "C" __global__ __launch_bounds__(FX_WG) void k_pack_xyzi32(const float4 *src, uint32_t n, float *dst) {
  for (uint32_t i = blockIdx.x * FX_WG + threadIdx.x; i < n; i += gridDim.x * FX_WG) {
    const float4 v = src[i];
    float4 *o = reinterpret_cast<float4 *>(dst + (size_t)i * 8);
    o[0] = make_float4(v.x, v.y, v.z, 1.0f);
    o[1] = make_float4(v.w, 0.0f, 0.0f, 0.0f);
  }
}

// ====================================================================== launchers
#ifdef FX_TEST_HOOKS
// Test hook (fx_test_sort_replay_device): the cluster-order replay exactly as cc_order runs it, on
// arbitrary size sequences; one 64-thread workgroup per sequence, n <= 192.
extern "C" __global__ __launch_bounds__(64) void k_test_sort_replay(const uint32_t *sizes, uint32_t n, uint32_t *perm) {
  __shared__ uint32_t crec[192], tmp[192], stk[FX_SORT_STACK_WORDS];
  const uint32_t *src = sizes + (size_t)blockIdx.x * n;
  for (uint32_t c = threadIdx.x; c < n; c += 64) crec[c] = (src[c] << 16) | c;
  __syncthreads();
  uint16_t *pos = reinterpret_cast<uint16_t *>(tmp);
  if (n <= 64)
    sort_partition_wave<1>(crec, (int)n, (int *)stk, pos, pos + n);
  else if (n <= 128)
    sort_partition_wave<2>(crec, (int)n, (int *)stk, pos, pos + n);
  else
    sort_partition_wave<3>(crec, (int)n, (int *)stk, pos, pos + n);
  __syncthreads();
  for (uint32_t c = threadIdx.x; c < n; c += 64) {
    const uint32_t rec = crec[c], sz = rec >> 16;
    uint32_t at = 0;
    for (uint32_t d = 0; d < n; ++d) {
      const uint32_t sd = crec[d] >> 16;
      at += (sd > sz || (sd == sz && d < c)) ? 1u : 0u;
    }
    tmp[at] = rec;
  }
  __syncthreads();
  for (uint32_t c = threadIdx.x; c < n; c += 64) perm[(size_t)blockIdx.x * n + c] = tmp[c] & 0xffffu;
}

#endif  // FX_TEST_HOOKS

extern "C" {

size_t fxk_ring_large_lds_bytes(uint32_t cap, uint32_t ccap) {
  return (size_t)(SegCfg<FX_RING_LARGE_T>::kWords + FX_RING_WORDS_PER_POINT * cap + FX_RING_WORDS_PER_CLUSTER * ccap) * 4;
}
void fxk_rings_large(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t cap, uint32_t ccap, uint32_t grid,
                     uint32_t after_runs2) {
  hipLaunchKernelGGL(k_rings_large, dim3(grid), dim3(FX_RING_LARGE_T), fxk_ring_large_lds_bytes(cap, ccap), s, P, B, cap, ccap, after_runs2);
}
size_t fxk_merge_lds_bytes(uint32_t cap, uint32_t n_rings) { return merge_words(cap, cap, n_rings, true) * 4; }
size_t fxk_merge_huge_lds_bytes(uint32_t cap, uint32_t ccap, uint32_t n_rings) { return merge_words(cap, ccap, n_rings, false) * 4; }
size_t fxk_gather_lds_bytes(uint32_t max_keypoints) { return (size_t)gather_words(std::min(max_keypoints, (uint32_t)FX_GATHER_KCAP), FX_GATHER_WIDE_T) * 4; }
size_t fxk_dense_finish_lds_bytes(void) {
  const size_t k = FX_DFIN_K;
  return (2 * k + k / 2 + 1984 + FX_TABLE_WORDS + 16) * 4;
}
uint32_t fxk_dense_cells(void) { return FX_DCELLS; }
// build parameters of this translation unit the host sizes buffers by (a build with other values must not outrun them)
uint32_t fxk_group_cap(void) { return FX_GROUP_CAP; }  // bins k_desc_group records per row (FxBuffers::desc_bins)
uint32_t fxk_dfin_k(void) { return FX_DFIN_K; }        // binned neighbours k_dense_finish sorts in LDS
size_t fxk_desc_lds_bytes(uint32_t cap) { return (size_t)(16 + FX_DESC_WORDS_PER_POINT * cap + FX_DESC_BINS) * 4; }

hipError_t fxk_configure(size_t ring_big, size_t merge_big, size_t merge_huge, size_t desc_big, size_t gather) {
  hipError_t e;
  e = hipFuncSetAttribute((const void *)k_merge_huge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)merge_huge);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)k_merge_huge_a, hipFuncAttributeMaxDynamicSharedMemorySize, (int)merge_huge);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)k_merge_huge_b, hipFuncAttributeMaxDynamicSharedMemorySize, (int)merge_huge);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)k_merge_huge_c, hipFuncAttributeMaxDynamicSharedMemorySize, (int)merge_huge);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)k_gather, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gather);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)k_gather_wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gather);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)k_gather_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gather);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)k_gather_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gather);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)k_gather_passes, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gather);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)k_rings_large, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring_big);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)k_merge_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)merge_big);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)k_desc_mid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)desc_big);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)k_dense_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fxk_dense_finish_lds_bytes());
  return e;
}

uint32_t fxk_near_words(uint32_t max_points) { return (max_points + FX_PREP_TILE - 1) / FX_PREP_TILE * (FX_PREP_TILE / 128); }  // one bit per 4 points
void fxk_prep(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, float near_margin, float el0, float inv_step,
              uint32_t clk_slot) {
  hipLaunchKernelGGL(k_prep, dim3(batch), dim3(FX_PREP_T), 0, s, P, B, near_margin, el0, inv_step, clk_slot);
}
// several workgroups a scan for batches that would otherwise leave most of the chip idle (see k_prep_count)
#define FX_PREP_SLICES_MAX 16u
uint32_t fxk_prep_slices_max(void) { return FX_PREP_SLICES_MAX; }
void fxk_prep_sliced(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, uint32_t slices, float near_margin, float el0, float inv_step,
                     uint32_t clk_slot) {
  hipLaunchKernelGGL(k_prep_count, dim3(slices, batch), dim3(FX_PREP_T), 0, s, P, B);
  hipLaunchKernelGGL(k_prep_sliced, dim3(slices, batch), dim3(FX_PREP_T), 0, s, P, B, near_margin, el0, inv_step, clk_slot);
}
void fxk_bucket_sliced(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, uint32_t slices, float el0, float inv_step,
                       uint32_t clk_next) {
  const size_t lds = (48 + (size_t)P.n_rings * (2 + FX_BUCKET_NW) + 1) * 4;
  hipLaunchKernelGGL(k_bucket_sliced, dim3(slices, batch), dim3(FX_BUCKET_T), lds, s, P, B, el0, inv_step, clk_next);
}
void fxk_bucket(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, float el0, float inv_step,
                uint32_t clk_next) {
  const size_t lds = (48 + (size_t)P.n_rings * (2 + FX_BUCKET_NW) + 1) * 4;
  if (P.n_rings > 24)
    hipLaunchKernelGGL(k_bucket_many, dim3(batch), dim3(FX_BUCKET_T), lds, s, P, B, el0, inv_step, clk_next);
  else
    hipLaunchKernelGGL(k_bucket, dim3(batch), dim3(FX_BUCKET_T), lds, s, P, B, el0, inv_step, clk_next);
}
size_t fxk_ring_runs_lds_bytes(void) { return (size_t)rr_words<FX_RR_S, FX_RR_RN>() * 4; }
void fxk_rings_runs2(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t max_pts, uint32_t grid) {
  constexpr size_t lds = (size_t)rr_words<384, 256>() * 4;
  hipLaunchKernelGGL(k_rings_runs2, dim3(grid), dim3(64), lds, s, P, B, max_pts);
}
void fxk_rings_runs(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, uint32_t max_pts, uint32_t grid) {
  const uint32_t n_items = batch * (uint32_t)P.n_rings;
  if (grid > n_items) grid = (n_items + 7) / 8 * 8;
  hipLaunchKernelGGL(k_rings_runs, dim3(grid), dim3(64), fxk_ring_runs_lds_bytes(), s, P, B, max_pts, n_items);
}
void fxk_merge_small(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, uint32_t cap) {
  hipLaunchKernelGGL(k_merge_small, dim3(batch), dim3(FX_MSMALL_T), fxk_merge_lds_bytes(cap, P.n_rings), s, P, B, cap);
}
void fxk_merge_big(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t cap, uint32_t grid, uint32_t last) {
  hipLaunchKernelGGL(k_merge_big, dim3(grid), dim3(FX_MBIG_T), fxk_merge_lds_bytes(cap, P.n_rings), s, P, B, cap, last);
}
void fxk_merge_huge(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t cap, uint32_t ccap, uint32_t grid) {
  hipLaunchKernelGGL(k_merge_huge, dim3(grid), dim3(FX_MBIG_T), fxk_merge_huge_lds_bytes(cap, ccap, P.n_rings), s, P, B, cap, ccap);
}
// the same as three launches, `slices` workgroups a scan in the pair loop (needs FxBuffers::merge_hp)
size_t fxk_merge_hp_words(uint32_t cap) { return merge_hp_words(cap); }
uint32_t fxk_merge_slices_max(void) { return FX_MERGE_SLICES; }
void fxk_merge_huge_split(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t cap, uint32_t ccap, uint32_t grid, uint32_t slices) {
  const size_t lds = fxk_merge_huge_lds_bytes(cap, ccap, P.n_rings);
  hipLaunchKernelGGL(k_merge_huge_a, dim3(grid), dim3(FX_MBIG_T), lds, s, P, B, cap, ccap);
  hipLaunchKernelGGL(k_merge_huge_b, dim3(std::min(slices, (uint32_t)FX_MERGE_SLICES), grid), dim3(FX_MBIG_T), lds, s, P, B, cap, ccap);
  hipLaunchKernelGGL(k_merge_huge_c, dim3(grid), dim3(FX_MBIG_T), lds, s, P, B, cap, ccap);
}
// the fused front kernel takes sensors of up to FX_FRONT_RMAX rings (scans that do not fit its tables go on to k_front_redo)
uint32_t fxk_front_max_rings(void) { return FX_FRONT_RMAX; }
uint32_t fxk_front_merge_cap(void) { return FX_FRONT_MERGE; }
void fxk_front(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, float near_margin, float el0, float inv_step,
               uint32_t clk_slot, uint32_t merge_cap, uint32_t force_redo) {
  hipLaunchKernelGGL(k_front, dim3(batch), dim3(FX_FRONT_T), front_lds_bytes(), s, P, B, near_margin, el0, inv_step, clk_slot, merge_cap, force_redo);
}
void fxk_front_redo(hipStream_t s, const FxDevParams &P, const FxBuffers &B, float el0, float inv_step, uint32_t huge_ccap, uint32_t force_slow,
                    uint32_t grid) {
  hipLaunchKernelGGL(k_front_redo, dim3(grid), dim3(FX_FRONT_T), front_lds_bytes(), s, P, B, el0, inv_step, huge_ccap, force_slow);
}
// scratch words of one workgroup of the slow tier; its LDS (the bodies' scratch words in front, the merge's ring bases)
size_t fxk_slow_words(uint32_t max_ring_points, uint32_t max_candidates, uint32_t huge_ccap) {
  return (std::max(slow_ring_words(max_ring_points), slow_merge_words(max_candidates, huge_ccap)) + 3) & ~(size_t)3;
}
// (also computes the batch's keypoint offsets: its last workgroup, see k_slow)
void fxk_slow(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t huge_ccap, uint32_t grid, uint32_t batch, uint32_t clk_next) {
  const size_t lds = std::max((size_t)SegCfg<FX_SLOW_T>::kWords, (size_t)FX_MERGE_HEAD + ((2 * ((size_t)P.n_rings + 1) + 3) & ~(size_t)3)) * 4;
  const uint32_t g = std::max(1u, std::min(grid, P.gs_slots));
  hipLaunchKernelGGL(k_slow, dim3(g), dim3(FX_SLOW_T), lds, s, P, B, huge_ccap, batch, clk_next);
}
void fxk_front_ab(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, float near_margin, float el0, float inv_step,
                  uint32_t clk_slot, uint32_t force_redo) {
  hipLaunchKernelGGL(k_front_ab, dim3(batch), dim3(FX_FRONT_T), front_ab_lds_bytes(), s, P, B, near_margin, el0, inv_step, clk_slot, force_redo);
}
void fxk_front_cd(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, uint32_t clk_slot, uint32_t merge_cap, uint32_t lean,
                  uint32_t self_n, uint32_t force_redo) {
  if (lean && !self_n)
    hipLaunchKernelGGL(k_front_cdl, dim3(batch), dim3(FX_FRONT_CDL_T), front_lean_lds_bytes(), s, P, B, clk_slot, merge_cap);
  else
    hipLaunchKernelGGL(k_front_cd, dim3(batch), dim3(FX_FRONT_T), front_lds_bytes(), s, P, B, clk_slot, merge_cap, self_n, force_redo);
}
hipError_t fxk_configure_front(void) {
  hipError_t e = hipFuncSetAttribute((const void *)k_front, hipFuncAttributeMaxDynamicSharedMemorySize, (int)front_lds_bytes());
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)k_front_ab, hipFuncAttributeMaxDynamicSharedMemorySize, (int)front_ab_lds_bytes());
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)k_front_cd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)front_lds_bytes());
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void *)k_front_redo, hipFuncAttributeMaxDynamicSharedMemorySize, (int)front_lds_bytes());
}
// one workgroup per scan when the batch fills the GPU anyway (it is then the only writer of the scan's lists: no
// global atomics, and it settles the RNG ordinals itself), several when it does not (streaming)
#ifndef FX_GATHER_SLICES
#define FX_GATHER_SLICES 1u
#endif
// One workgroup per scan (the sole writer of the scan's lists: no global atomics) from 64 scans on — 256 threads when
// the batch alone fills the GPU, 1024 (k_gather_wide) when it does not; only a handful of scans (streaming) is split
// over several workgroups per scan.
uint32_t fxk_gather_slices(uint32_t batch) { return batch >= 64u ? FX_GATHER_SLICES : (batch >= 16u ? 4u : 16u); }
#define FX_GATHER_COUNTED_MAX 16u
uint32_t fxk_gather_counted_max(void) { return FX_GATHER_COUNTED_MAX; }
// counted: > 1 = that many workgroups a scan in two launches (a counting pass, then the scatter with every slice's list
// positions known: gather_body's MODE) — batches of few big scans; needs FxBuffers::gather_cnt.  Returns the workgroups a scan
// the lists were written by (> 1: the RNG ordinals are k_rng_ord's).
uint32_t fxk_gather(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, float box_margin, uint32_t counted) {
  const uint32_t slices = fxk_gather_slices(batch);
  const size_t lds = (size_t)gather_words(std::min(P.max_keypoints, (uint32_t)FX_GATHER_KCAP), FX_GATHER_T) * 4;
  if (P.max_keypoints > FX_GATHER_KCAP) {
    hipLaunchKernelGGL(k_gather_passes, dim3(slices, batch), dim3(FX_GATHER_WIDE_T), (size_t)gather_words(FX_GATHER_KCAP, FX_GATHER_WIDE_T) * 4, s, P, B, box_margin);
  } else if (counted > 1u && B.gather_cnt) {
    counted = std::min(counted, (uint32_t)FX_GATHER_COUNTED_MAX);
    hipLaunchKernelGGL(k_gather_count, dim3(counted, batch), dim3(FX_GATHER_T), lds, s, P, B, box_margin);
    hipLaunchKernelGGL(k_gather_scatter, dim3(counted, batch), dim3(FX_GATHER_T), lds, s, P, B, box_margin);
    return counted;
  } else if (slices == 1 && batch < 1024u) {
    hipLaunchKernelGGL(k_gather_wide, dim3(1, batch), dim3(FX_GATHER_WIDE_T), (size_t)gather_words(std::min(P.max_keypoints, (uint32_t)FX_GATHER_KCAP), FX_GATHER_WIDE_T) * 4, s, P, B, box_margin);
  } else {
    hipLaunchKernelGGL(k_gather, dim3(slices, batch), dim3(FX_GATHER_T), lds, s, P, B, box_margin);
  }
  return slices;
}
void fxk_desc_group(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, uint32_t grid) {
  hipLaunchKernelGGL(k_desc_group, dim3(grid), dim3(FX_WG), (size_t)(FX_NWAVE * FX_GROUPS * FX_GROUP_WORDS + FX_TABLE_WORDS) * 4, s, P, B, batch);
}
void fxk_desc_mid(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, uint32_t cap, uint32_t n_wg,
                  uint32_t n_wave, uint32_t n_dslow) {
  const size_t lds_wave = (size_t)(FX_NWAVE * FX_WAVE_WORDS + FX_TABLE_WORDS) * 4, lds_wg = fxk_desc_lds_bytes(cap);
  static_assert((16 + FX_DESC_BINS + 16) * 4 <= (FX_NWAVE * FX_WAVE_WORDS + FX_TABLE_WORDS) * 4, "the dense tier's rows in k_desc_mid's LDS");
  n_dslow = std::min(n_dslow, P.gsd_slots);  // (a scratch region each)
  hipLaunchKernelGGL(k_desc_mid, dim3(n_wg + n_wave + n_dslow), dim3(FX_WG), lds_wave > lds_wg ? lds_wave : lds_wg, s, P, B, batch, cap, n_wg, n_dslow);
}
size_t fxk_dense_slow_words(uint32_t max_points) { return ((size_t)FX_DESC_WORDS_PER_POINT * max_points + 3) & ~(size_t)3; }
// (rows and items are taken by ticket: any grid is correct; the full ones are what is resident at once)
void fxk_dense(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t n_cu, uint32_t rows, uint32_t items, uint32_t skip) {
  auto grid = [](uint32_t want, uint32_t full) { return want < full ? (want ? want : 1u) : full; };
  hipLaunchKernelGGL(k_dense_sort, dim3(grid(rows, n_cu * FX_DSORT_PER_CU)), dim3(FX_DSORT_T), 0, s, P, B);
  // (skip: measurement only — a test build's FX_SKIP_EMPTY bits 2 / 3: the tier's share of a step with batches in flight)
  if (!(skip & 2u)) hipLaunchKernelGGL(k_dense_density, dim3(grid(items, n_cu * FX_DDENS_PER_CU)), dim3(FX_DDENS_T), 0, s, P, B);
  if (!(skip & 1u)) hipLaunchKernelGGL(k_dense_finish, dim3(grid(rows, n_cu)), dim3(FX_DFIN_T), fxk_dense_finish_lds_bytes(), s, P, B);
}
#ifdef FX_TEST_HOOKS
extern "C" __global__ __launch_bounds__(FX_WG) void k_test_elevation(const float *xyz, uint32_t n, const double *tab, float *fast, uint8_t *ok,
                                                                     float *exact) {
  __shared__ double s_atan[(FX_ATAN_N + 1) * (FX_ATAN_DEG + 1)];
  for (uint32_t r = threadIdx.x; r < (FX_ATAN_N + 1) * (FX_ATAN_DEG + 1); r += FX_WG) s_atan[r] = tab[r];
  __syncthreads();
  for (uint32_t i = blockIdx.x * FX_WG + threadIdx.x; i < n; i += gridDim.x * FX_WG) {
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    float f = 0.f;
    ok[i] = elevation_fast(x, y, z, s_atan, f) ? 1 : 0;
    fast[i] = f;
    exact[i] = elevation_deg(x, y, z);
  }
}
// Test hook: WithinR2::count (packed, fma + clamp) and the plain compare it replaces, one query per workgroup against n
// support points staged through LDS in chunks (any range start / length parity: the chunk bounds vary with the query).
extern "C" __global__ __launch_bounds__(FX_WG) void k_test_within(const float4 *sp, uint32_t n, const float4 *queries, float r2,
                                                                  uint32_t *packed, uint32_t *plain) {
  __shared__ float4 s_pts[1024];
  __shared__ uint32_t s_cnt[2];
  const float4 b = queries[blockIdx.x];
  const WithinR2 within(r2);
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  for (uint32_t c0 = 0; c0 < n; c0 += 1024) {
    const uint32_t m = min(1024u, n - c0);
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < m; t += FX_WG) s_pts[t] = sp[c0 + t];
    __syncthreads();
    // every thread a range of its own, odd starts and lengths included
    const uint32_t per = (m + FX_WG - 1) / FX_WG + (blockIdx.x & 3u);
    const uint32_t q0 = min(threadIdx.x * per, m), q1 = min(q0 + per, m);
    uint32_t a = within.count(s_pts, q0, q1, b.x, b.y, b.z), c = 0;
    for (uint32_t q = q0; q < q1; ++q) c += dist2(b.x, b.y, b.z, s_pts[q].x, s_pts[q].y, s_pts[q].z) < r2 ? 1u : 0u;
    // (ranges beyond FX_WG * per are nobody's when per was rounded up by the block's offset: the tail goes to thread 0)
    if (threadIdx.x == 0 && FX_WG * per < m) {
      a += within.count(s_pts, FX_WG * per, m, b.x, b.y, b.z);
      for (uint32_t q = FX_WG * per; q < m; ++q) c += dist2(b.x, b.y, b.z, s_pts[q].x, s_pts[q].y, s_pts[q].z) < r2 ? 1u : 0u;
    }
    if (a) atomicAdd(&s_cnt[0], a);
    if (c) atomicAdd(&s_cnt[1], c);
  }
  __syncthreads();
  if (threadIdx.x == 0) packed[blockIdx.x] = s_cnt[0], plain[blockIdx.x] = s_cnt[1];
}
void fxk_test_within(hipStream_t s, const float4 *sp, uint32_t n, const float4 *queries, uint32_t nq, float r2, uint32_t *packed, uint32_t *plain) {
  hipLaunchKernelGGL(k_test_within, dim3(nq), dim3(FX_WG), 0, s, sp, n, queries, r2, packed, plain);
}
void fxk_test_elevation(hipStream_t s, const float *xyz, uint32_t n, const double *tab, float *fast, uint8_t *ok, float *exact) {
  hipLaunchKernelGGL(k_test_elevation, dim3(1024), dim3(FX_WG), 0, s, xyz, n, tab, fast, ok, exact);
}
void fxk_test_sort_replay(hipStream_t s, const uint32_t *sizes, uint32_t n_seq, uint32_t n, uint32_t *perm) {
  hipLaunchKernelGGL(k_test_sort_replay, dim3(n_seq), dim3(64), 0, s, sizes, n, perm);
}
#endif  // FX_TEST_HOOKS
void fxk_rng_ord(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch) {
  hipLaunchKernelGGL(k_rng_ord, dim3((batch + FX_NWAVE - 1) / FX_NWAVE), dim3(FX_WG), 0, s, P, B, batch);
}
void fxk_pack_kp_records(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, void *dst,
                         uint32_t rec_kp) {
  hipLaunchKernelGGL(k_pack_kp_records, dim3(batch), dim3(FX_WG), 0, s, P, B, batch, (float4 *)dst, rec_kp);
}
size_t fxk_kp_block_bytes(uint32_t max_scans, uint32_t max_total) {
  return ((size_t)1 + kp_block_off_rows(max_scans) + kp_block_flag_rows(max_scans) + max_total) * 16;
}
void fxk_pack_kp_block(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, void *dst, uint32_t max_scans, uint32_t max_total,
                       uint32_t grid) {
  hipLaunchKernelGGL(k_pack_kp_block, dim3(grid), dim3(FX_WG), 0, s, P, B, batch, (uint32_t *)dst, max_scans, max_total);
}
void fxk_unpack_pc2(hipStream_t s, const void *src, uint32_t n, uint32_t point_step, uint32_t ox, uint32_t oy, uint32_t oz,
                    uint32_t oi, uint32_t big_endian, void *dst, uint32_t grid) {
  hipLaunchKernelGGL(k_unpack_pc2, dim3(grid), dim3(FX_WG), 0, s, (const uint8_t *)src, n, point_step, ox, oy, oz, oi,
                     big_endian, (float4 *)dst);
}
void fxk_pack_xyzi32(hipStream_t s, const void *src, uint32_t n, void *dst, uint32_t grid) {
  hipLaunchKernelGGL(k_pack_xyzi32, dim3(grid), dim3(FX_WG), 0, s, (const float4 *)src, n, (float *)dst);
}
void fxk_pack_features(hipStream_t s, const FxDevParams &P, const FxBuffers &B, uint32_t batch, void *dst,
                       uint32_t capacity, uint32_t grid) {
  hipLaunchKernelGGL(k_pack_features, dim3(grid), dim3(FX_WG), 0, s, P, B, batch, (uint8_t *)dst, capacity);
}

}  // extern "C"
