// fx_device.h — structures shared by the host driver and the gfx950 kernels.
#ifndef FX_DEVICE_H_
#define FX_DEVICE_H_
#include <stdint.h>

// One scan of the batch as the kernels see it.
struct FxScanMeta {
  const float *pts;   // device pointer, records of stride_f floats, x y z at 0 1 2
  uint32_t n;         // points in the scan
  uint32_t stride_f;  // record stride in floats (4 or 8)
  float R[9];         // rotateCloud matrix, row major (ref: node.cpp:161-165)
  uint32_t pad_;
};

// Per-context constants (narrowed exactly where PCL narrows them, see fx_api.cpp).
struct FxDevParams {
  // filterCloud limits as PassThrough stores them (float) (ref: node.cpp:169-183)
  float x_min, x_max, y_min, y_max, z_min, z_max;
  int32_t n_rings;
  // getCylinderSegments (ref: node.cpp:269-276, 314-316)
  float r2_cluster;
  uint32_t min_count, max_count;
  double gate_diameter;  // 2 * cluster_radius_threshold
  // secondary merge (ref: node.cpp:217, 222-229)
  double crt;
  float r2_merge;
  uint32_t ndc, secondary_max;
  // 3DSC (ref: node.cpp:350-352)
  float r2_search, r2_density, r2_support;
  int32_t estimate_descriptors;
  // capacities
  uint32_t max_points, max_ring_cands, max_candidates, max_keypoints, max_total_kp, max_kpc, max_neighbors,
      max_ring_points, list_cap, ring_slot_cap;
  uint32_t ring_list_cap;  // entries per XCD class of the deferred-ring lists
  uint32_t near_words;  // words of near bits per scan: one bit per 4 consecutive points = one 64-byte sector (k_prep -> k_gather)
  // dense tier (k_dense_*): support sets beyond dense_min points and lists that overflowed list_cap
  uint32_t dense_min;       // rows with more support points than this take the dense tier (1024)
  uint32_t ovf_cap;         // entries of a scan's overflow region (list entries beyond list_cap, any row of the scan)
  uint32_t dense_cap;       // entries of the sorted pool (and of the key pool) per batch
  uint32_t max_dense_rows;  // rows of the dense-row list / cell tables
  uint32_t dense_qcap;      // entries of the query pool (every cell's queries padded to four: up to 4 per support point)
  uint32_t dense_lds_keys;  // binned neighbours k_dense_finish sorts in LDS (14336; tests lower it to reach the key pool)
  uint32_t dense_won_points;  // support points of a row whose query marks k_dense_sort keeps as a bit map in LDS (65536; tests lower it)
  // slow tier (k_slow): scratch regions in HBM for what exceeds every LDS-sized tier
  uint32_t gs_slots;  // regions (= the largest grid k_slow is launched with)
  uint32_t gs_words;  // words per region
  uint32_t gsd_slots, gsd_words;  // the same for dense_slow_loop (the dense descriptor tier's rows without its four launches)
};

// 3DSC tables in device memory (built on the host by fx_sc3d_tables / fx_sc3d_xaxis).
struct FxScTables {
  float radii[16];
  float theta[12];
  float phi[13];
  float lut[165];  // [k*15 + j]; identical for every azimuth bin
};

// Every device buffer of a context.
#define FX_CLK_SLOTS 64
#define FX_N_COUNTERS 48  // counters k_prep / k_front clear for the batch
#define FX_CNT_REDO 48    // counters[48]: scans k_front hands to k_front_redo, [49]: scans handed to the slow tier, k_slow (cleared by k_offsets, after their readers)
#define FX_CNT_SLOW_TICKET 50  // counters[50]: workgroups of k_slow's launch that are done (the last one computes the batch's keypoint offsets and puts it back to 0)
#define FX_N_COUNTER_WORDS 56
#define FX_CNT_RUNS2_TICKET 40  // counters[40 + c]: next ring of XCD class c's list for k_rings_runs2
#define FX_ATAN_N 64      // table step of k_prep's arctangent: 1 / 64 over [0, 1]
#define FX_ATAN_DEG 6     // degree of the expansion about a table point (|offset| <= 1 / 128: truncation below 2^-51)
#define FX_ROW_DIRTY 0xffffffffu
#define FX_N_HINTS 8      // tier_hint[]: 0 / 1 rings handed to the second run tier / the workgroup tier (largest XCD class), 2 big merges, 3 huge merges, 4 dense rows, 5 dense support points, 6 scans k_front handed to k_front_redo, 7 scans handed to the slow tier (k_slow)
#define FX_CNT_QPOOL 32   // counters[32]: entries of the dense tier's query pool in use
#define FX_CNT_LARGE2 16  // counters[16 + c]: rings of XCD class c the second run tier hands to the workgroup tier
#define FX_CNT_LARGE 24  // counters[24 + c]: ... to the large tier
struct FxBuffers {
  const FxScanMeta *meta;
  const float2 *ring_win;  // [n_rings] (lo, hi) as float, inclusive
  const double *atan_tab;  // [FX_ATAN_N + 1][FX_ATAN_DEG + 1]: Taylor coefficients of atan about i / FX_ATAN_N (k_prep's elevation)
  const FxScTables *tables;
  const float2 *xaxis;  // [max_keypoints]
  // stage 1
  float4 *filt;          // [B][max_points]
  uint32_t *n_filt;      // [B]
  uint32_t *near_bits;   // [B][near_words]  bit: some point of that sector (4 points) is within the descriptor stage's reach of the filter box
  uint32_t *prep_cnt;       // [B][S]     survivors of slice s of the scan (k_prep_count; S workgroups a scan: small batches of big scans)
  uint32_t *prep_ring_cnt;  // [B][S][n_rings]  their ring counts (k_prep_sliced -> k_bucket_sliced)
  // stage 2a: ring-major copy of the filtered cloud
  float4 *ring_pts;         // [B][ring_slot_cap]
  uint32_t *ring_off;       // [B][n_rings]
  uint32_t *ring_cnt;       // [B][n_rings]
  // stage 2b
  float4 *ring_cand;        // [B][n_rings][max_ring_cands]
  uint32_t *ring_cand_size; // same shape
  uint32_t *ring_cand_cnt;  // [B][n_rings]
  float4 *kpc_pool;         // [B][ring_slot_cap]  ring r's members start at ring_off[r]
  uint32_t *kpc_pool_cand;  // [B][ring_slot_cap]  ring-local candidate slot
  uint32_t *kpc_ring_cnt;   // [B][n_rings]
  // stage 3
  float4 *cand;           // [B][max_candidates]
  uint32_t *cand_size;    // [B][max_candidates]
  int32_t *cand_kp;       // [B][max_candidates]
  uint32_t *n_cand;       // [B]
  float4 *keypoints;      // [B][max_keypoints]
  uint32_t *kp_size;      // [B][max_keypoints]
  uint32_t *kp_nbrs;      // [B][max_keypoints]
  uint32_t *n_kp;         // [B]
  uint32_t *kp_offset;    // [B+1]
  float4 *kpc;            // [B][max_kpc]
  uint32_t *kpc_cand;     // [B][max_kpc]
  uint32_t *n_kpc;        // [B]
  // stage 5
  float *desc;            // [max_total_kp][1989]
  // What the previous batches left in each descriptor row (rows outlive a batch: the buffer is zeroed when the context is
  // made, and a row is cleared by un-writing what was written to it): the number of non-zero bins k_desc_group wrote, or
  // FX_ROW_DIRTY when another tier (or a NaN fill) wrote the row — such a row is cleared whole.
  uint32_t *desc_nbins;   // [max_total_kp]
  uint16_t *desc_bins;    // [max_total_kp][FX_GROUP_CAP]: those bins
  uint32_t *flags;        // [B]
  // work lists for the large-capacity tiers
  uint32_t *huge_rings;   // [B*n_rings]  rings for the workgroup tier (k_rings_large), by XCD class
  uint32_t *huge_rings2;  // [B*n_rings]  rings the second run tier hands to the workgroup tier, by XCD class
  uint32_t *big_merge;    // [B]
  uint32_t *huge_merge;   // [B]  scans with more candidates than the LDS merge tiers hold
  uint32_t *front_n;      // [B]  k_front_ab -> k_front_cd: the scan's ring-major entries; 0: none (empty results written); FX_NONE: handed to k_front_redo
  uint32_t *redo;         // [B]  scans that do not fit k_front's LDS tables: k_front_redo runs the general kernels' bodies on them
  uint32_t *slow;         // [B]  scans with work for the slow tier (k_slow): rings or merges beyond every LDS-sized tier
  uint32_t *slow_state;   // [B]  1 while the scan is listed (k_slow clears it)
  uint32_t *ring_pending; // [B][(n_rings + 31) / 32]  bit r: ring r of the scan waits for k_slow (which clears it)
  uint32_t *gs_pool;      // [gs_slots][gs_words]  k_slow's scratch: the LDS tiers' per-point / per-cluster arrays, in HBM
  uint32_t *gsd_pool;     // [gsd_slots][gsd_words]  dense_slow_loop's scratch: a support set of up to max_points points
  uint32_t *merge_hp;     // [B][merge_hp_words]  the large merge tier as three launches (batches of few scans): a scan's slices' roots, its bin table, its state (null: not allocated — the one launch)
  float4 *merge_sorted;   // [B][max_candidates] (x, y, pseudo z, id) in bin order: k_merge_huge's pair tests (allocated only when that tier exists)
  uint32_t *list_desc;    // [max_total_kp]  rows whose list is too long for one wavefront (257 .. dense_min support points)
  uint32_t *wave_desc;    // [max_total_kp]  rows with 65..256 support points (one wavefront each)
  // dense tier (k_dense_*)
  uint32_t *dense_rows;   // [max_dense_rows]  rows of the tier (FX_NONE: no room in the pools, flagged)
  uint32_t *dense_order;  // [4][max_dense_rows]  slots by size class (largest rows first)
  uint32_t *dense_off;    // [max_dense_rows]  the row's region of dense_pts / dense_q
  uint32_t *dense_koff;   // [max_dense_rows]  the row's region of dense_key (rows whose keys do not fit LDS)
  uint32_t *dense_qoff;   // [max_dense_rows]  the row's region of dense_q
  uint32_t *dense_nq;     // [max_dense_rows]  queries (densities this row computes)
  uint32_t *dense_nm;     // [max_dense_rows]  binned neighbours (FX_NONE: failed row)
  uint32_t *dense_cells;  // [max_dense_rows][25 * 25 * 7]  end of every cell in the row's sorted region
  uint2 *dense_items;     // work items of k_dense_density: (slot, first query)
  float4 *dense_pts;      // [dense_cap]  support sets sorted by cell (x, y, z rotated, point index as bits)
  uint32_t *dense_q;      // [dense_qcap]  query lists (positions in the row's sorted region, cell by cell, padded to four)
  unsigned long long *dense_key;   // [dense_cap]  (bin, d2, index) keys of rows too large for the LDS sort
  unsigned long long *dens_cache;  // [B][max_points]  batch tag << 21 | local point density of the point (k_dense_density)
  unsigned long long *seq;         // [1]  batches processed (device side), the cache's tag
  float4 *ovf_pts;        // [B][ovf_cap]  list entries beyond list_cap, unordered (k_gather)
  uint32_t *ovf_kp;       // [B][ovf_cap]  keypoint ordinal of each
  uint32_t *ovf_cnt;      // [B]
  uint32_t *gather_cnt;   // [B][16][max_keypoints]  k_gather_count -> k_gather_scatter: a slice's entries per keypoint (several workgroups a scan: batches of few big scans)
  // per-keypoint support lists written by k_gather
  float4 *s_pts;          // [max_total_kp][list_cap]  (x, y, z rotated, point index as bits)
  uint32_t *s_cnt;        // [max_total_kp]
  uint2 *row_map;         // [max_total_kp]  (scan, keypoint ordinal) of each descriptor row
  float4 *row_kp;         // [max_total_kp]  the row's keypoint and its 3DSC x-axis (first-pass ordinal), so that the
  float2 *row_xa;         //                 per-keypoint kernels fetch everything a row needs in one round trip
  unsigned long long *clk;     // [FX_CLK_SLOTS][2] k_prep's first start / last end on the device's constant-rate clock, by batch
  unsigned long long *stamps;  // [32] diagnostic build only (-DFX_STAMPS)
  uint32_t *tier_hint;    // [FX_N_HINTS], pinned HOST memory: the batch's counts of work for the rarely used tiers, which the host sizes the next batch's launches of those tiers by (k_offsets, k_desc_mid write them; grid sizes only — never what is computed)
  uint32_t *counters;     // [FX_N_COUNTER_WORDS]: 16.. / 24.. rings handed on per XCD class; 1 big_merge, 4 list_desc, 6 dense rows (2 / 3 / 7 / 10: by size class), 8 wave_desc, 9 huge_merge, 12 key pool used, 13 sorted pool used, 14 density items, 15 / 11 / 5 / 0 tickets of k_dense_density / sort / finish_s / finish_l
};

#endif
