/*
 * fx.h — C-ABI of the MI355X-native per-scan detector/descriptor hot path.
 *
 * The reference (GAVLab/feature_extraction) has no plugin/FFI boundary; its hot
 * path is the body of FeatureExtractionNode::cloudCallback between message
 * conversion and the first publish (ref: src/feature_extraction_node.cpp:83-115).
 * Every entry point below replaces a piece of that body and cites it.
 *
 * Plain C: pointers, sizes, POD structs.  No torch / PCL / ROS types.
 * A context is NOT thread-safe: one context per (host thread, device).
 * All outputs are owned by the context and stay valid until the next
 * fx_process_batch / fx_destroy on that context.
 */
#ifndef FX_H_
#define FX_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FX_VERSION_MAJOR 0
#define FX_VERSION_MINOR 7

/* pcl::ShapeContext1980: 12 azimuth x 11 elevation x 15 radius bins + rf[9]
 * (ref: include/feature_extraction/feature_extraction_node.h:35-53,75). */
#define FX_DESC_BINS 1980
#define FX_DESC_RF 9
#define FX_DESC_FLOATS 1989 /* 7956 B per keypoint == sizeof(pcl::ShapeContext1980) */
/* pcl::PointDescriptor wire record of ~features (ref: node.h:35-42): x@0 y@4 z@8
 * pad@12 intensity@16 descriptor@20 rf@7940, sizeof 7984 (EIGEN_ALIGN16). */
#define FX_FEATURE_RECORD_BYTES 7984

typedef enum fx_status {
  FX_OK = 0,
  FX_ERR_INVALID_ARG = 1,
  FX_ERR_NO_DEVICE = 2, /* no HIP device / extension unusable: never a CPU fallback */
  FX_ERR_HIP = 3,
  FX_ERR_OOM = 4,
  FX_ERR_TOO_LARGE = 5 /* batch or scan exceeds the limits given to fx_create */
} fx_status;

/* Per-scan flag bits (fx_batch_view.flags).  Capacity overflow never truncates
 * silently: the scan's outputs are then incomplete and the bit says which stage. */
#define FX_FLAG_RING_OVERFLOW 0x1u      /* a ring held more points than limits.max_ring_points */
#define FX_FLAG_CAND_OVERFLOW 0x2u      /* more per-ring candidates than limits.max_candidates */
#define FX_FLAG_KP_OVERFLOW 0x4u        /* more keypoints than limits.max_keypoints */
#define FX_FLAG_NBR_OVERFLOW 0x8u       /* the dense descriptor tier's pools are exhausted (limits.max_dense_points; the scan's
                                         * overflow region holds max_points entries): the keypoint's descriptor is NaN */
#define FX_FLAG_TOTAL_KP_OVERFLOW 0x10u /* batch-wide keypoint pool exhausted */
#define FX_FLAG_KPC_OVERFLOW 0x20u      /* keypoint_cloud exceeded its pool */
#define FX_FLAG_INTERNAL 0x40u          /* a self-check of the library failed for this scan (the sliced streaming pass's two
                                           counts of a slice's survivors disagree): its results are not to be trusted — a bug */

/* The 14 ROS private parameters of the reference node (ref: node.cpp:9-34,
 * members node.h:115-127) + the constants the reference hard-codes for the
 * VLP-16 (ref: node.cpp:195, 200, 227), exposed so 64/128-ring sensors work.
 * Doubles/ints exactly as the reference stores them; narrowing to float happens
 * inside, where PCL does it (see oracle/fx_oracle.cpp). */
typedef struct fx_params {
  int32_t cloud_leveling;            /* ref: node.cpp:9   default 1 (host shell concern) */
  double x_min, x_max;               /* ref: node.cpp:14-15  0, 75   */
  double y_min, y_max;               /* ref: node.cpp:16-17  -30, 30 */
  double z_min, z_max;               /* ref: node.cpp:18-19  -1.5, 5 */
  double cluster_tolerance;          /* ref: node.cpp:24  0.65 */
  int32_t cluster_min_count;         /* ref: node.cpp:25  5    */
  int32_t cluster_max_count;         /* ref: node.cpp:26  50   */
  double cluster_radius_threshold;   /* ref: node.cpp:27  0.15 */
  int32_t number_detection_channels; /* ref: node.cpp:28  1    */
  int32_t estimate_descriptors;      /* ref: node.cpp:33  1    */
  double descriptor_radius;          /* ref: node.cpp:34  2.5  */
  /* -- hard-coded in the reference, parameters here -- */
  int32_t n_rings;       /* ref: node.cpp:195  `i<16`                       */
  double el0_deg;        /* ref: node.cpp:200  (i-7)*2-1 at i=0  => -15     */
  double el_step_deg;    /* ref: node.cpp:200  2 deg; window = centre +- step/2 (ref: :201) */
  int32_t secondary_max; /* ref: node.cpp:227  setMaxClusterSize(16)        */
} fx_params;

/* Capacities fixed at fx_create (no allocation in the steady state). 0 = default. */
typedef struct fx_limits {
  uint32_t max_batch;           /* scans per fx_process_batch                       */
  uint32_t max_points;          /* points per scan                                  */
  uint32_t max_ring_points;     /* points per (scan, ring)               (def 2048; <= 32768.  Rings of up to ~2400 points are
                                 * clustered in LDS — azimuth-ordered rings of few runs at any size —, larger ones on scratch
                                 * in HBM: slower, same result) */
  uint32_t max_ring_candidates; /* candidates one ring may emit          (def 256)  */
  uint32_t max_candidates;      /* per-ring candidates per scan, all rings (def 2048; <= 32768.  <= ~3800 live in LDS as points,
                                 * beyond that — up to ~16000 — the large merge tier keeps the coordinates in HBM, beyond
                                 * that everything: slower, same result) */
  uint32_t max_keypoints;       /* keypoints per scan                    (def 256)  */
  uint32_t max_neighbors;       /* support-list slots per keypoint row (def 1024, 4096 for scans of more than 65536
                                 * points; 16 B each, capped at 4096).  Not a cap on the support set: larger sets overflow
                                 * into a per-scan region and take the dense tier */
  uint32_t max_total_keypoints; /* keypoints per batch (descriptor pool) (def max_batch*64) */
  uint32_t max_kpc_points;      /* keypoint_cloud points per scan        (def 4096) */
  uint32_t max_dense_points;    /* support points per batch the dense descriptor tier sorts (rows of more than 1024
                                 * support points; def max(max_batch, 32)*max_points; 28 B each + 17.5 KB per 1024).  New in 0.4 */
  uint32_t max_overflow_points; /* support-list entries beyond max_neighbors a scan's rows may have together (its overflow
                                 * region: 20 B each, per scan of the batch; def max_points, 32/max_batch times that in
                                 * contexts of fewer than 32 scans).  New in 0.6 */
} fx_limits;

/* One scan = what cloudCallback receives after fromPCLPointCloud2
 * (ref: node.cpp:77-81): N points with x,y,z as float at byte offsets 0,4,8 of
 * each record.  stride_bytes = 16 for packed float4 (x,y,z,intensity), 32 for
 * PCL's in-memory pcl::PointXYZI.  Incoming intensity is ignored: the
 * reference overwrites it with the elevation angle (ref: node.cpp:154).
 * roll/pitch are the node's members (ref: node.h:116, set at node.cpp:57-70). */
typedef struct fx_scan_desc {
  const void *points; /* host or device pointer (see FX_IN_DEVICE), 16-byte aligned */
  uint32_t n_points;
  uint32_t stride_bytes; /* multiple of 16 in [16, 256] */
  double roll, pitch;    /* radians; narrowed to float like Eigen::AngleAxisf (ref: node.cpp:163-164) */
} fx_scan_desc;

#define FX_IN_DEVICE 0x1u   /* fx_scan_desc.points are device pointers */
#define FX_OUT_HOST 0x2u    /* copy results to the context's pinned host mirrors and synchronise */
#define FX_OUT_DEBUG 0x4u   /* with FX_OUT_HOST: also copy candidates / membership arrays */
#define FX_OUT_CLOUDS 0x8u  /* with FX_OUT_HOST: also copy the filtered cloud and keypoint_cloud */

/* View of one batch's results.  `d_` = device pointers (always set),
 * `h_` = pinned-host mirrors (set when FX_OUT_HOST, else NULL).
 * Keypoint k of scan b:  keypoints[b*max_keypoints + k]  (x, y, z, elevation_deg)
 *                         = the `~keypoints` cloud (ref: node.cpp:129-131, 238-257).
 * Its descriptor:        descriptors[(kp_offset[b] + k) * FX_DESC_FLOATS ...]
 *                         = pcl::ShapeContext1980 {descriptor[1980], rf[9]} (ref: node.cpp:353).
 * filtered[b*max_points + i], i < n_filtered[b] = the `~cloud` topic (ref: node.cpp:137-139).
 * kpc[b*max_kpc_points + i], i < n_kpc[b] = the `~keypoint_cloud` topic (ref: node.cpp:133-135, 206, 323). */
typedef struct fx_batch_view {
  uint32_t batch;
  uint32_t max_points, max_keypoints, max_candidates, max_kpc_points;
  uint32_t total_keypoints; /* valid only after FX_OUT_HOST */
  /* device */
  const uint32_t *d_n_keypoints; /* [B] */
  const uint32_t *d_kp_offset;   /* [B+1] exclusive prefix of n_keypoints */
  const float *d_keypoints;      /* [B][max_keypoints][4] */
  const float *d_descriptors;    /* [max_total_keypoints][1989]; rows [0, total_keypoints) are this batch's.  READ-ONLY: a row
                                  * keeps its content until a later batch uses it and is then cleared by un-writing what was
                                  * written to it — writing into this buffer corrupts later batches */
  const uint32_t *d_flags;       /* [B] FX_FLAG_* */
  const uint32_t *d_n_filtered;  /* [B] */
  const float *d_filtered;       /* [B][max_points][4] */
  const uint32_t *d_n_kpc;       /* [B] */
  const float *d_kpc;            /* [B][max_kpc_points][4] */
  /* host mirrors */
  const uint32_t *h_n_keypoints;
  const uint32_t *h_kp_offset;
  const float *h_keypoints;
  const float *h_descriptors;
  const uint32_t *h_flags;
  const uint32_t *h_n_filtered;
  const float *h_filtered;
  const uint32_t *h_n_kpc;
  const float *h_kpc;
  /* debug / membership (host mirrors only with FX_OUT_DEBUG) */
  const uint32_t *h_n_candidates;  /* [B]  size of keypoints_full (ref: node.cpp:205) */
  const float *h_candidates;       /* [B][max_candidates][4]  per-ring centroids, ring order */
  const uint32_t *h_cand_size;     /* [B][max_candidates]  points in the per-ring cluster */
  const int32_t *h_cand_keypoint;  /* [B][max_candidates]  keypoint ordinal the candidate merged into, -1 if none */
  const uint32_t *h_kpc_cand;      /* [B][max_kpc_points]  candidate ordinal of each keypoint_cloud point */
  const uint32_t *h_kp_size;       /* [B][max_keypoints]   candidates merged into the keypoint */
  const uint32_t *h_kp_neighbors;  /* [B][max_keypoints]   3DSC neighbours within descriptor_radius */
} fx_batch_view;

/* Per-stage device time of the last batch (HIP events on the context's stream). */
#define FX_N_STAGES 9
typedef struct fx_timings {
  /* k_prep, k_bucket, k_rings_runs (one wavefront per ring), k_rings_large (the second run tier of many-ring sensors + the workgroup ring tier), k_merge (merge tiers
   * small / big / large, k_offsets), k_gather (+ k_rng_ord on small batches), k_desc_group, k_desc_mid (wave rows and list
   * rows in one launch; + the longest lists), k_desc_rare (re-gather, whole-CU and slab tiers) */
  float ms[FX_N_STAGES];
  float total_ms;
  /* k_prep's execution span on the device's constant-rate clock: first workgroup's start to last workgroup's end — what
   * rocprofv3 reports as the kernel's duration.  The HIP-event span ms[0] also counts the launch's wait for free CUs,
   * which matters when several contexts keep the GPU busy.  0 when the batch is too old for its clock slot. */
  float k_prep_exec_ms;
} fx_timings;

typedef struct fx_ctx fx_ctx;

uint32_t fx_version(void);
/* ABI guard.  The structs of this header may grow at their END between minor versions (fx_limits did in 0.4 and 0.6) and the
 * library reads every member: a caller compiled against another header must not get as far as fx_create.  Pass the version and
 * the sizes the caller was compiled with (FX_CHECK_ABI() does); FX_ERR_INVALID_ARG, with the mismatch in fx_last_error(),
 * when they are not the library's.  Always initialise fx_params / fx_limits with fx_params_default / fx_params_launch /
 * fx_limits_default of the same library before overriding members. */
fx_status fx_check_abi(uint32_t header_version, size_t sizeof_params, size_t sizeof_limits, size_t sizeof_scan_desc,
                       size_t sizeof_batch_view);
#define FX_CHECK_ABI() \
  fx_check_abi(((uint32_t)FX_VERSION_MAJOR << 16) | FX_VERSION_MINOR, sizeof(fx_params), sizeof(fx_limits), sizeof(fx_scan_desc), sizeof(fx_batch_view))
const char *fx_status_str(fx_status s);
/* message of the last failing call on this thread (HIP error string etc.) */
const char *fx_last_error(void);

/* ref: node.cpp:9-34 (defaults) */
void fx_params_default(fx_params *p);
/* ref: launch/keypoint_playback.launch:17-33 (the preset the launch file sets) */
void fx_params_launch(fx_params *p);
void fx_limits_default(fx_limits *l, uint32_t max_batch, uint32_t max_points);
/* The same for sensors whose keypoints rarely have more than max_neighbors (1024) support points — VLP-16-class scans at the
 * reference's descriptor radius: the dense descriptor tier's pools and the overflow regions hold a few such rows per batch
 * instead of every point of it (1024 scans of 28 800 points: 4.5 GB of device memory instead of 7).  Results are the same;
 * a batch with more dense rows than the pools hold flags them (FX_FLAG_NBR_OVERFLOW), as with any limit.  New in 0.6 */
void fx_limits_sparse(fx_limits *l, uint32_t max_batch, uint32_t max_points);

/* Replaces the FeatureExtractionNode constructor's parameter block (ref: node.cpp:3-34).
 * Allocates every device/host buffer; fails with FX_ERR_NO_DEVICE when no GPU. */
fx_status fx_create(const fx_params *params, const fx_limits *limits, int device_id, fx_ctx **out);
void fx_destroy(fx_ctx *ctx);
/* hipStream_t to launch on (NULL = the context's own stream). */
fx_status fx_set_stream(fx_ctx *ctx, void *hip_stream);
/* The hipStream_t the context launches on (its own unless fx_set_stream gave it another): for event waits and for wrapping
 * it in the caller's framework (torch.cuda.ExternalStream). */
fx_status fx_get_stream(fx_ctx *ctx, void **hip_stream);
/* Streaming mode: batches of up to max_batch scans are replayed as one HIP graph per
 * batch size instead of ~30 separate launches (0 = never).  Needs a non-NULL stream; ignored while
 * profiling is on.  Results are identical either way. */
fx_status fx_set_graph_batch(fx_ctx *ctx, uint32_t max_batch);
/* How many contexts the caller keeps busy on this device at a time (default 1; bench.py and fx::MultiGpu run four).  A
 * launch-policy hint only — results never depend on it: with several batches in flight the grid-stride kernels take
 * smaller grids, so that the batches share the chip instead of each claiming all of it (k_desc_group: one workgroup a CU
 * instead of ten is +2 % on four batches in flight and -10 % on a batch alone).  New in 0.6 */
fx_status fx_set_batches_in_flight(fx_ctx *ctx, uint32_t n);
/* depth > 0: record HIP events around every stage kernel for the next batches, keeping the
 * last `depth` batches; 0 disables.  fx_get_timings reads the batch `back` calls ago
 * (0 = most recent) and waits for it to finish. */
fx_status fx_set_profiling(fx_ctx *ctx, int depth);
/* Restrict the events to some stages (bit i = stage i of fx_timings; default all): every event costs a
 * few microseconds of stream time, so a throughput measurement times only the kernel it needs.  The
 * first and last event of a batch are always recorded (total_ms); untimed stages read 0. */
fx_status fx_set_profiling_stages(fx_ctx *ctx, uint32_t stage_mask);
fx_status fx_get_timings(fx_ctx *ctx, uint32_t back, fx_timings *t);
/* Algorithmic bytes every stage of the LAST batch had to move — what it must read of its inputs plus what it must
 * write of its outputs, each once, from the batch's own counts (points, survivors, ring members, candidates, support
 * points, keypoints); no padding, no re-reads.  The figure a per-kernel roofline fraction divides by the kernel's
 * duration.  Waits for the batch.  (SURVEY.md 8d's per-scan B_alg is the whole path's: bytes[0]'s read + the keypoint
 * and descriptor writes.) */
typedef struct fx_stage_bytes {
  double read[FX_N_STAGES], written[FX_N_STAGES];
} fx_stage_bytes;
fx_status fx_get_stage_bytes(fx_ctx *ctx, fx_stage_bytes *out);
fx_status fx_get_limits(const fx_ctx *ctx, fx_limits *l);

/* Replaces cloudCallback's body for a batch of B scans (ref: node.cpp:83-115):
 * getElevationAngles (:147-156) -> rotateCloud (:159-167) -> filterCloud (:169-183)
 * -> estimateKeypoints (:185-259, getCylinderSegments :261-327)
 * -> estimateDescriptors (:329-355).  Empty scans give K = 0 (ref: :209-210, :263-264). */
fx_status fx_process_batch(fx_ctx *ctx, const fx_scan_desc *scans, uint32_t batch, uint32_t flags,
                           fx_batch_view *out);
/* Wait for the context's stream. */
fx_status fx_synchronize(fx_ctx *ctx);

/* pcl::concatenateFields(keypoints, descriptors) (ref: node.cpp:119): packs the last
 * batch's keypoints + descriptors into 7984-byte pcl::PointDescriptor records on
 * the device.  dst_device must hold total_keypoints records. */
fx_status fx_pack_features(fx_ctx *ctx, void *dst_device, uint32_t capacity_records);

/* ---- PointCloud2 wire formats ----
 * Ingress: what pcl_conversions::toPCL + pcl::fromPCLPointCloud2 do for the node (ref: node.cpp:79-81):
 * pick the float32 fields x, y, z (and optionally intensity) by their byte offsets out of
 * point_step-byte records (extra fields such as `ring` are skipped; point_step may be any size) into
 * the packed float4 layout fx_scan_desc takes.  data_device and dst_device_xyzi are device pointers;
 * dst holds n_points * 16 bytes.  The offsets come from the message's PointField list. */
typedef struct fx_pc2_layout {
  uint32_t point_step;
  uint32_t offset_x, offset_y, offset_z;
  uint32_t offset_intensity; /* 0xffffffff: no such field (the path ignores incoming intensity anyway) */
  uint32_t is_bigendian;
} fx_pc2_layout;
fx_status fx_unpack_pointcloud2(fx_ctx *ctx, const void *data_device, uint32_t n_points, const fx_pc2_layout *layout,
                                void *dst_device_xyzi);
/* Egress: one scan's cloud of the last batch as pcl_ros serialises a PointCloud<PointXYZI>
 * (ref: node.cpp:129-139): 32-byte records, x@0 y@4 z@8 intensity@16 (fields x, y, z, intensity;
 * point_step 32).  dst_device holds capacity_points * 32 bytes; *n_points_out = points written. */
#define FX_CLOUD_KEYPOINTS 0u      /* ~keypoints      */
#define FX_CLOUD_FILTERED 1u       /* ~cloud          */
#define FX_CLOUD_KEYPOINT_CLOUD 2u /* ~keypoint_cloud */
fx_status fx_pack_pointxyzi(fx_ctx *ctx, uint32_t which, uint32_t scan, void *dst_device, uint32_t capacity_points,
                            uint32_t *n_points_out);

/* Fixed-stride keypoint records of the last batch for the cross-GPU gather (one RCCL
 * collective per batch): per scan (1 + rec_keypoints) float4 = {n_kp, flags, 0, 0 as u32}
 * followed by rec_keypoints (x, y, z, elevation) entries, zero padded.  dst_device must hold
 * batch * (1 + rec_keypoints) * 16 bytes. */
fx_status fx_pack_keypoint_records(fx_ctx *ctx, void *dst_device, uint32_t rec_keypoints);

/* The same keypoints as ONE compact block for the cross-GPU gather (0.7): a batch's keypoints packed
 * in scan order behind their offsets, instead of max-stride records (54 keypoints a VLP-16 scan
 * against a stride of 256: a quarter of the bytes on xGMI and in every receiver's HBM).  The block is
 * fx_keypoint_block_bytes(max_scans, max_total_keypoints) bytes whatever the batch holds — every rank
 * hands the collective the same count — as rows of 16 bytes:
 *   row 0                       {scans, keypoints stored, OR of all flags, max_total_keypoints} (u32)
 *   then ceil((max_scans+1)/4)  kp_offset[max_scans + 1] (u32): scan b's keypoints are rows
 *                               [kp_offset[b], kp_offset[b+1]) of the keypoint area; entries beyond
 *                               the batch repeat the total
 *   then ceil(max_scans/4)      flags[max_scans] (u32; 0 beyond the batch)
 *   then max_total_keypoints    (x, y, z, elevation) rows, zero beyond the total
 * A batch with more keypoints than max_total_keypoints is cut there; the scans that lose keypoints
 * (and row 0) carry FX_FLAG_KP_OVERFLOW.  Replaces what the reference would publish per scan on
 * ~keypoints (ref: node.cpp:133-135) for the scans of a whole batch.  Enqueued on the context's
 * stream behind the last fx_process_batch. */
size_t fx_keypoint_block_bytes(uint32_t max_scans, uint32_t max_total_keypoints);
fx_status fx_pack_keypoint_block(fx_ctx *ctx, void *dst_device, uint32_t max_scans, uint32_t max_total_keypoints);

/* Rotation matrix of rotateCloud (ref: node.cpp:161-164): R = Ry(pitch)*Rx(roll)
 * through Eigen's AngleAxisf -> Quaternionf -> toRotationMatrix, all float. Host only. */
void fx_rotation_from_roll_pitch(double roll, double pitch, float R[9]);
/* ShapeContext3DEstimation::initCompute tables (radii[16], theta[12], phi[13], lut[1980])
 * for descriptor_radius R: rmin = R/10 (ref: node.cpp:350-352). Host only. */
void fx_sc3d_tables(double R, float *radii16, float *theta12, float *phi13, float *lut1980);
/* x-axis of keypoint ordinal k from the boost::mt19937(12345) stream of 3DSC. Host only. */
void fx_sc3d_xaxis(uint32_t k, float xy[2]);

/* Synthetic scan generator (the reference ships no data).
 * Writes n_rings*n_az float4 (x,y,z,0) in firing order (azimuth-major, ring-minor). Host only. */
typedef struct fx_synth_cfg {
  uint32_t n_rings, n_az;
  double el0_deg, el_step_deg;
  uint32_t n_poles;
  double pole_radius, pole_height;
  double x_lo, x_hi, y_lo, y_hi; /* pole centres uniform in this box */
  double sensor_height;          /* ground plane z = -sensor_height */
  double wall_radius;            /* enclosing cylinder, always hit => fixed N */
  uint64_t seed;
} fx_synth_cfg;
void fx_synth_cfg_vlp16(fx_synth_cfg *c, uint64_t seed);
uint32_t fx_synth_scan(const fx_synth_cfg *c, float *xyzi_out, uint32_t capacity_points);

/* ---- test entry points: only in the TEST build of the library (lib/libfx_hip_test.so, compiled with -DFX_TEST_HOOKS,
 * which also reads the environment hooks tests use to push work through the rarely used tiers).  The product library
 * exports none of them and contains none of the k_test_* kernels. ---- */
#ifdef FX_TEST_HOOKS
/* Test hook: host build of the cluster-order replay the kernels run on one GPU lane
 * (csrc/fx_sort_replay.h).  perm_out[s] = ordinal of the cluster PCL returns at position s. */
void fx_test_sort_replay(const uint32_t *sizes, uint32_t n, uint32_t *perm_out);
/* same, with the final insertion phase replaced by the stable ranking the kernels use */
void fx_test_sort_replay_ranked(const uint32_t *sizes, uint32_t n, uint32_t *perm_out);
/* same, with the partition step stated through position lists (the rule the wavefront version follows) */
void fx_test_sort_replay_lists(const uint32_t *sizes, uint32_t n, uint32_t *perm_out);
/* Test hook: the replay as the kernels run it (wavefront partition phase + ranking) on device `device`,
 * one workgroup per sequence; n <= 192 per sequence.  sizes / perm_out: host arrays [n_seq][n]. */
fx_status fx_test_sort_replay_device(int device, const uint32_t *sizes, uint32_t n_seq, uint32_t n, uint32_t *perm_out);
/* Test hook: the two evaluations of a point's elevation angle the filter stage has (ref: node.cpp:147-156) on host
 * points xyz[n][3]: the table-driven one (fast_out, and whether it vouches for its value: fast_ok_out) and the one
 * through the library's fp64 atan2 that takes the points it does not vouch for (exact_out). */
fx_status fx_test_elevation_device(int device, const float *xyz, uint32_t n, float *fast_out, uint8_t *fast_ok_out, float *exact_out);
/* Test hook: 3DSC's local point density count as the descriptor kernels evaluate it (two points per packed instruction, the
 * comparison folded into a clamped fma) and as the plain `dist2 < r2` it replaces: for each of nq queries (xyzw records) the
 * number of the n support points (xyzw records) closer than sqrt(r2), both ways. */
fx_status fx_test_within_device(int device, const float *support_xyzw, uint32_t n, const float *query_xyzw, uint32_t nq, float r2,
                                uint32_t *packed_out, uint32_t *plain_out);
#endif /* FX_TEST_HOOKS */

#ifdef __cplusplus
}
#endif
#endif /* FX_H_ */
