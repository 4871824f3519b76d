/*
 * fx_oracle.h — C interface of the CPU oracle (TEST INFRASTRUCTURE, not product).
 *
 * The oracle is a plain C++17 restatement of the reference's per-scan pipeline
 * (ref: src/feature_extraction_node.cpp:147-355) plus the PCL 1.8 / FLANN /
 * Eigen / Boost semantics those lines invoke.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or data, and PCL
 * cannot be built in this image, so this restatement is checked only against the
 * known-answer constants of SURVEY.md Appendix B and an independent scipy
 * cross-check of cluster membership (tests/test_oracle_*.py).
 */
#ifndef FX_ORACLE_H_
#define FX_ORACLE_H_
#include <stdint.h>
#include "../include/fx.h" /* fx_params only (interface definition, no product code) */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fxo_result fxo_result;

#define FXO_SEARCH_BRUTE 0  /* O(N^2) literal statement of the radius predicate */
#define FXO_SEARCH_KDTREE 1 /* own exact kd-tree (leaf 15): the timed CPU baseline */
#define FXO_TRIG_F64_ROUNDED 0 /* phi/theta via fp64 atan2/acos rounded once to fp32 */
#define FXO_TRIG_LIBM_F32 1    /* literal atan2f/acosf of this glibc (diagnostic mode) */
/* Switchable policies, OR-ed into trig_kind: where the restatement had to pick ONE reading of an un-vendored dependency
 * (PARITY UNPINNED: none of PCL / Eigen / Boost is in this image).  Whoever runs tools/pcl_crosscheck against a real PCL
 * flips the switch that matches the installed versions; tests/test_oracle_policies.py counts what each one moves. */
#define FXO_POLICY_SKIP_EPSILON 0x10     /* 3DSC skips a neighbour when d2 < FLT_EPSILON (SURVEY.md A.8-6's reading) instead of
                                          * pcl::utils::equal's default tolerance numeric_limits<float>::min() */
#define FXO_POLICY_EIGEN32_NORMALIZE 0x20 /* Eigen 3.2: normalize() divides by the norm unguarded (a zero vector becomes NaN);
                                          * Eigen 3.3 leaves a zero vector alone */
#define FXO_POLICY_STD_UNIFORM_FLOAT 0x40 /* PCL >= 1.10: std::mt19937 + std::uniform_real_distribution<float>(0, 1) for the
                                          * x-axis draws instead of boost::mt19937 + boost::uniform_01 (double, narrowed) */

/* Runs cloudCallback's body (ref: node.cpp:83-115) on one scan.
 * points: n records of stride_floats floats, x,y,z at 0,1,2. */
fxo_result *fxo_run(const fx_params *p, const float *points, uint32_t n, uint32_t stride_floats,
                    double roll, double pitch, int search_kind, int trig_kind);
void fxo_free(fxo_result *r);

uint32_t fxo_n_filtered(const fxo_result *r);
uint32_t fxo_n_candidates(const fxo_result *r);
uint32_t fxo_n_keypoints(const fxo_result *r);
uint32_t fxo_n_kpc(const fxo_result *r);
/* each copies into caller storage sized from the counts above */
void fxo_rotated(const fxo_result *r, float *xyzi);        /* [n][4] cloud_full after rotateCloud (intensity = elevation) */
void fxo_filtered(const fxo_result *r, float *xyzi);       /* [n_filtered][4] */
void fxo_candidates(const fxo_result *r, float *xyzi, uint32_t *size, int32_t *keypoint); /* keypoints_full */
void fxo_kpc(const fxo_result *r, float *xyzi, uint32_t *cand);
void fxo_keypoints(const fxo_result *r, float *xyzi, uint32_t *size, uint32_t *neighbors);
void fxo_descriptors(const fxo_result *r, float *desc1989); /* [n_keypoints][1989] */
/* per-ring membership: label[i] for each filtered point i and ring r: -1 = not in ring,
 * else the smallest filtered-cloud index of its connected component.  [n_rings][n_filtered] */
void fxo_ring_labels(const fxo_result *r, int32_t *labels);

/* stand-alone pieces for known-answer tests */
void fxo_rotation(double roll, double pitch, float R[9]);
void fxo_sc3d_tables(double R, float *radii16, float *theta12, float *phi13, float *lut1980);
void fxo_sc3d_rng(uint32_t n_draws, uint32_t *u32_out, float *f32_out);
float fxo_radius2(double r);              /* (float)(r*r), r as the caller's double (A.4) */
float fxo_cluster_radius2(double tol);    /* EuclideanClusterExtraction narrows tol to float first */
void fxo_sort_by_size_desc(const uint32_t *sizes, uint32_t n, uint32_t *perm_out); /* libstdc++ std::sort(rbegin,rend) */
float fxo_elevation_deg(float x, float y, float z); /* ref: node.cpp:150-154 */
void fxo_antiqsort(uint32_t n, uint32_t *sizes_out); /* adversarial size sequence for std::sort(rbegin, rend) */

/* CPU baseline: whole pipeline (kd-tree) on cycled scans with `threads` host threads for ~`seconds`; returns elapsed s */
double fxo_bench_throughput(const fx_params *p, const float *points, uint32_t n_scans, uint32_t n, uint32_t stride_floats,
                            double roll, double pitch, uint32_t threads, double seconds, uint64_t *scans_done,
                            uint64_t *keypoints);

#ifdef __cplusplus
}
#endif
#endif
