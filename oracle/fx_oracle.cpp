/*
 * fx_oracle.cpp — CPU oracle: TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * A dependency-free C++17 restatement of the reference's per-scan pipeline,
 *   ref: src/feature_extraction_node.cpp:147-355
 * and of the PCL 1.8.x / FLANN / Eigen 3.3 / Boost.Random / libstdc++ behaviour
 * those lines call into (none of these libraries exist in
 * the build image, so their semantics are restated from the published sources).
 *
 * PARITY UNPINNED: the reference has no tests / golden data and PCL cannot be
 * built here.  What pins this file: SURVEY.md Appendix B known answers and a scipy
 * connected-components cross-check (tests/test_oracle_*.py).
 *
 * Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math (see Makefile).
 * Float arithmetic is written out operation by operation in the order PCL/FLANN/
 * Eigen evaluate it; do not "simplify" expressions in this file.
 */
#include "fx_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <random>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

namespace {

struct P4 {
  float x, y, z, i; /* pcl::PointXYZI payload (intensity carries the elevation, ref: node.cpp:154) */
};
typedef std::vector<P4> Cloud;

/* ---------------------------------------------------------------- A.4 */
/* FLANN L2_Simple<float>: result = 0; for each dim: diff = a-b; result += diff*diff */
static inline float dist2(const P4 &q, const P4 &p) {
  float result = 0.0f, diff;
  diff = q.x - p.x;
  result += diff * diff;
  diff = q.y - p.y;
  result += diff * diff;
  diff = q.z - p.z;
  result += diff * diff;
  return result;
}
/* KdTreeFLANN::radiusSearch: static_cast<float>(radius * radius), radius a double */
static inline float radius2(double r) { return static_cast<float>(r * r); }

static inline bool finite3(const P4 &p) {
  return std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z);
}

struct DistIndex { /* flann::DistanceIndex: ordered by (dist, index) */
  float d;
  int idx;
  bool operator<(const DistIndex &o) const { return (d < o.d) || ((d == o.d) && idx < o.idx); }
};

/* Radius search over a fixed cloud: membership d2 < r2 strict (RadiusResultSet::addPoint). */
class Searcher {
 public:
  virtual ~Searcher() {}
  virtual void radius(const P4 &q, float r2, std::vector<DistIndex> &out) const = 0;
};

class BruteSearcher : public Searcher {
 public:
  explicit BruteSearcher(const Cloud &c) : c_(c) {}
  void radius(const P4 &q, float r2, std::vector<DistIndex> &out) const override {
    out.clear();
    for (size_t i = 0; i < c_.size(); ++i) {
      if (!finite3(c_[i])) continue; /* non-finite points are not indexed (is_dense=false path) */
      float d = dist2(q, c_[i]);
      if (d < r2) out.push_back({d, (int)i});
    }
  }

 private:
  const Cloud &c_;
};

/* Own exact kd-tree, leaf size 15 like FLANN KDTreeSingleIndex.  Pruning bounds are
 * evaluated in double with a safety margin, so the result SET is identical to the
 * brute-force predicate; leaves apply the fp32 test above. */
class KdSearcher : public Searcher {
 public:
  explicit KdSearcher(const Cloud &c) : c_(c) {
    idx_.reserve(c.size());
    for (size_t i = 0; i < c.size(); ++i)
      if (finite3(c[i])) idx_.push_back((int)i);
    if (!idx_.empty()) {
      nodes_.reserve(idx_.size() / 4 + 8);
      build(0, (int)idx_.size());
    }
  }
  void radius(const P4 &q, float r2, std::vector<DistIndex> &out) const override {
    out.clear();
    if (nodes_.empty()) return;
    const double lim = (double)r2 * 1.00001 + 1e-30;
    search(0, q, r2, lim, out);
  }

 private:
  struct Node {
    int lo, hi;     /* leaf: range in idx_ */
    int left, right; /* children (-1 for leaf) */
    int dim;
    float split_lo, split_hi; /* max of left / min of right along dim */
  };
  static float coord(const P4 &p, int d) { return d == 0 ? p.x : (d == 1 ? p.y : p.z); }
  int build(int lo, int hi) {
    int me = (int)nodes_.size();
    nodes_.push_back(Node{lo, hi, -1, -1, 0, 0.f, 0.f});
    if (hi - lo <= 15) return me;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = lo; i < hi; ++i)
      for (int d = 0; d < 3; ++d) {
        float v = coord(c_[idx_[i]], d);
        mn[d] = std::min(mn[d], v);
        mx[d] = std::max(mx[d], v);
      }
    int dim = 0;
    for (int d = 1; d < 3; ++d)
      if (mx[d] - mn[d] > mx[dim] - mn[dim]) dim = d;
    if (!(mx[dim] > mn[dim])) return me; /* all coincident: keep as one leaf */
    int mid = (lo + hi) / 2;
    std::nth_element(idx_.begin() + lo, idx_.begin() + mid, idx_.begin() + hi,
                     [&](int a, int b) { return coord(c_[a], dim) < coord(c_[b], dim); });
    float slo = -FLT_MAX, shi = FLT_MAX;
    for (int i = lo; i < mid; ++i) slo = std::max(slo, coord(c_[idx_[i]], dim));
    for (int i = mid; i < hi; ++i) shi = std::min(shi, coord(c_[idx_[i]], dim));
    int l = build(lo, mid);
    int r = build(mid, hi);
    nodes_[me].left = l;
    nodes_[me].right = r;
    nodes_[me].dim = dim;
    nodes_[me].split_lo = slo;
    nodes_[me].split_hi = shi;
    return me;
  }
  void search(int n, const P4 &q, float r2, double lim, std::vector<DistIndex> &out) const {
    const Node &nd = nodes_[n];
    if (nd.left < 0) {
      for (int i = nd.lo; i < nd.hi; ++i) {
        float d = dist2(q, c_[idx_[i]]);
        if (d < r2) out.push_back({d, idx_[i]});
      }
      return;
    }
    double v = coord(q, nd.dim);
    double dl = v - (double)nd.split_lo; /* >0: q is right of every left point */
    double dr = (double)nd.split_hi - v; /* >0: q is left of every right point */
    if (!(dl > 0 && dl * dl > lim)) search(nd.left, q, r2, lim, out);
    if (!(dr > 0 && dr * dr > lim)) search(nd.right, q, r2, lim, out);
  }
  const Cloud &c_;
  std::vector<int> idx_;
  std::vector<Node> nodes_;
};

static std::unique_ptr<Searcher> make_searcher(const Cloud &c, int kind) {
  if (kind == FXO_SEARCH_KDTREE) return std::unique_ptr<Searcher>(new KdSearcher(c));
  return std::unique_ptr<Searcher>(new BruteSearcher(c));
}

/* ---------------------------------------------------------------- A.2 */
/* Eigen: AngleAxisf(pitch,Y) * AngleAxisf(roll,X) -> Quaternionf product -> toRotationMatrix
 * (ref: node.cpp:161-164). */
static void rotation_matrix(double roll_d, double pitch_d, float R[9]) {
  const float pitch = (float)pitch_d, roll = (float)roll_d; /* AngleAxisf narrows */
  /* Quaternion = AngleAxis: ha = 0.5f*angle; w = cos(ha); vec = sin(ha)*axis */
  const float hy = 0.5f * pitch, hx = 0.5f * roll;
  const float aw = std::cos(hy), ax = std::sin(hy) * 0.0f, ay = std::sin(hy) * 1.0f, az = std::sin(hy) * 0.0f;
  const float bw = std::cos(hx), bx = std::sin(hx) * 1.0f, by = std::sin(hx) * 0.0f, bz = std::sin(hx) * 0.0f;
  /* quaternion product a*b (generic formula; with the zero components every term but
   * one vanishes exactly, so association is irrelevant) */
  const float w = aw * bw - ax * bx - ay * by - az * bz;
  const float x = aw * bx + ax * bw + ay * bz - az * by;
  const float y = aw * by + ay * bw + az * bx - ax * bz;
  const float z = aw * bz + az * bw + ax * by - ay * bx;
  const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0f - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = 1.0f - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = 1.0f - (txx + tyy);
}

/* ref: node.cpp:147-156 getElevationAngles */
static inline float elevation_deg(float xf, float yf, float zf) {
  double x = xf, y = yf, z = zf, xp, az, el_deg;
  az = atan2(y, x);
  xp = cos(az) * x + sin(az) * y;
  el_deg = atan2(z, xp) * 180 / M_PI;
  return (float)el_deg;
}

/* pcl::PassThrough on one float field with float limits, inclusive, stable (A.3). */
enum Field { FX_, FY_, FZ_, FI_ };
static inline float field(const P4 &p, Field f) {
  switch (f) {
    case FX_: return p.x;
    case FY_: return p.y;
    case FZ_: return p.z;
    default: return p.i;
  }
}
static void passthrough(const Cloud &in, Field f, double lo_d, double hi_d, Cloud &out,
                        std::vector<int> *kept = nullptr) {
  const float lo = (float)lo_d, hi = (float)hi_d; /* setFilterLimits(const float&, const float&) */
  Cloud tmp;
  tmp.reserve(in.size());
  if (kept) kept->clear();
  for (size_t i = 0; i < in.size(); ++i) {
    const P4 &p = in[i];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    float v = field(p, f);
    if (!std::isfinite(v)) continue;
    if (v < lo || v > hi) continue;
    tmp.push_back(p);
    if (kept) kept->push_back((int)i);
  }
  out.swap(tmp);
}

/* ---------------------------------------------------------------- A.5 / A.6 */
struct Cluster {
  std::vector<int> indices; /* ascending, unique */
};
static bool compareClusters(const Cluster &a, const Cluster &b) { return a.indices.size() < b.indices.size(); }

/* pcl::EuclideanClusterExtraction::extract over an unorganised cloud.
 * all_labels (optional): min index of every point's component, accepted or not. */
static void euclidean_clusters(const Cloud &cloud, double tolerance_d, int min_size, int max_size, int search_kind,
                               std::vector<Cluster> &clusters, std::vector<int> *all_labels = nullptr) {
  clusters.clear();
  if (all_labels) all_labels->assign(cloud.size(), -1);
  if (cloud.empty()) return;
  const float tolerance = static_cast<float>(tolerance_d); /* extract(): static_cast<float>(cluster_tolerance_) */
  const float r2 = radius2((double)tolerance);             /* radiusSearch(point, double radius) */
  const unsigned min_pts = (unsigned)min_size, max_pts = (unsigned)max_size;
  std::unique_ptr<Searcher> tree = make_searcher(cloud, search_kind);
  std::vector<bool> processed(cloud.size(), false);
  std::vector<DistIndex> nn;
  for (int i = 0; i < (int)cloud.size(); ++i) {
    if (processed[i]) continue;
    std::vector<int> seed_queue;
    int sq_idx = 0;
    seed_queue.push_back(i);
    processed[i] = true;
    while (sq_idx < (int)seed_queue.size()) {
      tree->radius(cloud[seed_queue[sq_idx]], r2, nn);
      for (size_t j = 0; j < nn.size(); ++j) { /* unsorted tree: nn_start_idx = 0 */
        if (processed[nn[j].idx]) continue;
        seed_queue.push_back(nn[j].idx);
        processed[nn[j].idx] = true;
      }
      sq_idx++;
    }
    if (all_labels)
      for (int m : seed_queue) (*all_labels)[m] = i; /* i is the component's smallest index */
    if (seed_queue.size() >= min_pts && seed_queue.size() <= max_pts) {
      Cluster r;
      r.indices = seed_queue;
      std::sort(r.indices.begin(), r.indices.end());
      r.indices.erase(std::unique(r.indices.begin(), r.indices.end()), r.indices.end());
      clusters.push_back(r);
    }
  }
  /* "Sort the clusters based on their size (largest one first)" — libstdc++ introsort,
   * unstable for > 16 clusters; this call IS the specification of the tie order (A.6). */
  std::sort(clusters.rbegin(), clusters.rend(), compareClusters);
}

/* ---------------------------------------------------------------- A.8 */
struct Sc3dTables {
  float radii[16], theta[12], phi[13], lut[FX_DESC_BINS];
};
static inline float deg2rad_f(float a) { return a * 0.017453293f; }
static inline float rad2deg_f(float a) { return a * 57.29578f; }

/* ShapeContext3DEstimation::initCompute, azimuth 12 / elevation 11 / radius 15 bins */
static void sc3d_tables(double search_radius, double min_radius, Sc3dTables &t) {
  const size_t azimuth_bins = 12, elevation_bins = 11, radius_bins = 15;
  float azimuth_interval = 360.0f / static_cast<float>(azimuth_bins);
  float elevation_interval = 180.0f / static_cast<float>(elevation_bins);
  for (size_t j = 0; j < radius_bins + 1; j++)
    t.radii[j] = static_cast<float>(exp(log(min_radius) + ((static_cast<float>(j) / static_cast<float>(radius_bins)) *
                                                           log(search_radius / min_radius))));
  for (size_t k = 0; k < elevation_bins + 1; k++) t.theta[k] = static_cast<float>(k) * elevation_interval;
  for (size_t l = 0; l < azimuth_bins + 1; l++) t.phi[l] = static_cast<float>(l) * azimuth_interval;
  float integr_phi = deg2rad_f(t.phi[1]) - deg2rad_f(t.phi[0]);
  float e = 1.0f / 3.0f;
  for (size_t j = 0; j < radius_bins; j++) {
    float integr_r = (t.radii[j + 1] * t.radii[j + 1] * t.radii[j + 1] / 3.0f) -
                     (t.radii[j] * t.radii[j] * t.radii[j] / 3.0f);
    for (size_t k = 0; k < elevation_bins; k++) {
      float integr_theta = cosf(deg2rad_f(t.theta[k])) - cosf(deg2rad_f(t.theta[k + 1]));
      float V = integr_phi * integr_theta * integr_r;
      for (size_t l = 0; l < azimuth_bins; l++)
        t.lut[(l * elevation_bins * radius_bins) + k * radius_bins + j] = 1.0f / powf(V, e);
    }
  }
}

/* Eigen fixed-size-3 reductions: redux_novec_unroller => c0 + (c1 + c2) (A.8-15) */
struct V3 {
  float v[3];
};
static inline float dot3(const V3 &a, const V3 &b) { return a.v[0] * b.v[0] + (a.v[1] * b.v[1] + a.v[2] * b.v[2]); }
static inline float sqnorm3(const V3 &a) { return dot3(a, a); }
static inline void normalize3(V3 &a, bool eigen32 = false) { /* Eigen 3.3: z = squaredNorm(); if (z > 0) *this /= sqrt(z); 3.2: *this /= norm() */
  float z = sqnorm3(a);
  if (z > 0.0f || eigen32) {
    float s = std::sqrt(z);
    a.v[0] /= s;
    a.v[1] /= s;
    a.v[2] /= s;
  }
}
static inline V3 cross3(const V3 &a, const V3 &b) {
  V3 c;
  c.v[0] = a.v[1] * b.v[2] - a.v[2] * b.v[1];
  c.v[1] = a.v[2] * b.v[0] - a.v[0] * b.v[2];
  c.v[2] = a.v[0] * b.v[1] - a.v[1] * b.v[0];
  return c;
}

struct Result {
  fx_params p;
  Cloud rotated, filtered, keypoints_full, keypoints, kpc;
  std::vector<uint32_t> cand_size, kp_size, kp_neighbors, kpc_cand;
  std::vector<int32_t> cand_keypoint;
  std::vector<float> descriptors; /* K x 1989 */
  std::vector<int32_t> ring_labels; /* n_rings x n_filtered */
};

/* ref: node.cpp:261-327 getCylinderSegments */
static void get_cylinder_segments(const fx_params &P, const Cloud &cloud, int search_kind, Cloud &keypoints,
                                  std::vector<uint32_t> &sizes, Cloud &keypoint_cloud,
                                  std::vector<uint32_t> &kpc_cand, uint32_t cand_base,
                                  std::vector<int> *all_labels) {
  if (all_labels) all_labels->assign(cloud.size(), -1);
  if (cloud.size() <= 0) return;
  std::vector<Cluster> clusterIndices;
  euclidean_clusters(cloud, P.cluster_tolerance, P.cluster_min_count, P.cluster_max_count, search_kind,
                     clusterIndices, all_labels);
  if (clusterIndices.size() <= 0) return;
  for (size_t i = 0; i < clusterIndices.size(); ++i) {
    P4 pt_centroid = {0, 0, 0, 0};
    Cloud cluster;
    double x, y, z;
    double sumx = 0.0, sumy = 0.0, sumz = 0.0;
    double minx = 1000.0, maxx = -1000.0;
    double miny = 1000.0, maxy = -1000.0;
    int clusterSize = (int)clusterIndices[i].indices.size();
    for (int j = 0; j < clusterSize; ++j) {
      x = cloud[clusterIndices[i].indices[j]].x;
      y = cloud[clusterIndices[i].indices[j]].y;
      z = cloud[clusterIndices[i].indices[j]].z;
      sumx += x;
      sumy += y;
      sumz += z;
      if (x < minx) minx = x;
      if (y < miny) miny = y;
      if (x > maxx) maxx = x;
      if (y > maxy) maxy = y;
      P4 pt;
      pt.x = (float)x;
      pt.y = (float)y;
      pt.z = (float)z;
      pt.i = cloud[clusterIndices[i].indices[j]].i;
      cluster.push_back(pt);
    }
    double diameter = pow(pow(maxx - minx, 2) + pow(maxy - miny, 2), 0.5);
    if (diameter < (2 * P.cluster_radius_threshold)) {
      pt_centroid.x = (float)(sumx / ((double)clusterSize));
      pt_centroid.y = (float)(sumy / ((double)clusterSize));
      pt_centroid.z = (float)(sumz / ((double)clusterSize));
      pt_centroid.i = cloud[clusterIndices[i].indices[0]].i;
      uint32_t cand_id = cand_base + (uint32_t)keypoints.size();
      keypoints.push_back(pt_centroid);
      sizes.push_back((uint32_t)clusterSize);
      for (const P4 &m : cluster) {
        keypoint_cloud.push_back(m);
        kpc_cand.push_back(cand_id);
      }
    }
  }
}

/* ref: node.cpp:185-259 estimateKeypoints */
static void estimate_keypoints(Result &R, int search_kind) {
  const fx_params &P = R.p;
  const Cloud &cloud = R.filtered;
  Cloud &keypoints_full = R.keypoints_full;
  R.ring_labels.assign((size_t)std::max(P.n_rings, 0) * cloud.size(), -1);
  for (int i = 0; i < P.n_rings; ++i) {
    Cloud cylinderCentroids, cylinderCloud, channel;
    std::vector<uint32_t> sizes, kpc_cand;
    /* ref: :200 channelElevationDegrees = (i-7)*2-1; generalised: el0 + i*step, window +- step/2 */
    double channelElevationDegrees = P.el0_deg + (double)i * P.el_step_deg;
    double half = P.el_step_deg / 2.0;
    std::vector<int> kept;
    passthrough(cloud, FI_, channelElevationDegrees - half, channelElevationDegrees + half, channel, &kept);
    std::vector<int> labels;
    get_cylinder_segments(P, channel, search_kind, cylinderCentroids, sizes, cylinderCloud, kpc_cand,
                          (uint32_t)keypoints_full.size(), &labels);
    for (size_t j = 0; j < kept.size(); ++j)
      R.ring_labels[(size_t)i * cloud.size() + kept[j]] = labels[j] < 0 ? -1 : kept[labels[j]];
    keypoints_full.insert(keypoints_full.end(), cylinderCentroids.begin(), cylinderCentroids.end());
    R.cand_size.insert(R.cand_size.end(), sizes.begin(), sizes.end());
    R.kpc.insert(R.kpc.end(), cylinderCloud.begin(), cylinderCloud.end());
    R.kpc_cand.insert(R.kpc_cand.end(), kpc_cand.begin(), kpc_cand.end());
  }
  R.cand_keypoint.assign(keypoints_full.size(), -1);
  if (keypoints_full.size() <= 0) return;

  /* Combine keypoints within same proximity (ref: :212-232) */
  std::vector<double> zhold(keypoints_full.size());
  for (size_t i = 0; i < keypoints_full.size(); ++i) {
    zhold[i] = keypoints_full[i].z;
    keypoints_full[i].z = (float)(keypoints_full[i].i * 0.75 * P.cluster_radius_threshold / 2);
  }
  std::vector<Cluster> clusterIndices;
  euclidean_clusters(keypoints_full, P.cluster_radius_threshold, P.number_detection_channels, P.secondary_max,
                     search_kind, clusterIndices);
  for (size_t i = 0; i < keypoints_full.size(); ++i) keypoints_full[i].z = (float)zhold[i];
  if (clusterIndices.size() <= 0) return;

  for (size_t i = 0; i < clusterIndices.size(); ++i) {
    P4 pt_centroid = {0, 0, 0, 0};
    double sumx = 0.0, sumy = 0.0, sumz = 0.0;
    int clusterSize = (int)clusterIndices[i].indices.size();
    for (int j = 0; j < clusterSize; ++j) {
      sumx += keypoints_full[clusterIndices[i].indices[j]].x;
      sumy += keypoints_full[clusterIndices[i].indices[j]].y;
      sumz += keypoints_full[clusterIndices[i].indices[j]].z;
      R.cand_keypoint[clusterIndices[i].indices[j]] = (int32_t)i;
    }
    pt_centroid.x = (float)(sumx / ((double)clusterSize));
    pt_centroid.y = (float)(sumy / ((double)clusterSize));
    pt_centroid.z = (float)(sumz / ((double)clusterSize));
    pt_centroid.i = keypoints_full[clusterIndices[i].indices[0]].i;
    R.keypoints.push_back(pt_centroid);
    R.kp_size.push_back((uint32_t)clusterSize);
  }
}

/* ref: node.cpp:329-355 estimateDescriptors == pcl::ShapeContext3DEstimation::compute (A.8) */
static void estimate_descriptors(Result &R, int search_kind, int trig_policy) {
  const int trig_kind = trig_policy & 0xf;
  const bool skip_epsilon = (trig_policy & FXO_POLICY_SKIP_EPSILON) != 0, eigen32 = (trig_policy & FXO_POLICY_EIGEN32_NORMALIZE) != 0,
             std_uniform = (trig_policy & FXO_POLICY_STD_UNIFORM_FLOAT) != 0;
  const fx_params &P = R.p;
  const Cloud &surface = R.rotated; /* cloud_full: unfiltered, rotated (ref: :115) */
  const Cloud &input = R.keypoints;
  R.kp_neighbors.assign(input.size(), 0);
  if (input.size() <= 0) return;
  R.descriptors.assign(input.size() * (size_t)FX_DESC_FLOATS, 0.0f);

  const size_t azimuth_bins = 12, elevation_bins = 11, radius_bins = 15;
  const double search_radius = P.descriptor_radius;          /* ref: :350 */
  const double min_radius = P.descriptor_radius / 10.0;      /* ref: :351 */
  const double density_radius = P.descriptor_radius / 5.0;   /* ref: :352 */
  Sc3dTables T;
  sc3d_tables(search_radius, min_radius, T);
  std::mt19937 rng(12345u); /* boost::mt19937 seeded 12345u, fresh per compute() (ref: :343) */
  std::uniform_real_distribution<float> uni01(0.0f, 1.0f); /* (FXO_POLICY_STD_UNIFORM_FLOAT: PCL >= 1.10) */
  auto rnd = [&]() -> double {
    if (std_uniform) return (double)uni01(rng);
    return (double)rng() * (1.0 / 4294967296.0); /* boost::uniform_01 */
  };

  std::unique_ptr<Searcher> tree = make_searcher(surface, search_kind);
  const float r2_search = radius2(search_radius);
  const float r2_density = radius2(density_radius);
  std::vector<DistIndex> nn, dens;

  for (size_t point_index = 0; point_index < input.size(); ++point_index) {
    float *desc = &R.descriptors[point_index * (size_t)FX_DESC_FLOATS];
    float *rf = desc + FX_DESC_BINS;
    const P4 &kp = input[point_index];
    if (!finite3(kp)) {
      for (size_t i = 0; i < (size_t)FX_DESC_BINS; ++i) desc[i] = std::numeric_limits<float>::quiet_NaN();
      continue;
    }
    tree->radius(kp, r2_search, nn);
    std::sort(nn.begin(), nn.end()); /* search::KdTree() default: sorted results */
    const size_t neighb_cnt = nn.size();
    R.kp_neighbors[point_index] = (uint32_t)neighb_cnt;
    if (neighb_cnt == 0) {
      for (size_t i = 0; i < (size_t)FX_DESC_BINS; ++i) desc[i] = std::numeric_limits<float>::quiet_NaN();
      continue; /* no RNG draw */
    }
    /* minIndex only selects the normal, and every normal is (0,0,1) (ref: :337-340) */
    V3 normal = {{0.0f, 0.0f, 1.0f}};
    V3 origin = {{kp.x, kp.y, kp.z}};
    V3 x_axis;
    x_axis.v[0] = static_cast<float>(rnd());
    x_axis.v[1] = static_cast<float>(rnd());
    x_axis.v[2] = static_cast<float>(rnd());
    /* !equal(normal[2], 0) branch */
    x_axis.v[2] = -(normal.v[0] * x_axis.v[0] + normal.v[1] * x_axis.v[1]) / normal.v[2];
    normalize3(x_axis, eigen32);

    for (size_t ne = 0; ne < neighb_cnt; ne++) {
      /* pcl::utils::equal(nn_dists[ne], 0.0f): default tolerance std::numeric_limits<float>::min()
       * (pcl/common/utils.h) — only the point the keypoint sits on is skipped */
      if (std::fabs(nn[ne].d - 0.0f) < (skip_epsilon ? std::numeric_limits<float>::epsilon() : std::numeric_limits<float>::min())) continue;
      const P4 &nbp = surface[nn[ne].idx];
      V3 neighbour = {{nbp.x, nbp.y, nbp.z}};
      float r = sqrtf(nn[ne].d);
      /* pcl::geometry::project(neighbour, origin, normal, proj) */
      V3 po = {{neighbour.v[0] - origin.v[0], neighbour.v[1] - origin.v[1], neighbour.v[2] - origin.v[2]}};
      float lambda = dot3(normal, po);
      V3 proj = {{neighbour.v[0] - lambda * normal.v[0], neighbour.v[1] - lambda * normal.v[1],
                  neighbour.v[2] - lambda * normal.v[2]}};
      proj.v[0] -= origin.v[0];
      proj.v[1] -= origin.v[1];
      proj.v[2] -= origin.v[2];
      normalize3(proj, eigen32);
      V3 cross = cross3(x_axis, proj);
      float cn = std::sqrt(sqnorm3(cross));
      float xd = dot3(x_axis, proj);
      float phi;
      if (trig_kind == FXO_TRIG_LIBM_F32)
        phi = rad2deg_f(atan2f(cn, xd));
      else
        phi = rad2deg_f((float)atan2((double)cn, (double)xd));
      phi = dot3(cross, normal) < 0.f ? (360.0f - phi) : phi;
      V3 no = po;
      normalize3(no, eigen32);
      float theta = dot3(normal, no);
      float tc = std::min(1.0f, std::max(-1.0f, theta));
      if (trig_kind == FXO_TRIG_LIBM_F32)
        theta = rad2deg_f(acosf(tc));
      else
        theta = rad2deg_f((float)acos((double)tc));

      size_t j = 0, k = 0, l = 0;
      for (size_t rad = 1; rad < radius_bins + 1; rad++)
        if (r <= T.radii[rad]) {
          j = rad - 1;
          break;
        }
      for (size_t ang = 1; ang < elevation_bins + 1; ang++)
        if (theta <= T.theta[ang]) {
          k = ang - 1;
          break;
        }
      for (size_t ang = 1; ang < azimuth_bins + 1; ang++)
        if (phi <= T.phi[ang]) {
          l = ang - 1;
          break;
        }
      tree->radius(nbp, r2_density, dens);
      int point_density = (int)dens.size();
      if (point_density == 0) continue;
      float w = (1.0f / static_cast<float>(point_density)) * T.lut[(l * elevation_bins * radius_bins) + (k * radius_bins) + j];
      desc[(l * elevation_bins * radius_bins) + (k * radius_bins) + j] += w;
    }
    memset(rf, 0, sizeof(float) * 9);
  }
}

static Result *run(const fx_params &P, const float *pts, uint32_t n, uint32_t stride, double roll, double pitch,
                   int search_kind, int trig_kind) {
  Result *R = new Result();
  R->p = P;
  Cloud &cloud_full = R->rotated;
  cloud_full.resize(n);
  for (uint32_t i = 0; i < n; ++i) {
    cloud_full[i].x = pts[(size_t)i * stride + 0];
    cloud_full[i].y = pts[(size_t)i * stride + 1];
    cloud_full[i].z = pts[(size_t)i * stride + 2];
    cloud_full[i].i = 0.0f;
  }
  /* getElevationAngles (ref: :87) — sensor frame, before rotation */
  for (uint32_t i = 0; i < n; ++i) cloud_full[i].i = elevation_deg(cloud_full[i].x, cloud_full[i].y, cloud_full[i].z);
  /* rotateCloud (ref: :92): pcl::transformPointCloud, dense branch, scalars left to right */
  float M[9];
  rotation_matrix(roll, pitch, M);
  for (uint32_t i = 0; i < n; ++i) {
    float x = cloud_full[i].x, y = cloud_full[i].y, z = cloud_full[i].z;
    cloud_full[i].x = static_cast<float>(M[0] * x + M[1] * y + M[2] * z + 0.0f);
    cloud_full[i].y = static_cast<float>(M[3] * x + M[4] * y + M[5] * z + 0.0f);
    cloud_full[i].z = static_cast<float>(M[6] * x + M[7] * y + M[8] * z + 0.0f);
  }
  /* *cloud = *cloud_full; filterCloud(cloud) (ref: :97-99, :169-183): z, then y, then x */
  R->filtered = cloud_full;
  passthrough(R->filtered, FZ_, P.z_min, P.z_max, R->filtered);
  passthrough(R->filtered, FY_, P.y_min, P.y_max, R->filtered);
  passthrough(R->filtered, FX_, P.x_min, P.x_max, R->filtered);
  estimate_keypoints(*R, search_kind);
  if (P.estimate_descriptors) estimate_descriptors(*R, search_kind, trig_kind);
  return R;
}

}  // namespace

struct fxo_result {
  Result *r;
};

extern "C" {

fxo_result *fxo_run(const fx_params *p, const float *points, uint32_t n, uint32_t stride_floats, double roll,
                    double pitch, int search_kind, int trig_kind) {
  fxo_result *h = new fxo_result();
  h->r = run(*p, points, n, stride_floats, roll, pitch, search_kind, trig_kind);
  return h;
}
void fxo_free(fxo_result *r) {
  if (!r) return;
  delete r->r;
  delete r;
}
uint32_t fxo_n_filtered(const fxo_result *r) { return (uint32_t)r->r->filtered.size(); }
uint32_t fxo_n_candidates(const fxo_result *r) { return (uint32_t)r->r->keypoints_full.size(); }
uint32_t fxo_n_keypoints(const fxo_result *r) { return (uint32_t)r->r->keypoints.size(); }
uint32_t fxo_n_kpc(const fxo_result *r) { return (uint32_t)r->r->kpc.size(); }
static void copy_cloud(const Cloud &c, float *out) {
  if (!c.empty()) memcpy(out, c.data(), c.size() * sizeof(P4));
}
void fxo_rotated(const fxo_result *r, float *xyzi) { copy_cloud(r->r->rotated, xyzi); }
void fxo_filtered(const fxo_result *r, float *xyzi) { copy_cloud(r->r->filtered, xyzi); }
void fxo_candidates(const fxo_result *r, float *xyzi, uint32_t *size, int32_t *keypoint) {
  copy_cloud(r->r->keypoints_full, xyzi);
  for (size_t i = 0; i < r->r->cand_size.size(); ++i) size[i] = r->r->cand_size[i];
  for (size_t i = 0; i < r->r->cand_keypoint.size(); ++i) keypoint[i] = r->r->cand_keypoint[i];
}
void fxo_kpc(const fxo_result *r, float *xyzi, uint32_t *cand) {
  copy_cloud(r->r->kpc, xyzi);
  for (size_t i = 0; i < r->r->kpc_cand.size(); ++i) cand[i] = r->r->kpc_cand[i];
}
void fxo_keypoints(const fxo_result *r, float *xyzi, uint32_t *size, uint32_t *neighbors) {
  copy_cloud(r->r->keypoints, xyzi);
  for (size_t i = 0; i < r->r->kp_size.size(); ++i) size[i] = r->r->kp_size[i];
  for (size_t i = 0; i < r->r->kp_neighbors.size(); ++i) neighbors[i] = r->r->kp_neighbors[i];
}
void fxo_descriptors(const fxo_result *r, float *d) {
  if (!r->r->descriptors.empty()) memcpy(d, r->r->descriptors.data(), r->r->descriptors.size() * sizeof(float));
}
void fxo_ring_labels(const fxo_result *r, int32_t *labels) {
  if (!r->r->ring_labels.empty())
    memcpy(labels, r->r->ring_labels.data(), r->r->ring_labels.size() * sizeof(int32_t));
}

void fxo_rotation(double roll, double pitch, float R[9]) { rotation_matrix(roll, pitch, R); }
void fxo_sc3d_tables(double R, float *radii16, float *theta12, float *phi13, float *lut1980) {
  Sc3dTables T;
  sc3d_tables(R, R / 10.0, T);
  memcpy(radii16, T.radii, sizeof(T.radii));
  memcpy(theta12, T.theta, sizeof(T.theta));
  memcpy(phi13, T.phi, sizeof(T.phi));
  memcpy(lut1980, T.lut, sizeof(T.lut));
}
void fxo_sc3d_rng(uint32_t n_draws, uint32_t *u32_out, float *f32_out) {
  std::mt19937 rng(12345u);
  for (uint32_t i = 0; i < n_draws; ++i) {
    uint32_t u = (uint32_t)rng();
    u32_out[i] = u;
    f32_out[i] = static_cast<float>((double)u * (1.0 / 4294967296.0));
  }
}
float fxo_radius2(double r) { return radius2(r); }
float fxo_cluster_radius2(double tol) { return radius2((double)static_cast<float>(tol)); }
void fxo_sort_by_size_desc(const uint32_t *sizes, uint32_t n, uint32_t *perm_out) {
  /* the permutation std::sort produces depends only on the comparison outcomes, not on the
   * element type, so (size, ordinal) records stand in for pcl::PointIndices */
  struct Rec {
    uint32_t size, id;
  };
  std::vector<Rec> v(n);
  for (uint32_t i = 0; i < n; ++i) v[i] = {sizes[i], i};
  std::sort(v.rbegin(), v.rend(), [](const Rec &a, const Rec &b) { return a.size < b.size; });
  for (uint32_t i = 0; i < n; ++i) perm_out[i] = v[i].id;
}
float fxo_elevation_deg(float x, float y, float z) { return elevation_deg(x, y, z); }

/* CPU baseline: `threads` host threads run the whole pipeline (kd-tree search) on the given scans,
 * cycled, one scan per thread at a time, for about `seconds` of wall time.  Returns elapsed seconds. */
double fxo_bench_throughput(const fx_params *p, const float *points, uint32_t n_scans, uint32_t n, uint32_t stride_floats,
                            double roll, double pitch, uint32_t threads, double seconds, uint64_t *scans_done,
                            uint64_t *keypoints) {
  std::atomic<uint64_t> next(0), done(0), kps(0);
  const auto t0 = std::chrono::steady_clock::now();
  auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  auto worker = [&]() {
    while (elapsed() < seconds) {
      const uint64_t i = next.fetch_add(1);
      Result *r = run(*p, points + (size_t)(i % n_scans) * n * stride_floats, n, stride_floats, roll, pitch,
                      FXO_SEARCH_KDTREE, FXO_TRIG_F64_ROUNDED);
      kps.fetch_add(r->keypoints.size());
      done.fetch_add(1);
      delete r;
    }
  };
  std::vector<std::thread> pool;
  for (uint32_t t = 0; t < threads; ++t) pool.emplace_back(worker);
  for (auto &t : pool) t.join();
  *scans_done = done.load();
  *keypoints = kps.load();
  return elapsed();
}

/* McIlroy's "killer adversary for quicksort" run against the very std::sort call of A.6: returns
 * a size sequence that drives libstdc++'s introsort into its depth limit (heap-sort fallback),
 * so the tests can exercise that branch of the replay too. */
void fxo_antiqsort(uint32_t n, uint32_t *sizes_out) {
  std::vector<int> val(n), ptr(n);
  const int gas = (int)n - 1;
  int nsolid = 0, candidate = 0;
  for (uint32_t i = 0; i < n; ++i) {
    ptr[i] = (int)i;
    val[i] = gas;
  }
  auto less = [&](int x, int y) {
    if (val[x] == gas && val[y] == gas) {
      if (x == candidate)
        val[x] = nsolid++;
      else
        val[y] = nsolid++;
    }
    if (val[x] == gas)
      candidate = x;
    else if (val[y] == gas)
      candidate = y;
    return val[x] < val[y];
  };
  std::sort(ptr.rbegin(), ptr.rend(), less);
  for (uint32_t i = 0; i < n; ++i) sizes_out[i] = (uint32_t)val[i] + 1u;
}

} /* extern "C" */
