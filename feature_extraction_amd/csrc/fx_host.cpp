// fx_host.cpp — host-only pieces of the C-ABI: parameter presets, the rotation matrix the
// node builds from roll/pitch, the 3DSC lookup tables and RNG stream, and the synthetic
// scan generator.  Everything here is tiny per-context set-up work; the per-scan hot path
// lives in fx_kernels.hip.
#include <cmath>
#include <cstring>
#include <random>

#include "../../include/fx.h"
#include "fx_sort_replay.h"

extern "C" {

uint32_t fx_version(void) { return (FX_VERSION_MAJOR << 16) | FX_VERSION_MINOR; }

const char *fx_status_str(fx_status s) {
  switch (s) {
    case FX_OK: return "ok";
    case FX_ERR_INVALID_ARG: return "invalid argument";
    case FX_ERR_NO_DEVICE: return "no usable HIP device (the HIP path is mandatory; there is no CPU fallback)";
    case FX_ERR_HIP: return "HIP runtime error";
    case FX_ERR_OOM: return "out of memory";
    case FX_ERR_TOO_LARGE: return "batch or scan exceeds the context limits";
  }
  return "unknown";
}

// ref: src/feature_extraction_node.cpp:9-34
void fx_params_default(fx_params *p) {
  std::memset(p, 0, sizeof(*p));
  p->cloud_leveling = 1;
  p->x_min = 0.0;
  p->x_max = 75.0;
  p->y_min = -30.0;
  p->y_max = 30.0;
  p->z_min = -1.5;
  p->z_max = 5.0;
  p->cluster_tolerance = 0.65;
  p->cluster_min_count = 5;
  p->cluster_max_count = 50;
  p->cluster_radius_threshold = 0.15;
  p->number_detection_channels = 1;
  p->estimate_descriptors = 1;
  p->descriptor_radius = 2.5;
  // ref: node.cpp:195 (16 channels), :200 ((i-7)*2-1 => -15 + 2 i), :227 (max 16)
  p->n_rings = 16;
  p->el0_deg = -15.0;
  p->el_step_deg = 2.0;
  p->secondary_max = 16;
}

// ref: launch/keypoint_playback.launch:17-33
void fx_params_launch(fx_params *p) {
  fx_params_default(p);
  p->cloud_leveling = 1;
  p->cluster_tolerance = 1.0;
  p->cluster_min_count = 1;
  p->cluster_max_count = 1000;
  p->cluster_radius_threshold = 0.2;
  p->number_detection_channels = 2;
  p->x_max = 100.0;
  p->x_min = 0.0;
  p->y_max = 50.0;
  p->y_min = -50.0;
  p->z_max = 4.0;
  p->z_min = -1.5;
  p->descriptor_radius = 2.5;
}

void fx_limits_default(fx_limits *l, uint32_t max_batch, uint32_t max_points) {
  std::memset(l, 0, sizeof(*l));
  l->max_batch = max_batch;
  l->max_points = max_points;
  l->max_ring_points = 2048;
  l->max_ring_candidates = 256;
  l->max_candidates = 2048;
  l->max_keypoints = 256;
  // support-list entries per descriptor row (16 B each; the pool is max_total_keypoints rows): VLP-16-sized scans
  // stay far below 1024; dense many-ring scans get 4096.  Not a cap on the support set: what does not fit a row's list
  // goes to the scan's overflow region and the row to the dense tier.
  l->max_neighbors = max_points > 65536u ? 4096u : 1024u;
  l->max_total_keypoints = max_batch * 64u;
  l->max_kpc_points = 4096;
  // dense tier pools: as many entries as the batch has points (16 + 4 + 8 bytes each) — a batch averages its scans — but
  // never fewer than 32 scans' worth: a context for one scan at a time must hold a scan whose keypoints' support sets
  // overlap many times over (hundreds of keypoints, descriptor radius beyond a metre: the differential fuzz finds them)
  const unsigned long long dp = (unsigned long long)(max_batch > 32u ? max_batch : 32u) * max_points;
  l->max_dense_points = dp > 0xfff00000ull ? 0xfff00000u : (uint32_t)dp;
  // a scan's overflow region (list entries beyond max_neighbors, any row of the scan): one entry per point of the scan — in
  // contexts of fewer than 32 scans as many more as keep the regions at 32 scans' worth together (a single scan whose rows
  // overflow their lists many times over has no batch to average with: descriptor radii of 2-3 m on the differential fuzz's
  // scenes need up to three entries per point)
  const unsigned long long per = (unsigned long long)max_points * (max_batch >= 32u ? 1u : 32u / (max_batch ? max_batch : 1u));
  l->max_overflow_points = per > 0x7ff00000ull ? 0x7ff00000u : (uint32_t)per;
}

void fx_limits_sparse(fx_limits *l, uint32_t max_batch, uint32_t max_points) {
  fx_limits_default(l, max_batch, max_points);
  // room for a few dense rows a batch (a keypoint beside a wall): 64 rows' worth of the longest lists, four scans' points at least
  const unsigned long long dp = 4ull * max_points > 262144ull ? 4ull * max_points : 262144ull;
  if (dp < l->max_dense_points) l->max_dense_points = (uint32_t)dp;
  const uint32_t ovf = max_points < 8192u ? max_points : 8192u;
  if (ovf < l->max_overflow_points) l->max_overflow_points = ovf;
}

// rotateCloud (ref: node.cpp:159-167): Eigen::AngleAxisf(pitch, Y) * Eigen::AngleAxisf(roll, X).
// Eigen turns each angle-axis into a quaternion (w = cos(a/2), v = sin(a/2) axis), multiplies
// them and expands the product to a matrix; with one axis each, the product has the closed
// form q = (cy cx, cy sx, sy cx, -sy sx), every component one float product.
void fx_rotation_from_roll_pitch(double roll, double pitch, float R[9]) {
  const float hp = 0.5f * (float)pitch, hr = 0.5f * (float)roll;
  const float cy = std::cos(hp), sy = std::sin(hp);
  const float cx = std::cos(hr), sx = std::sin(hr);
  const float w = cy * cx, x = cy * sx, y = sy * cx, z = 0.0f - sy * sx;  // (0 - y1 x2: +0 at zero angles, as Eigen's product)
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

// pcl::ShapeContext3DEstimation::initCompute with the node's settings (ref: node.cpp:350-352):
// search radius R, minimal radius R/10; 12 azimuth x 11 elevation x 15 radius bins.
void fx_sc3d_tables(double R, float *radii, float *theta, float *phi, float *lut) {
  const double rmin = R / 10.0;
  const int NA = 12, NE = 11, NR = 15;
  const float az_step = 360.0f / (float)NA, el_step = 180.0f / (float)NE;
  const double lr = std::log(R / rmin), l0 = std::log(rmin);
  for (int j = 0; j <= NR; ++j) {
    const float frac = (float)j / (float)NR;
    radii[j] = (float)std::exp(l0 + (double)frac * lr);
  }
  for (int k = 0; k <= NE; ++k) theta[k] = (float)k * el_step;
  for (int l = 0; l <= NA; ++l) phi[l] = (float)l * az_step;
  const float d2r = 0.017453293f;  // pcl::deg2rad(float)
  const float dphi = phi[1] * d2r - phi[0] * d2r;
  const float third = 1.0f / 3.0f;
  for (int j = 0; j < NR; ++j) {
    const float r1 = radii[j + 1], r0 = radii[j];
    const float dr = (r1 * r1 * r1 / 3.0f) - (r0 * r0 * r0 / 3.0f);
    for (int k = 0; k < NE; ++k) {
      const float dth = cosf(theta[k] * d2r) - cosf(theta[k + 1] * d2r);
      const float vol = dphi * dth * dr;
      const float inv = 1.0f / powf(vol, third);
      for (int l = 0; l < NA; ++l) lut[(l * NE + k) * NR + j] = inv;
    }
  }
}

// 3DSC reference x-axis of keypoint ordinal k: three draws per keypoint from
// boost::mt19937(12345) through boost::uniform_01 (u32 / 2^32); the third draw is replaced
// by -(n.x x0 + n.y x1)/n.z = -0 because every normal is (0,0,1) (ref: node.cpp:337-340),
// then the vector is normalised.  Only x and y survive.
void fx_sc3d_xaxis(uint32_t k, float xy[2]) {
  std::mt19937 gen(12345u);
  gen.discard((unsigned long long)k * 3ull);
  const float a = (float)((double)gen() * (1.0 / 4294967296.0));
  const float b = (float)((double)gen() * (1.0 / 4294967296.0));
  const float c = -0.0f;
  const float n2 = a * a + (b * b + c * c);
  if (n2 > 0.0f) {
    const float n = std::sqrt(n2);
    xy[0] = a / n;
    xy[1] = b / n;
  } else {
    xy[0] = a;
    xy[1] = b;
  }
}

// ---------------------------------------------------------------- synthetic scans
void fx_synth_cfg_vlp16(fx_synth_cfg *c, uint64_t seed) {
  c->n_rings = 16;
  c->n_az = 1800;
  c->el0_deg = -15.0;
  c->el_step_deg = 2.0;
  c->n_poles = 64;
  c->pole_radius = 0.10;
  c->pole_height = 6.0;
  c->x_lo = 3.0;
  c->x_hi = 70.0;
  c->y_lo = -28.0;
  c->y_hi = 28.0;
  c->sensor_height = 1.8;
  c->wall_radius = 90.0;
  c->seed = seed;
}

static inline uint64_t splitmix64(uint64_t &s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

uint32_t fx_synth_scan(const fx_synth_cfg *c, float *out, uint32_t capacity) {
  const uint32_t n = c->n_rings * c->n_az;
  if (n > capacity) return 0;
  uint64_t s = c->seed * 0x2545F4914F6CDD1Dull + 0x1234567ull;
  const uint32_t P = c->n_poles;
  double *px = new double[P ? P : 1], *py = new double[P ? P : 1];
  for (uint32_t i = 0; i < P; ++i) {
    const double u = (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0);
    const double v = (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0);
    px[i] = c->x_lo + u * (c->x_hi - c->x_lo);
    py[i] = c->y_lo + v * (c->y_hi - c->y_lo);
  }
  const double h = c->sensor_height, top = c->pole_height - h, pr2 = c->pole_radius * c->pole_radius;
  const double kDeg = 3.14159265358979323846 / 180.0;
  for (uint32_t a = 0; a < c->n_az; ++a) {
    const double az = (double)a * 360.0 / (double)c->n_az * kDeg;
    const double ca = std::cos(az), sa = std::sin(az);
    for (uint32_t e = 0; e < c->n_rings; ++e) {
      const double el = (c->el0_deg + (double)e * c->el_step_deg) * kDeg;
      const double ce = std::cos(el), se = std::sin(el);
      const double dx = ce * ca, dy = ce * sa, dz = se;
      double t = c->wall_radius / ce;  // enclosing cylinder: always hit
      if (dz < 0.0) {
        const double tg = -h / dz;
        if (tg < 100.0 && tg < t) t = tg;
      }
      const double a2 = dx * dx + dy * dy;
      for (uint32_t i = 0; i < P; ++i) {
        const double b = dx * px[i] + dy * py[i];
        const double cc = px[i] * px[i] + py[i] * py[i] - pr2;
        const double disc = b * b - a2 * cc;
        if (disc <= 0.0) continue;
        const double tp = (b - std::sqrt(disc)) / a2;
        if (tp <= 0.0 || tp >= t) continue;
        const double z = tp * dz;
        if (z < -h || z > top) continue;
        t = tp;
      }
      float *o = out + ((size_t)a * c->n_rings + e) * 4;
      o[0] = (float)(t * dx);
      o[1] = (float)(t * dy);
      o[2] = (float)(t * dz);
      o[3] = 0.0f;
    }
  }
  delete[] px;
  delete[] py;
  return n;
}

#ifdef FX_TEST_HOOKS
// Host build of the order-replay used by the kernels (same header, same code path), so the
// CPU test-suite can check it against libstdc++'s std::sort without a GPU.
void fx_test_sort_replay(const uint32_t *sizes, uint32_t n, uint32_t *perm_out) {
  uint32_t *rec = new uint32_t[n ? n : 1];
  for (uint32_t i = 0; i < n; ++i) rec[i] = (sizes[i] << 16) | i;
  int stk[FX_SORT_STACK_WORDS];
  fx_sort_replay_desc(rec, n, stk);
  for (uint32_t i = 0; i < n; ++i) perm_out[i] = rec[i] & 0xffffu;
  delete[] rec;
}
// The variant the kernels use: sequential partition phase + stable ranking.
void fx_test_sort_replay_ranked(const uint32_t *sizes, uint32_t n, uint32_t *perm_out) {
  uint32_t *rec = new uint32_t[2 * (n ? n : 1)];
  for (uint32_t i = 0; i < n; ++i) rec[i] = (sizes[i] << 16) | i;
  int stk[FX_SORT_STACK_WORDS];
  fx_sort_replay_desc_ranked(rec, n, stk, rec + n);
  for (uint32_t i = 0; i < n; ++i) perm_out[i] = rec[i] & 0xffffu;
  delete[] rec;
}
// Phase 1 through the position-list statement of the partition step (the rule the wavefront version
// in fx_kernels.hip implements with ballots), then the stable ranking.
void fx_test_sort_replay_lists(const uint32_t *sizes, uint32_t n, uint32_t *perm_out) {
  uint32_t *rec = new uint32_t[2 * (n ? n : 1)];
  uint16_t *pos = new uint16_t[2 * (n ? n : 1)];
  for (uint32_t i = 0; i < n; ++i) rec[i] = (sizes[i] << 16) | i;
  int stk[FX_SORT_STACK_WORDS];
  if (n >= 2) {
    fx_sort_detail::RevView v{rec, (int)n};
    fx_sort_partition_phase_lists(v, (int)n, stk, pos, pos + n);
    uint32_t *tmp = rec + n;
    for (uint32_t c = 0; c < n; ++c) {
      const uint32_t sz = rec[c] >> 16;
      uint32_t at = 0;
      for (uint32_t d = 0; d < n; ++d) {
        const uint32_t sd = rec[d] >> 16;
        at += (sd > sz || (sd == sz && d < c)) ? 1u : 0u;
      }
      tmp[at] = rec[c];
    }
    for (uint32_t c = 0; c < n; ++c) rec[c] = tmp[c];
  }
  for (uint32_t i = 0; i < n; ++i) perm_out[i] = rec[i] & 0xffffu;
  delete[] rec;
  delete[] pos;
}

#endif  // FX_TEST_HOOKS

}  // extern "C"
