"""Independent fp64 statement of the 3D shape context descriptor — TEST INFRASTRUCTURE.

Written from the published algorithm, not from oracle/fx_oracle.cpp and not from the kernels:

  A. Frome, D. Huber, R. Kolluri, T. Buelow, J. Malik: "Recognizing Objects in Range Data Using
  Regional Point Descriptors", ECCV 2004, section 2.1 (3D shape contexts):
    * support region = sphere of radius r_max around the basis point, north pole = surface normal;
    * J radial shells with logarithmic boundaries R_j = exp(ln r_min + (j / J) ln(r_max / r_min)),
      K equal elevation divisions of [0, 180] deg, L equal azimuth divisions of [0, 360) deg;
    * a point p_i falling in bin (j, k, l) contributes w(p_i) = 1 / (rho_i * cbrt(V(j, k, l))),
      V = volume of the bin, rho_i = number of points within a small radius delta of p_i.

  and the conventions the PCL class documents for pcl::ShapeContext3DEstimation (the class the
  reference instantiates, ref: src/feature_extraction_node.cpp:343-353): J = 15, K = 11, L = 12,
  r_min = R / 10, delta = R / 5 (ref: :350-352), bin index l * (K * J) + k * J + j, azimuth measured in the
  tangent plane from a reference direction drawn from mt19937(12345) (three draws per keypoint,
  the first two give the direction once it is made orthogonal to the normal), all normals +z
  (ref: :337-340), values exactly on a boundary belong to the lower bin, points closer than r_min to
  the first shell, the basis point itself skipped.

Everything is evaluated in float64 with numpy from those definitions (angles through atan2 of the
tangent-plane coordinates, not through cross products); only set membership (d2 < r2) uses the float32
squared distance, because membership is an integer-exact question.  Used by
tests/test_sc3d_independent.py to cross-check the oracle's descriptor stage, which otherwise has no
check that does not share its author's reading of the algorithm.
"""
import numpy as np

J_BINS, K_BINS, L_BINS = 15, 11, 12


class MT19937:
    """Matsumoto & Nishimura's reference generator (init_genrand seeding), 32-bit outputs."""

    def __init__(self, seed):
        self.mt = [0] * 624
        self.mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            self.mt[i] = (1812433253 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.idx = 624

    def _twist(self):
        mt = self.mt
        for i in range(624):
            y = (mt[i] & 0x80000000) | (mt[(i + 1) % 624] & 0x7FFFFFFF)
            mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
        self.idx = 0

    def u32(self):
        if self.idx >= 624:
            self._twist()
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF


def f32_sqdist(points, q):
    """float32 squared Euclidean distance, summed x, y, z in that order (membership tests only)."""
    d = points.astype(np.float32) - np.asarray(q, np.float32)
    return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]


def bin_volumes(R):
    """Shell boundaries, and V(j, k) (independent of l) as the integral of r^2 sin(theta) over the bin."""
    r_min = R / 10.0
    radii = np.exp(np.log(r_min) + np.arange(J_BINS + 1) / J_BINS * np.log(R / r_min))
    theta = np.radians(np.arange(K_BINS + 1) * (180.0 / K_BINS))
    dphi = np.radians(360.0 / L_BINS)
    shell = (radii[1:] ** 3 - radii[:-1] ** 3) / 3.0             # [J]
    cap = np.cos(theta[:-1]) - np.cos(theta[1:])                 # [K]
    return radii, cap[:, None] * shell[None, :] * dphi           # [K, J]


def describe(surface, keypoints, R, edge_tol_deg=1e-3, edge_tol_m=1e-6):
    """surface: [N, 3] float32 (the rotated, unfiltered cloud); keypoints: [K, 3] float32, in keypoint order.
    Returns (desc [K, 1980] float64 (NaN rows for keypoints without neighbours), n_neighbours [K],
    tainted [K, 1980] bool — bins next to a neighbour that lies within the tolerances of a bin boundary,
    where a float32 evaluation may legitimately choose the other bin)."""
    surface = np.ascontiguousarray(surface, np.float32)
    finite = np.isfinite(surface).all(axis=1)
    surf = surface[finite]
    r2 = np.float32(np.float64(R) * np.float64(R))
    rd = np.float64(R) / 5.0
    r2_density = np.float32(rd * rd)
    radii, vol = bin_volumes(float(R))
    weight_of_bin = 1.0 / np.cbrt(vol)  # [K, J]
    rng = MT19937(12345)
    K = len(keypoints)
    desc = np.zeros((K, J_BINS * K_BINS * L_BINS))
    tainted = np.zeros((K, J_BINS * K_BINS * L_BINS), bool)
    n_nb = np.zeros(K, np.int64)
    surf64 = surf.astype(np.float64)
    for kidx in range(K):
        o = keypoints[kidx, :3]
        d2 = f32_sqdist(surf, o)
        nb = np.where(d2 < r2)[0]
        n_nb[kidx] = len(nb)
        if len(nb) == 0:
            desc[kidx] = np.nan  # and no reference direction is drawn for this keypoint
            continue
        u = [rng.u32() for _ in range(3)]
        ax = np.float32(u[0] / 4294967296.0), np.float32(u[1] / 4294967296.0)
        a0 = np.arctan2(np.float64(ax[1]), np.float64(ax[0]))  # direction of the reference axis in the xy plane
        # everything a density query can reach
        sup = np.where(d2 < np.float32((R * 1.2 + 1e-3) ** 2))[0]
        sup_pts = surf[sup]
        for i in nb:
            if d2[i] < np.finfo(np.float32).tiny:
                continue  # the basis point itself
            v = surf64[i] - o.astype(np.float64)
            r = np.sqrt(np.float64(d2[i]))
            theta = np.degrees(np.arccos(np.clip(v[2] / np.linalg.norm(v), -1.0, 1.0)))
            if v[0] == 0.0 and v[1] == 0.0:
                phi = 0.0  # on the pole: no tangent-plane direction
            else:
                phi = np.degrees(np.arctan2(v[1], v[0]) - a0) % 360.0
            j = int(np.searchsorted(radii[1:], r, side="left"))
            k = int(np.searchsorted(np.arange(1, K_BINS + 1) * (180.0 / K_BINS), theta, side="left"))
            l = int(np.searchsorted(np.arange(1, L_BINS + 1) * (360.0 / L_BINS), phi, side="left"))
            j = 0 if j >= J_BINS else j
            k = 0 if k >= K_BINS else k
            l = 0 if l >= L_BINS else l
            rho = int((f32_sqdist(sup_pts, surf[i]) < r2_density).sum())
            b = (l * K_BINS + k) * J_BINS + j
            desc[kidx, b] += weight_of_bin[k, j] / rho
            near_r = np.abs(radii - r).min() < edge_tol_m
            tm = theta / (180.0 / K_BINS)
            # (0 and 180 deg bound the range: nothing lies on their other side)
            near_t = 0 < np.rint(tm) < K_BINS and np.abs(tm - np.rint(tm)) * (180.0 / K_BINS) < edge_tol_deg
            pm = phi / (360.0 / L_BINS)
            near_p = np.abs(pm - np.rint(pm)) * (360.0 / L_BINS) < edge_tol_deg
            # on or near the pole the azimuth is ill-conditioned: a float32 evaluation can land anywhere (PCL's
            # literal float32 projection leaves a rounding residue along the normal, which normalises to +-z and
            # reads as 90 deg; with no residue it reads as 0 deg)
            near_pole = np.hypot(v[0], v[1]) < 1e-4 * max(r, 1e-12)
            if near_r or near_t or near_p or near_pole:
                js = {j, max(j - 1, 0), min(j + 1, J_BINS - 1)} if near_r else {j}
                ks = {k, max(k - 1, 0), min(k + 1, K_BINS - 1)} if near_t else {k}
                ls = {l, (l - 1) % L_BINS, (l + 1) % L_BINS} if near_p else {l}
                if near_pole:
                    ls = set(range(L_BINS))
                for jj in js:
                    for kk in ks:
                        for ll in ls:
                            tainted[kidx, (ll * K_BINS + kk) * J_BINS + jj] = True
    return desc, n_nb, tainted
