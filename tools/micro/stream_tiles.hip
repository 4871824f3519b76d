// Diagnostic: what k_front's streaming pass could reach.  One 512-thread workgroup per scan (28 800 records of 16 bytes), two a
// CU (80 KB of LDS each), tiles of 2048 records with the next tile's loads in flight and one barrier a tile — k_front's shape —
// and nothing else: a few compares per record and a ballot.  Variants of the load instruction:
//   0  dword x + dwordx2 y z under `if (i < n)` (what the compiler makes of k_front's load_tile)
//   1  dwordx3, unconditional (index clamped)
//   2  dwordx4, unconditional
//   3  dwordx4, two tiles of loads in flight
//   hipcc --offload-arch=gfx950 -O2 -o stream_tiles tools/micro/stream_tiles.hip && ./stream_tiles
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float __attribute__((address_space(1))) gfloat;
typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int T = 512, U = 4, TILE = 64 * U * (T / 64);

template <int MODE>
__global__ __launch_bounds__(T) void k_tiles(const float *p, unsigned n, unsigned *out) {
  extern __shared__ unsigned smem[];
  const gfloat *g = (const gfloat *)(p + (size_t)blockIdx.x * n * 4);
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned cnt = 0;
  float vx[U], vy[U], vz[U], nx[U], ny[U], nz[U];
  auto load = [&](unsigned t0, float (&x)[U], float (&y)[U], float (&z)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned i = t0 + wave * (64 * U) + u * 64 + lane;
      const gfloat *q = g + (size_t)(i < n ? i : n - 1) * 4;
      if (MODE == 0) {
        x[u] = i < n ? q[0] : __builtin_nanf("");
        y[u] = q[1], z[u] = q[2];
      } else if (MODE == 1) {
        const f3 v = *(const __attribute__((address_space(1))) f3 *)q;
        x[u] = v.x, y[u] = v.y, z[u] = v.z;
      } else {
        const f4 v = *(const __attribute__((address_space(1))) f4 *)q;
        x[u] = v.x, y[u] = v.y, z[u] = v.z + 0.f * v.w;
      }
    }
  };
  load(0, vx, vy, vz);
  for (unsigned t0 = 0; t0 < n; t0 += TILE) {
    load(t0 + TILE, nx, ny, nz);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool k = vx[u] >= 0.f && vx[u] <= 100.f && vy[u] >= -50.f && vy[u] <= 50.f && vz[u] >= -1.5f && vz[u] <= 4.f;
      cnt += (unsigned)__popcll(__ballot(k));
    }
    if (lane == 0) smem[(t0 / TILE & 1) * 8 + wave] = cnt;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) vx[u] = nx[u], vy[u] = ny[u], vz[u] = nz[u];
  }
  if (threadIdx.x == 0) out[blockIdx.x] = cnt + smem[0];
}

int main() {
  const unsigned n = 28800, BMAX = 1024;
  const int W = 6;
  float *d;
  unsigned *o;
  hipMalloc(&d, (size_t)BMAX * n * 16 * W);
  hipMalloc(&o, BMAX * 4);
  hipMemset(d, 0, (size_t)BMAX * n * 16 * W);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (unsigned B : {1u, 8u, 256u, 512u, 1024u})
  for (int lds_kb : {80, 16})
    for (int mode = 0; mode < 3; ++mode) {
      const int R = 12;
      auto launch = [&](int rep) {
        const float *src = d + (size_t)(rep % W) * BMAX * n * 4;
        const size_t lds = (size_t)lds_kb * 1024;
        if (mode == 0) {
          hipFuncSetAttribute((const void *)k_tiles<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          hipLaunchKernelGGL(k_tiles<0>, dim3(B), dim3(T), lds, 0, src, n, o);
        } else if (mode == 1) {
          hipFuncSetAttribute((const void *)k_tiles<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          hipLaunchKernelGGL(k_tiles<1>, dim3(B), dim3(T), lds, 0, src, n, o);
        } else {
          hipFuncSetAttribute((const void *)k_tiles<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          hipLaunchKernelGGL(k_tiles<2>, dim3(B), dim3(T), lds, 0, src, n, o);
        }
      };
      for (int rep = 0; rep < 3; ++rep) launch(rep);
      hipEventRecord(a);
      for (int rep = 0; rep < R; ++rep) launch(rep);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      printf("%4u scans, LDS %2d KB a workgroup (%s a CU), loads %s: %.4f ms, %.3f TB/s\n", B, lds_kb, lds_kb == 80 ? "two" : lds_kb == 52 ? "three" : "five+",
             mode == 0 ? "dword + dwordx2, conditional" : mode == 1 ? "dwordx3" : "dwordx4", ms / R, (double)B * n * 16 / (ms / R * 1e-3) / 1e12);
    }
  return 0;
}
