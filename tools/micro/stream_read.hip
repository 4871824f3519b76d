// Diagnostic: practical HBM read bandwidth of the box for the bench's input size (472 MB), to put k_prep's 3.1 TB/s
// beside.  Variants: 16-byte loads, 12-of-16-byte loads (k_prep's pattern), grid sizes.
//   hipcc --offload-arch=gfx950 -O3 -o stream_read tools/micro/stream_read.hip && ./stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float __attribute__((address_space(1))) gfloat;
template <int BYTES>
__global__ void k_read(const float *p, size_t n4, float *out) {
  const gfloat *g = (const gfloat *)p;
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 4 * stride) {
    float v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t j = i + u * stride < n4 ? i + u * stride : n4 - 1;
      v[u][0] = g[4 * j], v[u][1] = g[4 * j + 1], v[u][2] = g[4 * j + 2];
      v[u][3] = BYTES == 16 ? g[4 * j + 3] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u][0] + v[u][1] + v[u][2] + v[u][3];
  }
  if (acc == 12345.678f) out[0] = acc;
}
// k_filter's first mapping: every workgroup streams its own contiguous region (half a scan), 1024 records a wavefront and trip
__global__ void k_read_regions(const float *p, size_t n4, float *out, size_t per_block) {
  const gfloat *g = (const gfloat *)p;
  float acc = 0.f;
  const size_t lo = (size_t)blockIdx.x * per_block, hi = lo + per_block < n4 ? lo + per_block : n4;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (size_t c0 = lo + (size_t)wave * 1024; c0 < hi; c0 += (size_t)nw * 1024) {
    float v[16][3];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      size_t j = c0 + u * 64 + lane;
      j = j < n4 ? j : n4 - 1;
      v[u][0] = g[4 * j], v[u][1] = g[4 * j + 1], v[u][2] = g[4 * j + 2];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u][0] + v[u][1] + v[u][2];
  }
  if (acc == 12345.678f) out[0] = acc;
}
// the same trips dealt in address order: wavefront g of G takes the 1024-record chunks g, g + G, ...
__global__ void k_read_ordered(const float *p, size_t n4, float *out) {
  const gfloat *g = (const gfloat *)p;
  float acc = 0.f;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const size_t G = (size_t)gridDim.x * nw;
  for (size_t c0 = ((size_t)blockIdx.x * nw + wave) * 1024; c0 < n4; c0 += G * 1024) {
    float v[16][3];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      size_t j = c0 + u * 64 + lane;
      j = j < n4 ? j : n4 - 1;
      v[u][0] = g[4 * j], v[u][1] = g[4 * j + 1], v[u][2] = g[4 * j + 2];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u][0] + v[u][1] + v[u][2];
  }
  if (acc == 12345.678f) out[0] = acc;
}
int main() {
  const size_t n4 = (size_t)1024 * 28800;  // float4 records
  float *d, *o;
  const int W = 6;  // windows of 472 MB read in turn: 2.8 GB, far beyond the 256 MB infinity cache
  hipMalloc(&d, n4 * 16 * W);
  hipMalloc(&o, 4);
  hipMemset(d, 1, n4 * 16 * W);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int bytes : {16, 12})
    for (int grid : {512, 2048, 8192})
      for (int bs : {256, 512}) {
        for (int rep = 0; rep < 3; ++rep) {
          if (bytes == 16) hipLaunchKernelGGL(k_read<16>, dim3(grid), dim3(bs), 0, 0, d + (size_t)(rep % W) * n4 * 4, n4, o);
          else hipLaunchKernelGGL(k_read<12>, dim3(grid), dim3(bs), 0, 0, d + (size_t)(rep % W) * n4 * 4, n4, o);
        }
        hipEventRecord(a);
        const int R = 12;
        for (int rep = 0; rep < R; ++rep) {
          if (bytes == 16) hipLaunchKernelGGL(k_read<16>, dim3(grid), dim3(bs), 0, 0, d + (size_t)(rep % W) * n4 * 4, n4, o);
          else hipLaunchKernelGGL(k_read<12>, dim3(grid), dim3(bs), 0, 0, d + (size_t)(rep % W) * n4 * 4, n4, o);
        }
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("load %2d B/record  grid %5d x %3d : %.3f ms  %.2f TB/s (of the 472 MB)\n", bytes, grid, bs, ms / R, n4 * 16 / (ms / R * 1e-3) / 1e12);
      }
  for (int mode = 0; mode < 2; ++mode)
    for (int grid : {512, 1024, 2048, 4096}) {
      const int R = 12;
      for (int rep = 0; rep < R + 3; ++rep) {
        if (rep == 3) hipEventRecord(a);
        const float *src = d + (size_t)(rep % W) * n4 * 4;
        if (mode == 0) hipLaunchKernelGGL(k_read_regions, dim3(grid), dim3(256), 0, 0, src, n4, o, (n4 + grid - 1) / grid);
        else hipLaunchKernelGGL(k_read_ordered, dim3(grid), dim3(256), 0, 0, src, n4, o);
      }
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      printf("%s grid %5d x 256, 16 loads a lane: %.3f ms  %.2f TB/s\n", mode == 0 ? "own region per workgroup" : "chunks in address order ", grid, ms / R,
             n4 * 16 / (ms / R * 1e-3) / 1e12);
    }
  return 0;
}
