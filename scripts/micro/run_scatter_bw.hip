// Write pattern of a one-pass MSD partition: every workgroup (tile of 16384 (key, payload) pairs) writes one run of
// 16384 / NB pairs into each of NB bucket segments.  How does the store rate depend on the run length (NB) and on whether
// the destination (pairs x 8 B per outcome, times G outcomes in flight) fits the 256 MiB Infinity Cache?
// Build: hipcc --offload-arch=gfx950 -O3 run_scatter_bw.hip -o run_scatter_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// grid (tiles, G); bucket b's segment holds tiles * R pairs, tile t writes [t * R, (t + 1) * R) of it (stride: seg_stride pairs)
__global__ __launch_bounds__(1024) void scatter_runs(u32x2* __restrict__ dst, int nb, int run, size_t seg_stride, size_t outcome_stride) {
  u32x2* o = dst + blockIdx.y * outcome_stride;
  const int t = blockIdx.x;
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int idx = k * 1024 + threadIdx.x;
    const int b = idx / run, r = idx - b * run;
    // spread buckets over the tile the way digit-sorted data leaves LDS: idx order = bucket order
    o[static_cast<size_t>(b) * seg_stride + static_cast<size_t>(t) * run + r] = u32x2{static_cast<unsigned>(idx), static_cast<unsigned>(t)};
  }
  (void)nb;
}

int main() {
  const size_t M = 8386560;               // 4096 * 4095 / 2
  const int tiles = static_cast<int>((M + 16383) / 16384);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int G : {1, 2, 4, 16}) {
    for (int slack : {1, 2}) {
      u32x2* dst;
      const size_t per_outcome = static_cast<size_t>(tiles) * 16384 * slack;
      if (hipMalloc(&dst, per_outcome * 8 * G) != hipSuccess) { printf("alloc failed\n"); return 1; }
      for (int nb : {64, 256, 1024, 2048, 4096, 8192, 16384}) {
        const int run = 16384 / nb;
        const size_t seg_stride = static_cast<size_t>(tiles) * run * slack;
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
          hipEventRecord(e0);
          hipLaunchKernelGGL(scatter_runs, dim3(tiles, G), dim3(1024), 0, 0, dst, nb, run, seg_stride, per_outcome);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (rep > 0 && ms < best) best = ms;
        }
        const double bytes = static_cast<double>(tiles) * 16384 * 8 * G;
        printf("G=%2d slack=%d buckets=%5d run=%4d pairs (%5d B): %.3f ms = %.1f us per outcome, %.0f GB/s\n", G, slack, nb, run, run * 8, best, best * 1e3 / G, bytes / 1e6 / best);
      }
      hipFree(dst);
    }
  }
  return 0;
}
