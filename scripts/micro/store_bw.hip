// Achievable HBM store bandwidth on this card for the head's access pattern: every workgroup streams 16-byte stores
// over a contiguous range (no loads, no arithmetic).  Variants: plain global_store_dwordx4, buffer stores, and
// non-temporal stores; grid sized like the head launch.  Build: hipcc --offload-arch=gfx950 -O3 store_bw.hip -o store_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void fill(float* __restrict__ out, size_t n_vec, size_t per_block) {
  const size_t b0 = static_cast<size_t>(blockIdx.x) * per_block;
  const size_t b1 = b0 + per_block < n_vec ? b0 + per_block : n_vec;
  const f32x4 v = {1.f, 2.f, 3.f, static_cast<float>(blockIdx.x)};
  f32x4* o = reinterpret_cast<f32x4*>(out);
  for (size_t i = b0 + threadIdx.x; i < b1; i += 256) {
    if (MODE == 0) o[i] = v;
    else __builtin_nontemporal_store(v, &o[i]);
  }
}

int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? strtoull(argv[1], nullptr, 10) : (60ull << 30);
  float* out;
  if (hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  const size_t n_vec = bytes / 16;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (size_t blocks : {size_t(2048), size_t(8192), size_t(28672), size_t(131072)}) {
      const size_t per_block = (n_vec + blocks - 1) / blocks;
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(fill<0>, dim3(blocks), dim3(256), 0, 0, out, n_vec, per_block);
        else hipLaunchKernelGGL(fill<1>, dim3(blocks), dim3(256), 0, 0, out, n_vec, per_block);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      printf("mode=%s blocks=%zu bytes=%.1f GB  %.3f ms  %.0f GB/s\n", mode ? "nontemporal" : "plain", blocks, bytes / 1e9, best, bytes / 1e6 / best);
    }
  return 0;
}
