// What bounds the all-pairs head when scores are stored?  A loop with the head's store stream (per wave and stage: 32 x
// buffer_store_dword, lane = column, two 128-B row segments per instruction, 256 rows x 64 columns per workgroup and stage)
// and a tunable amount of matrix work / LDS operand traffic beside it, no barriers, no loads.  Reports time, bytes/s and
// the shader clock held during the loop (s_memtime / s_memrealtime stamps in a buffer nothing else reads).
//   hipcc --offload-arch=gfx950 -O3 store_mfma.hip -o store_mfma && ./store_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct Args { float* out; unsigned long long* stamps; int n, labels; };

// SHAPE 0: v_mfma_f32_32x32x16_bf16, NM per stage.  SHAPE 1: v_mfma_f32_16x16x32_bf16, 2*NM per stage (same flops).
// LDSR: ds_read_b128 per MFMA (0, 1 or 2): operands re-read from LDS like the head's B fragments.
template <int NM, int SHAPE, int LDSR>
__global__ __launch_bounds__(512, 2) void k(const Args p) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  for (int i = tid; i < 65536 / 16; i += 512) reinterpret_cast<f32x4*>(smem)[i] = f32x4{1.0f * i, 0.5f, 0.25f, 2.f};
  __syncthreads();
  const long long l = blockIdx.y, row0 = (long long)blockIdx.x * 256;
  float* slab = p.out + (l * p.n + row0) * p.n;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(slab, 0, (int)(256ll * p.n * 4), 0x00020000);
  bf16x8 a[8], b;
  for (int s = 0; s < 8; ++s) for (int j = 0; j < 8; ++j) a[s][j] = (__bf16)(0.01f * ((lane * 7 + s * 3 + j) % 13) - 0.05f);
  for (int j = 0; j < 8; ++j) b[j] = (__bf16)(0.02f * ((lane + j) % 11) - 0.1f);
  const int nst = p.n / 64;
  unsigned long long t0 = 0, r0 = 0;
  if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  f32x16 acc[2];
  f32x4 acc4[8];
  for (int t = 0; t < 2; ++t) for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
  for (int t = 0; t < 8; ++t) acc4[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int start = (blockIdx.x * 5u + blockIdx.y * 3u) % (unsigned)nst;
  for (int s0 = 0; s0 < nst; ++s0) {
    int s = s0 + start; if (s >= nst) s -= nst;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      bf16x8 bb = b;
      if constexpr (LDSR >= 1) bb = *reinterpret_cast<const bf16x8*>(smem + ((lane * 16 + i * 1024 + s0 * 64) & 65535 & ~15));
      if constexpr (LDSR >= 2) {
        const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(smem + ((lane * 16 + i * 1024 + 32768 + s0 * 64) & 65535 & ~15));
        for (int j = 0; j < 8; ++j) bb[j] = bb[j] + b2[j];
      }
      if constexpr (SHAPE == 0) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 7], bb, acc[i & 1], 0, 0, 0);
      else {
        acc4[(2 * i) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 7], bb, acc4[(2 * i) & 7], 0, 0, 0);
        acc4[(2 * i + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + 1) & 7], bb, acc4[(2 * i + 1) & 7], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int row = wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
        const long long e = (long long)row * p.n + s * 64 + 32 * t + r;
        float val = SHAPE == 0 ? acc[t][v] : acc4[(t * 4 + (v >> 2)) & 7][v & 3];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rsrc, (unsigned)(e * 4), 0, 0);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    const size_t w = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    p.stamps[2 * w] = t1 - t0;
    p.stamps[2 * w + 1] = r1 - r0;
  }
}

template <int NM, int SHAPE, int LDSR>
void run(const Args& a, const char* name) {
  const dim3 grid(a.n / 256, a.labels), block(512);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, SHAPE, LDSR>), grid, block, 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0) best = std::min(best, ms);
  }
  const size_t nwg = (size_t)grid.x * grid.y;
  std::vector<unsigned long long> st(2 * nwg);
  hipMemcpy(st.data(), a.stamps, st.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> clk;
  for (size_t i = 0; i < nwg; ++i) if (st[2 * i + 1]) clk.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 100.0);   // MHz
  std::sort(clk.begin(), clk.end());
  const double bytes = (double)a.labels * a.n * a.n * 4;
  printf("%-34s %8.3f ms  %6.0f GB/s  clock(median) %6.0f MHz  -> %6.0f B/cycle chip-wide\n", name, best, bytes / 1e6 / best,
         clk[clk.size() / 2], bytes / (best * 1e-3) / (clk[clk.size() / 2] * 1e6));
  fflush(stdout);
}

int main() {
  Args a; a.n = 4096; a.labels = 896;
  if (hipMalloc(&a.out, (size_t)a.labels * a.n * a.n * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&a.stamps, (size_t)(a.n / 256) * a.labels * 16);
  run<0, 0, 0>(a, "stores only");
  run<16, 0, 0>(a, "16 mfma32 (bf16 head)");
  run<16, 0, 1>(a, "16 mfma32 + 1 ds_read/mfma");
  run<48, 0, 0>(a, "48 mfma32 (bf16x3 head)");
  run<48, 0, 1>(a, "48 mfma32 + 1 ds_read/mfma");
  run<48, 0, 2>(a, "48 mfma32 + 2 ds_read/mfma");
  run<48, 1, 0>(a, "96 mfma16 (same flops)");
  run<48, 1, 1>(a, "96 mfma16 + 1 ds_read/2mfma");
  run<96, 0, 0>(a, "96 mfma32");
  run<0, 0, 0>(a, "stores only (again)");
  return 0;
}
