// Does s_waitcnt vmcnt(K) after [LDS-DMA load][K younger stores] guarantee that the load has landed?
// variant 3 (control): NO stores, so vmcnt(32) does not wait at all -> stale reads must show up.
// variant 0: the K stores are out of range of their buffer descriptor (dropped); variant 1: real stores to
// hot lines; variant 2: real stores, load from a hot (L2-resident) line.  Prints how many waves saw stale LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void glds16(const void* gsrc, unsigned dst_uniform) {
  unsigned keep; const unsigned dst = __builtin_amdgcn_readfirstlane(dst_uniform);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
template <int VARIANT>
__global__ __launch_bounds__(64) void probe(const unsigned* src, size_t stride_words, unsigned* sink, unsigned* stale, int rounds) {
  __shared__ __attribute__((aligned(16))) unsigned buf[256];
  const int lane = threadIdx.x;
  unsigned n_stale = 0;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(sink + (size_t)blockIdx.x * 4096, 0, 4096 * 4, 0x00020000);
  for (int it = 0; it < rounds; ++it) {
    for (int i = lane; i < 256; i += 64) buf[i] = 0xDEADBEEFu;
    __syncthreads();
    const size_t line = (VARIANT == 2) ? 0 : ((size_t)blockIdx.x * rounds + it) * stride_words;
    glds16(src + line + lane * 4, (unsigned)(size_t)(lds_void*)buf);
    if (VARIANT != 3) {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const unsigned off = (VARIANT == 0) ? 0xFFFFFFFFu : (unsigned)((k * 64 + lane) * 4);
        __builtin_amdgcn_raw_buffer_store_b32(it * 1000u + k, rs, off, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    const unsigned v = buf[lane * 4];
    if (v == 0xDEADBEEFu) n_stale++;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  unsigned any = __ballot(n_stale != 0) != 0;
  if (lane == 0) stale[blockIdx.x] = n_stale ? n_stale : any;
}
int main() {
  const int blocks = 2048, rounds = 64;
  const size_t stride_words = 1 << 14;                       // 64 KB apart: every load a fresh line far from the others
  const size_t words = (size_t)blocks * rounds * stride_words + 1024;
  unsigned *src, *sink, *stale;
  hipMalloc(&src, words * 4); hipMemset(src, 0x11, words * 4);
  hipMalloc(&sink, (size_t)blocks * 4096 * 4); hipMalloc(&stale, blocks * 4);
  unsigned* h = (unsigned*)malloc(blocks * 4);
  for (int variant = 0; variant < 4; ++variant) {
    hipMemset(stale, 0, blocks * 4);
    // evict: touch a big buffer so the source lines are cold again
    hipMemset(sink, variant, (size_t)blocks * 4096 * 4);
    if (variant == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(64), 0, 0, src, stride_words, sink, stale, rounds);
    if (variant == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(64), 0, 0, src, stride_words, sink, stale, rounds);
    if (variant == 3) hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(64), 0, 0, src, stride_words, sink, stale, rounds);
    if (variant == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(64), 0, 0, src, stride_words, sink, stale, rounds);
    hipDeviceSynchronize();
    hipMemcpy(h, stale, blocks * 4, hipMemcpyDeviceToHost);
    long tot = 0, blk = 0; for (int i = 0; i < blocks; ++i) { tot += h[i]; blk += h[i] != 0; }
    printf("variant %d: stale reads %ld of %d (blocks affected %ld)  err=%s\n", variant, tot, blocks * rounds, blk, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
