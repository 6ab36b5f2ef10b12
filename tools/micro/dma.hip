// Diagnostic: per-CU rate of global_load_lds (16 B / lane, L2-resident source) alone, of ds_read_b128 alone, and of the two
// together - do the LDS-DMA writes of the GEMM staging take LDS cycles away from the fragment reads?
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/dma.hip -o tools/micro/dma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <bool DMA, bool RD, int NDMA>
__global__ __launch_bounds__(512, 2) void dma_kernel(const unsigned char* src, float* out, int iters) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[128 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 128 * 1024 / 4; i += 512) ((unsigned*)lds)[i] = 0x3f803f80u;
  __syncthreads();
  const unsigned char* g = src + ((size_t)(blockIdx.x & 255) * 64 * 1024) + wave * 8192 + lane * 16;
  const unsigned la = (unsigned)(size_t)(lane * 16 + wave * 1024);
  bf16x8_t r[10];
  for (int it = 0; it < iters; ++it) {
    if (DMA) {
#pragma unroll
      for (int i = 0; i < NDMA; ++i)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + i * 1024 + (it & 1) * 4096 * 0), (lds_ptr_t)(lds + 64 * 1024 + wave * 8192 + i * 1024), 16, 0, 0);
    }
    if (RD) {
      const unsigned addr = la + ((it & 3) << 13);
#pragma unroll
      for (int i = 0; i < 10; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "n"(i * 1024 * 0 + i * 16 * 0 + (i & 1) * 8192 * 0 + i * 512 * 0));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 10; ++i) asm volatile("" ::"v"(r[i]));
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (out && tid == 100000) out[0] = lds[tid];
}

template <bool DMA, bool RD, int NDMA> void run(const char* name, const unsigned char* src) {
  const int iters = 2000, wgs = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((dma_kernel<DMA, RD, NDMA>), dim3(wgs), dim3(512), 0, 0, src, (float*)nullptr, iters);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((dma_kernel<DMA, RD, NDMA>), dim3(wgs), dim3(512), 0, 0, src, (float*)nullptr, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double dma = DMA ? (double)wgs * 8 * iters * NDMA * 1024 : 0, rd = RD ? (double)wgs * 8 * iters * 10 * 1024 : 0;
  printf("%-22s %8.3f ms   DMA %6.1f B/clk/CU (%5.1f TB/s)   ds_read %6.1f B/clk/CU   [2.4 GHz, 256 CUs]\n", name, ms, dma / (ms * 1e-3) / 256 / 2.4e9,
         dma / ms / 1e9, rd / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
  unsigned char* src; (void)hipMalloc(&src, (size_t)256 * 64 * 1024 + 65536);
  (void)hipMemset(src, 1, (size_t)256 * 64 * 1024 + 65536);
  run<true, false, 7>("dma x7", src);
  run<false, true, 7>("ds_read x10", src);
  run<true, true, 7>("dma x7 + ds_read x10", src);
  run<true, true, 4>("dma x4 + ds_read x10", src);
  run<true, true, 2>("dma x2 + ds_read x10", src);
  (void)hipFree(src);
  return 0;
}
