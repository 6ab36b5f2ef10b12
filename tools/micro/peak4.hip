// Micro-benchmarks of the per-CU ceilings the GEMM main loop lives under (diagnostic, not part of the library):
//   mfma : 8 waves / CU, each issuing independent mfma_f32_16x16x32_bf16 back to back
//   lds  : 8 waves / CU, each issuing ds_read_b128 back to back (conflict-free, lane-linear)
//   both : the two interleaved (10 reads per 24 MFMAs, the NF = 6 ratio)
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/peak.hip -o tools/micro/peak ; run: tools/micro/peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int MODE, int TH>
__global__ __launch_bounds__(TH, TH / 256) void peak_kernel(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
  const int tid = threadIdx.x;
  constexpr bool LDSM = MODE == 1 || MODE == 2 || MODE == 3, MFM = MODE != 1, BAR = MODE >= 3;
  for (int i = tid; i < 64 * 1024 / 4; i += TH) ((unsigned*)lds)[i] = 0x3f803f80u;
  __syncthreads();
  f32x4_t acc[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t a[4], b[6];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(lds + tid * 16 + i * 8192);
#pragma unroll
  for (int i = 0; i < 6; ++i) b[i] = *reinterpret_cast<const bf16x8_t*>(lds + tid * 16 + i * 8192);
  bf16x8_t r[10];
  const unsigned char* base = lds + (tid & 63) * 16 + (tid >> 6) * 1024;
  const unsigned la = (unsigned)(size_t)(base - lds);
  bf16x8_t r2[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) r2[i] = i < 4 ? a[i] : b[i - 4];
  auto reads = [&](bf16x8_t (&dst)[10], int it) {
    const unsigned addr = la + ((it & 3) << 11);
#pragma unroll
    for (int i = 0; i < 10; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[i]) : "v"(addr), "n"(i * 4096));
  };
  auto mm = [&](const bf16x8_t (&src)[10]) {
#pragma unroll
    for (int f = 0; f < 6; ++f)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[f * 4 + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(src[4 + f], src[m], acc[f * 4 + m], 0, 0, 0);
  };
  for (int it = 0; it < iters; it += 2) {
    if (LDSM) reads(r, it);
    __builtin_amdgcn_sched_barrier(0);
    if (MFM) mm(r2);
    __builtin_amdgcn_sched_barrier(0);
    if (LDSM) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 10; ++i) asm volatile("" ::"v"(r[i]));
      }
    }
    if (BAR) __builtin_amdgcn_s_barrier();
    if (LDSM) reads(r2, it + 1);
    __builtin_amdgcn_sched_barrier(0);
    if (MFM) mm(LDSM ? r : r2);
    __builtin_amdgcn_sched_barrier(0);
    if (LDSM) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 10; ++i) asm volatile("" ::"v"(r2[i]));
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 24; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.f) out[0] = s;
}

template <int MODE, int TH> void run(const char* name, int wgs) {
  float* out; hipMalloc(&out, 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((peak_kernel<MODE, TH>), dim3(wgs), dim3(TH), 0, 0, out, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((peak_kernel<MODE, TH>), dim3(wgs), dim3(TH), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = MODE != 1 ? (double)wgs * (TH / 64) * iters * 24 * 2.0 * 16 * 16 * 32 : 0;
  const double bytes = (MODE >= 1 && MODE <= 3) ? (double)wgs * (TH / 64) * iters * 10 * 1024 : 0;
  printf("%-13s th=%d %4d WGs: %8.3f ms  %8.1f TFLOP/s  LDS read %8.1f TB/s (%.1f B/clk/CU at 2.4 GHz over 256 CUs)\n", name, TH, wgs, ms, flops / ms / 1e9, bytes / ms / 1e9,
         bytes / (ms * 1e-3) / 256 / 2.4e9);
  hipFree(out);
}

int main() {
  for (int wgs : {256}) {
    run<0, 512>("mfma", wgs);
    run<2, 512>("both", wgs);
    run<3, 512>("both+barrier", wgs);
    run<0, 256>("mfma", wgs);
    run<1, 256>("lds", wgs);
    run<2, 256>("both", wgs);
    run<3, 256>("both+barrier", wgs);
    run<4, 256>("mfma+barrier", wgs);
  }
  return 0;
}
