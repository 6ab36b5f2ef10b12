// Tall-skinny products around the rank-r LoRA / adapter-scalar paths. All HBM-bound: each reads the
// big bf16 token matrix exactly once (algorithmic bytes = M*K*2) and keeps the rank-R side in
// registers.
//
//   skinny_nt : out[m, j]  = sum_k X[m,k] * Wt[j,k]      X bf16 [M,K], Wt fp32 [R,K]  -> fp32 [M,R]
//               (LoRA down-projection  x @ A^T,  loralib/layers.py:157-158,253,261; and  g @ B  in bwd)
//   skinny_tn : out[c, j] += sum_m Y[m,c] * U[m,j]        Y bf16 [M,C], U fp32 [M,R]   -> fp32 [C,R] or [R,C]
//               (LoRA weight grads  dB = g^T u,  dA = da^T x)
#include "common.h"

#define SK_ROWS 8

template <int R>
__global__ __launch_bounds__(256) void skinny_nt_kernel(const bf16_t* __restrict__ X, int64_t ldx,
                                                        const float* __restrict__ Wt, int64_t ldw,
                                                        float* __restrict__ out, int64_t ldo, int64_t M, int K,
                                                        float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * SK_ROWS;
  if (row0 >= M) return;
  float acc[SK_ROWS][R];
#pragma unroll
  for (int r = 0; r < SK_ROWS; ++r)
#pragma unroll
    for (int j = 0; j < R; ++j) acc[r][j] = 0.f;
  const int nch = K >> 3;
  for (int c = lane; c < nch; c += 64) {
    float w[R][8];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const float4 a = *reinterpret_cast<const float4*>(Wt + (int64_t)j * ldw + 8 * c);
      const float4 b = *reinterpret_cast<const float4*>(Wt + (int64_t)j * ldw + 8 * c + 4);
      w[j][0] = a.x; w[j][1] = a.y; w[j][2] = a.z; w[j][3] = a.w;
      w[j][4] = b.x; w[j][5] = b.y; w[j][6] = b.z; w[j][7] = b.w;
    }
#pragma unroll
    for (int r = 0; r < SK_ROWS; ++r) {
      int64_t m = row0 + r;
      m = m < M ? m : M - 1;
      const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(X + m * ldx + 8 * c);
      const float x[8] = {bf16_lo(xv.x), bf16_hi(xv.x), bf16_lo(xv.y), bf16_hi(xv.y),
                          bf16_lo(xv.z), bf16_hi(xv.z), bf16_lo(xv.w), bf16_hi(xv.w)};
#pragma unroll
      for (int j = 0; j < R; ++j) {
        float s = acc[r][j];
#pragma unroll
        for (int e = 0; e < 8; ++e) s += x[e] * w[j][e];
        acc[r][j] = s;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < SK_ROWS; ++r)
#pragma unroll
    for (int j = 0; j < R; ++j) acc[r][j] = wave_sum(acc[r][j]);
  // lane l < SK_ROWS*R writes element (r = l / R, j = l % R)
#pragma unroll
  for (int r = 0; r < SK_ROWS; ++r)
#pragma unroll
    for (int j = 0; j < R; ++j)
      if (lane == ((r * R + j) & 63) && row0 + r < M) out[(row0 + r) * ldo + j] = acc[r][j] * scale;
}

// grid: (col strips of 512, row splits); block = 4 waves sharing the strip, each wave a row slice.
// The four waves' partial sums are combined in LDS and only wave 0 issues the global float adds, and
// the row splits are sized for ~256 workgroups: the output has only C*R addresses, so every extra split
// is one more atomic per address (the first version had 4 per split and ~700 per address).
template <int R>
__global__ __launch_bounds__(256) void skinny_tn_kernel(const bf16_t* __restrict__ Y, int64_t ldy,
                                                        const float* __restrict__ U, int64_t ldu,
                                                        float* __restrict__ out, int64_t M, int C,
                                                        int64_t rows_per_block, int transposed, float scale,
                                                        float* __restrict__ partial) {
  __shared__ float red[64][8 * R + 1];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c0 = (blockIdx.x * 64 + lane) * 8;
  const bool active = c0 < C;
  const int64_t r_begin = (int64_t)blockIdx.y * rows_per_block;
  int64_t r_end = r_begin + rows_per_block;
  if (r_end > M) r_end = M;
  float acc[8][R];
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int j = 0; j < R; ++j) acc[e][j] = 0.f;
  if (active) {
#pragma unroll 4
    for (int64_t m = r_begin + wave; m < r_end; m += 4) {
      const u32x4_t yv = *reinterpret_cast<const u32x4_t*>(Y + m * ldy + c0);
      const float y[8] = {bf16_lo(yv.x), bf16_hi(yv.x), bf16_lo(yv.y), bf16_hi(yv.y),
                          bf16_lo(yv.z), bf16_hi(yv.z), bf16_lo(yv.w), bf16_hi(yv.w)};
      float u[R];
#pragma unroll
      for (int j = 0; j < R; j += 4) {
        const float4 t = *reinterpret_cast<const float4*>(U + m * ldu + j);
        u[j] = t.x; u[j + 1] = t.y; u[j + 2] = t.z; u[j + 3] = t.w;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int j = 0; j < R; ++j) acc[e][j] += y[e] * u[j];
    }
  }
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int j = 0; j < R; ++j) red[lane][e * R + j] = acc[e][j];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int j = 0; j < R; ++j) acc[e][j] += red[lane][e * R + j];
    }
    __syncthreads();
  }
  if (wave == 0 && active && partial) {      // two-stage mode: plain coalesced stores, reduced by skinny_tn_reduce_kernel
    float4* dst = reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.y * C + c0) * R);
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int j = 0; j < R; j += 4) dst[(e * R + j) >> 2] = make_float4(acc[e][j], acc[e][j + 1], acc[e][j + 2], acc[e][j + 3]);
    return;
  }
  if (wave == 0 && active) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const float v = acc[e][j] * scale;
        if (transposed) atomicAdd(out + (int64_t)j * C + c0 + e, v);
        else atomicAdd(out + (int64_t)(c0 + e) * R + j, v);
      }
  }
}

// out[c, j] (or out[j, c]) += scale * sum_s partial[s][c][j]
__global__ __launch_bounds__(256) void skinny_tn_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int C, int R,
                                                               int splits, int transposed, float scale) {
  const int64_t n = (int64_t)C * R;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float a = 0.f;
    for (int s = 0; s < splits; ++s) a += partial[(int64_t)s * n + i];
    const int c = (int)(i / R), j = (int)(i - (int64_t)c * R);
    float* o = transposed ? out + (int64_t)j * C + c : out + i;
    *o += a * scale;
  }
}

static void skinny_tn_geometry(int64_t M, int64_t C, int64_t* strips, int64_t* splits, int64_t* rpb) {
  *strips = ceil_div64(C, 512);
  int64_t sp = ceil_div64(256, *strips);              // aim for ~256 workgroups (one per CU)
  int64_t r = ceil_div64(M, sp);
  if (r < 64) r = 64;
  r = ceil_div64(r, 4) * 4;
  *rpb = r;
  *splits = ceil_div64(M, r);
}
// floats of scratch that make mfnet_skinny_tn deterministic and atomic-free (two-stage reduction)
extern "C" int64_t mfnet_skinny_tn_workspace_floats(int64_t M, int64_t C, int R) {
  if (M <= 0 || C <= 0 || R <= 0) return 0;
  int64_t strips, splits, rpb;
  skinny_tn_geometry(M, C, &strips, &splits, &rpb);
  return splits * C * R;
}

extern "C" int mfnet_skinny_nt(const void* X, int64_t ldx, const float* Wt, int64_t ldw, float* out,
                               int64_t ldo, int64_t M, int64_t K, int R, float scale, hipStream_t stream) {
  MF_CHECK_ARG(X && Wt && out, "skinny_nt: null pointer");
  MF_CHECK_ARG(K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldw % 4 == 0 && ldx >= K && ldw >= K, "skinny_nt: bad K/ld");
  MF_CHECK_ARG(R == 4 || R == 8 || R == 16, "skinny_nt: R=%d must be 4, 8 or 16", R);
  MF_CHECK_ARG(ldo >= R, "skinny_nt: bad ldo");
  if (M <= 0) return MFNET_OK;
  dim3 grid((unsigned)ceil_div64(M, 4 * SK_ROWS)), block(256);
#define SKNT(RR) hipLaunchKernelGGL((skinny_nt_kernel<RR>), grid, block, 0, stream, (const bf16_t*)X, ldx, Wt, ldw, out, ldo, M, (int)K, scale)
  if (R == 4) SKNT(4); else if (R == 8) SKNT(8); else SKNT(16);
#undef SKNT
  MF_LAUNCH_CHECK("skinny_nt");
  return MFNET_OK;
}

// out must be zero-initialised (or hold the value to accumulate onto) by the caller. With a workspace of
// mfnet_skinny_tn_workspace_floats(M, C, R) floats the reduction over row splits is two-stage (plain stores +
// one small reduce kernel: deterministic); without it the splits meet in float atomics on `out`.
extern "C" int mfnet_skinny_tn(const void* Y, int64_t ldy, const float* U, int64_t ldu, float* out,
                               int64_t M, int64_t C, int R, int transposed, float scale, float* workspace,
                               int64_t workspace_floats, hipStream_t stream) {
  MF_CHECK_ARG(Y && U && out, "skinny_tn: null pointer");
  MF_CHECK_ARG(C > 0 && C % 8 == 0 && ldy % 8 == 0 && ldy >= C, "skinny_tn: bad C/ldy");
  MF_CHECK_ARG(R == 4 || R == 8 || R == 16, "skinny_tn: R=%d must be 4, 8 or 16", R);
  MF_CHECK_ARG(ldu >= R && ldu % 4 == 0, "skinny_tn: bad ldu");
  if (M <= 0) return MFNET_OK;
  int64_t strips, splits, rpb;
  skinny_tn_geometry(M, C, &strips, &splits, &rpb);
  float* partial = nullptr;
  if (workspace) {
    MF_CHECK_ARG(workspace_floats >= splits * C * R, "skinny_tn: workspace too small (%lld < %lld floats)", (long long)workspace_floats, (long long)(splits * C * R));
    partial = workspace;
  }
  dim3 grid((unsigned)strips, (unsigned)splits), block(256);
#define SKTN(RR) hipLaunchKernelGGL((skinny_tn_kernel<RR>), grid, block, 0, stream, (const bf16_t*)Y, ldy, U, ldu, out, M, (int)C, rpb, transposed, scale, partial)
  if (R == 4) SKTN(4); else if (R == 8) SKTN(8); else SKTN(16);
#undef SKTN
  if (partial) {
    int64_t g = ceil_div64(C * R, 256);
    hipLaunchKernelGGL(skinny_tn_reduce_kernel, dim3((unsigned)g), dim3(256), 0, stream, partial, out, (int)C, R, (int)splits, transposed, scale);
  }
  MF_LAUNCH_CHECK("skinny_tn");
  return MFNET_OK;
}
