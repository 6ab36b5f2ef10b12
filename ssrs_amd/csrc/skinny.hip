// Tall-skinny products around the rank-r LoRA / adapter-scalar paths. All HBM-bound: each reads the
// big bf16 token matrix exactly once (algorithmic bytes = M*K*2) and keeps the rank-R side in
// registers.
//
//   skinny_nt : out[m, j]  = sum_k X[m,k] * Wt[g][j,k]   X bf16 [M,K], Wt fp32 [G][R,K] -> fp32 [M,R]  (MFMA)
//               (LoRA down-projection  x @ A^T,  loralib/layers.py:157-158,253,261; and  g @ B  in bwd)
//   skinny_tn : out[c, j] += sum_m Y[m,c] * U[m,j]        Y bf16 [M,C], U fp32 [M,R]   -> fp32 [C,R] or [R,C]
//               (LoRA weight grads  dB = g^T u,  dA = da^T x)
#include "common.h"

// out[m, j] = scale * sum_k X[m,k] * Wt[g(m)][j,k] on the matrix cores: the fp32 rank-R matrix is converted to
// bf16 fragments in registers (rows >= R are zero), a block owns 16 rows of X and its 4 waves split K; the
// accumulator tile is C[j][m], so each lane ends with 4 consecutive j of one row (one float4 store).
__global__ __launch_bounds__(1024) void skinny_nt_kernel(const bf16_t* __restrict__ X, int64_t ldx, const float* __restrict__ Wt,
                                                        int64_t ldw, int64_t w_group_stride, int64_t rows_per_group, int n_groups,
                                                        float* __restrict__ out, int64_t ldo, int64_t M, int K, int R, float scale) {
  __shared__ float red[15][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * 16;
  const int g = n_groups > 1 ? (int)((m0 / rows_per_group) % n_groups) : 0;
  int64_t m = m0 + fr;
  m = m < M ? m : M - 1;
  const bf16_t* xp = X + m * ldx + 8 * fq;
  const float* wp = Wt + (int64_t)g * w_group_stride + (int64_t)fr * ldw + 8 * fq;
  const bool wok = fr < R;
  f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int nsteps = K >> 5;
  auto wfrag = [&](int s) {
    union { bf16x8_t v; unsigned int w[4]; } wf;
    wf.w[0] = wf.w[1] = wf.w[2] = wf.w[3] = 0u;
    if (wok) {
      const float4 a = *reinterpret_cast<const float4*>(wp + 32 * s), b = *reinterpret_cast<const float4*>(wp + 32 * s + 4);
      wf.w[0] = pack_bf16x2(a.x, a.y); wf.w[1] = pack_bf16x2(a.z, a.w); wf.w[2] = pack_bf16x2(b.x, b.y); wf.w[3] = pack_bf16x2(b.z, b.w);
    }
    return wf.v;
  };
  // The block's 4..16 waves split K so that each has at most eight k-steps, all in flight at once (a step past the end
  // re-reads the last one and is not multiplied): one memory round trip per launch for K <= 4096.
  // A wave takes PAIRS of adjacent k-steps (64 + 64 B = one 128-B line of each of its 16 rows, requested back to back).
  for (int s = 0; s < nsteps; s += 8 * nw) {
    bf16x8_t xv[8], wv[8];
    int si[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) si[i] = s + 2 * (wave + nw * (i >> 1)) + (i & 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) xv[i] = *reinterpret_cast<const bf16x8_t*>(xp + 32 * min(si[i], nsteps - 1));
#pragma unroll
    for (int i = 0; i < 8; ++i) wv[i] = wfrag(min(si[i], nsteps - 1));
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (si[i] < nsteps) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[i], xv[i], acc, 0, 0, 0);
  }
  if (wave > 0) {
    red[wave - 1][lane][0] = acc[0]; red[wave - 1][lane][1] = acc[1]; red[wave - 1][lane][2] = acc[2]; red[wave - 1][lane][3] = acc[3];
  }
  __syncthreads();
  if (wave == 0 && m0 + fr < M) {
    for (int w = 0; w < nw - 1; ++w) { acc[0] += red[w][lane][0]; acc[1] += red[w][lane][1]; acc[2] += red[w][lane][2]; acc[3] += red[w][lane][3]; }
    float* op = out + (m0 + fr) * ldo + 4 * fq;      // j = 4*fq + r
    if (4 * fq + 3 < R) *reinterpret_cast<float4*>(op) = make_float4(acc[0] * scale, acc[1] * scale, acc[2] * scale, acc[3] * scale);
    else
      for (int r = 0; r < 4; ++r) if (4 * fq + r < R) op[r] = acc[r] * scale;
  }
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
                                                        float* __restrict__ partial, int64_t group_rows, int64_t out_group_stride) {
  // blockIdx.z = group: rows [z*group_rows, (z+1)*group_rows) of Y / U reduce into out + z*out_group_stride
  __shared__ float red[64][8 * R + 1];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c0 = (blockIdx.x * 64 + lane) * 8;
  const bool active = c0 < C;
  Y += (int64_t)blockIdx.z * group_rows * ldy;
  U += (int64_t)blockIdx.z * group_rows * ldu;
  out += (int64_t)blockIdx.z * out_group_stride;
  if (partial) partial += (int64_t)blockIdx.z * gridDim.y * C * R;
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

// out[g][c, j] (or out[g][j, c]) += scale * sum_s partial[g][s][c][j]; four independent partial sums per thread
__global__ __launch_bounds__(256) void skinny_tn_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int C, int R,
                                                               int splits, int transposed, float scale, int64_t out_group_stride) {
  const int64_t n = (int64_t)C * R;
  partial += (int64_t)blockIdx.y * splits * n;
  out += (int64_t)blockIdx.y * out_group_stride;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int s = 0;
    for (; s + 3 < splits; s += 4) {
      a0 += partial[(int64_t)s * n + i]; a1 += partial[(int64_t)(s + 1) * n + i];
      a2 += partial[(int64_t)(s + 2) * n + i]; a3 += partial[(int64_t)(s + 3) * n + i];
    }
    for (; s < splits; ++s) a0 += partial[(int64_t)s * n + i];
    const int c = (int)(i / R), j = (int)(i - (int64_t)c * R);
    float* o = transposed ? out + (int64_t)j * C + c : out + i;
    *o += (a0 + a1 + a2 + a3) * scale;
  }
}

// MFMA version of the two-stage path: out[c][j] = sum_m y[m][c] * u[m][j] as (y^T).(u) with the rows m as the
// contraction index. Each wave owns 64 columns of y; per 32-row step it copies its [32][64] bf16 slab into a
// wave-private LDS image (row-major, XOR-swizzled) and takes the TRANSPOSED A fragments from it with
// ds_read_b64_tr_b16; the B operand is u converted on the fly, split hi + lo in bf16 (two MFMAs into one
// accumulator), so the fp32 factor keeps ~16 mantissa bits. No barrier: LDS is never shared between waves.
// k-slot j of lane group fq <-> row 16*(j>>2) + 4*fq + (j&3) of the step (same permutation on both operands).
template <int R>
__global__ __launch_bounds__(256) void skinny_tn_mfma_kernel(const bf16_t* __restrict__ Y, int64_t ldy,
                                                             const float* __restrict__ U, int64_t ldu, int64_t M, int C,
                                                             int64_t rows_per_block, float* __restrict__ partial, int64_t group_rows) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4][2][32 * 128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  Y += (int64_t)blockIdx.z * group_rows * ldy;
  U += (int64_t)blockIdx.z * group_rows * ldu;
  partial += ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * C * R;
  const int cw = (blockIdx.x * 4 + wave) * 64;                   // first column of this wave
  if (cw >= C) return;
  const int64_t r_begin = (int64_t)blockIdx.y * rows_per_block;
  int64_t r_end = r_begin + rows_per_block;
  if (r_end > M) r_end = M;
  if (r_begin >= r_end) return;
  // staging: lane copies 16-B chunk (lane & 7) of rows (lane >> 3) + 8*i, i < 4
  const int srow = lane >> 3, sch = lane & 7;
  const int scol = min(cw + sch * 8, C - 8);                      // chunks beyond C re-read the last one (never stored)
  const int ucol = fr < R ? fr : 0;
  f32x4_t acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  u32x4_t yv[4];
  float uv[8];
  auto issue = [&](int64_t m0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int64_t m = m0 + srow + 8 * i;
      m = m < r_end ? m : r_end - 1;
      yv[i] = *reinterpret_cast<const u32x4_t*>(Y + m * ldy + scol);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t m = m0 + 16 * (j >> 2) + 4 * fq + (j & 3);
      uv[j] = (m < r_end && fr < R) ? U[(m < r_end ? m : r_end - 1) * ldu + ucol] : 0.f;
    }
  };
  issue(r_begin);
  int buf = 0;
  for (int64_t m0 = r_begin; m0 < r_end; m0 += 32, buf ^= 1) {
    unsigned char* img = lds[wave][buf];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = srow + 8 * i;
      *reinterpret_cast<u32x4_t*>(img + row * 128 + ((sch ^ (row & 7)) << 4)) = yv[i];
    }
    union { bf16x8_t v; unsigned int w[4]; } uh, ul;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uh.w[i] = pack_bf16x2(uv[2 * i], uv[2 * i + 1]);
      ul.w[i] = pack_bf16x2(uv[2 * i] - bf16_lo(uh.w[i]), uv[2 * i + 1] - bf16_hi(uh.w[i]));
    }
    if (m0 + 32 < r_end) issue(m0 + 32);
    const int trow = 4 * fq + (fr >> 2);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned char* a = img + trow * 128 + (((2 * t + ((fr >> 1) & 1)) ^ (trow & 7)) << 4) + ((fr & 1) << 3);
      union { bf16x8_t v; s16x4_t h[2]; } yf;
      yf.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)a);
      yf.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(a + 16 * 128));
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf.v, uh.v, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf.v, ul.v, acc[t], 0, 0, 0);
    }
  }
  // acc[t][r] = out[cw + 16 t + 4 fq + r][fr]
  if (fr < R) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = cw + 16 * t + 4 * fq + r;
        if (c < C) partial[(int64_t)c * R + fr] = acc[t][r];
      }
  }
}

static void skinny_tn_geometry(int64_t M, int64_t C, int64_t* strips, int64_t* splits, int64_t* rpb) {
  *strips = ceil_div64(C, 256);                       // four waves x 64 columns (MFMA path); the scalar path uses strips of 512
  int64_t r = ceil_div64(M, 32);                      // the second stage sums at most 32 partials per output
  if (r < 64) r = 64;
  r = ceil_div64(r, 32) * 32;
  *rpb = r;
  *splits = ceil_div64(M, r);
}
// floats of scratch that make mfnet_skinny_tn deterministic and atomic-free (two-stage reduction)
extern "C" int64_t mfnet_skinny_tn_workspace_floats(int64_t M, int64_t C, int R, int n_groups) {
  if (M <= 0 || C <= 0 || R <= 0 || n_groups <= 0) return 0;
  int64_t strips, splits, rpb;
  skinny_tn_geometry(M, C, &strips, &splits, &rpb);
  return n_groups * splits * C * R;
}

extern "C" int mfnet_skinny_nt(const void* X, int64_t ldx, const float* Wt, int64_t ldw, int64_t w_group_stride,
                               int64_t rows_per_group, int n_groups, float* out, int64_t ldo, int64_t M, int64_t K, int R,
                               float scale, hipStream_t stream) {
  MF_CHECK_ARG(X && Wt && out, "skinny_nt: null pointer");
  MF_CHECK_ARG(K > 0 && K % 32 == 0 && ldx % 8 == 0 && ldw % 4 == 0 && ldx >= K && ldw >= K, "skinny_nt: K=%lld must be a multiple of 32 (ld multiples of 8 / 4)", (long long)K);
  MF_CHECK_ARG(R >= 1 && R <= 16, "skinny_nt: R=%d must be in [1, 16]", R);
  MF_CHECK_ARG(ldo >= R && ldo % 4 == 0, "skinny_nt: ldo must be >= R and a multiple of 4");
  MF_CHECK_ARG(n_groups >= 1 && (n_groups == 1 || (rows_per_group > 0 && rows_per_group % 16 == 0)), "skinny_nt: rows_per_group must be a multiple of 16");
  MF_CHECK_ARG(K < (1ll << 31), "skinny_nt: K too large");
  if (M <= 0) return MFNET_OK;
  int nw = (int)ceil_div64(K / 32, 8);
  nw = nw < 4 ? 4 : (nw > 16 ? 16 : nw);
  hipLaunchKernelGGL(skinny_nt_kernel, dim3((unsigned)ceil_div64(M, 16)), dim3(64 * nw), 0, stream, (const bf16_t*)X, ldx, Wt, ldw, w_group_stride,
                     rows_per_group, n_groups, out, ldo, M, (int)K, R, scale);
  MF_LAUNCH_CHECK("skinny_nt");
  return MFNET_OK;
}

// out must be zero-initialised (or hold the value to accumulate onto) by the caller. With a workspace of
// mfnet_skinny_tn_workspace_floats(M, C, R) floats the reduction over row splits is two-stage (plain stores +
// one small reduce kernel: deterministic); without it the splits meet in float atomics on `out`.
extern "C" int mfnet_skinny_tn(const void* Y, int64_t ldy, const float* U, int64_t ldu, float* out,
                               int64_t M, int64_t C, int R, int transposed, float scale, int n_groups,
                               int64_t out_group_stride, float* workspace, int64_t workspace_floats, hipStream_t stream) {
  // M = rows PER GROUP; group z reads rows [z*M, (z+1)*M) and accumulates into out + z*out_group_stride
  MF_CHECK_ARG(Y && U && (out || workspace), "skinny_tn: null pointer (out may be NULL only with a workspace: partial tiles only)");
  MF_CHECK_ARG(C > 0 && C % 8 == 0 && ldy % 8 == 0 && ldy >= C, "skinny_tn: bad C/ldy");
  MF_CHECK_ARG(R == 4 || R == 8 || R == 16, "skinny_tn: R=%d must be 4, 8 or 16", R);
  MF_CHECK_ARG(ldu >= R && ldu % 4 == 0, "skinny_tn: bad ldu");
  MF_CHECK_ARG(n_groups >= 1 && n_groups <= 64 && (n_groups == 1 || out_group_stride >= C * R), "skinny_tn: bad group args");
  if (M <= 0) return MFNET_OK;
  int64_t strips, splits, rpb;
  skinny_tn_geometry(M, C, &strips, &splits, &rpb);
  float* partial = nullptr;
  if (workspace) {
    MF_CHECK_ARG(workspace_floats >= n_groups * splits * C * R, "skinny_tn: workspace too small (%lld < %lld floats)", (long long)workspace_floats, (long long)(n_groups * splits * C * R));
    partial = workspace;
  }
  dim3 block(256);
  if (partial) {
    dim3 grid((unsigned)strips, (unsigned)splits, (unsigned)n_groups);
#define SKTNM(RR) hipLaunchKernelGGL((skinny_tn_mfma_kernel<RR>), grid, block, 0, stream, (const bf16_t*)Y, ldy, U, ldu, M, (int)C, rpb, partial, M)
    if (R == 4) SKTNM(4); else if (R == 8) SKTNM(8); else SKTNM(16);
#undef SKTNM
  } else {
    dim3 grid((unsigned)ceil_div64(C, 512), (unsigned)splits, (unsigned)n_groups);
#define SKTN(RR) hipLaunchKernelGGL((skinny_tn_kernel<RR>), grid, block, 0, stream, (const bf16_t*)Y, ldy, U, ldu, out, M, (int)C, rpb, transposed, scale, partial, M, out_group_stride)
    if (R == 4) SKTN(4); else if (R == 8) SKTN(8); else SKTN(16);
#undef SKTN
  }
  if (partial && out) {
    int64_t g = ceil_div64(C * R, 256);
    hipLaunchKernelGGL(skinny_tn_reduce_kernel, dim3((unsigned)g, (unsigned)n_groups), dim3(256), 0, stream, partial, out, (int)C, R, (int)splits, transposed, scale, out_group_stride);
  }
  MF_LAUNCH_CHECK("skinny_tn");
  return MFNET_OK;
}

// =====================================================================================================
// Weight-gradient GEMM  out[n][k] = sum_m dy[m][n] * x[m][k]  (both operands bf16 row-major with the
// contraction index m leading - the layout every activation already has), fp32 out. Each WAVE owns a 64 x 64
// output tile and keeps wave-private row-major LDS slabs of its dy and x columns ([32 rows][64] each, XOR-
// swizzled): BOTH MFMA operands are "transposed" reads (ds_read_b64_tr_b16), so no transposed copy of an
// activation is ever written to HBM (the first version transposed both operands first: 70 launches and
// 0.8 ms per step). LDS operations of one wave execute in order, so the slab is single-buffered and the kernel
// has no barrier; the next step's global loads are in flight while the current one multiplies.
// The row range is cut into `splits`; split s stores its partial tile into out + s*N*ldo (mfnet_reduce_splits
// sums them), exactly like the split-K mode of mfnet_gemm_bf16.
// CONV: X is a channels-last map [B*cH*cW, cC] and the k index runs over (tap, channel) of its 3x3 / pad 1 neighbourhood
// (k = tap*cC + c): the weight gradient of a 3x3 convolution without the im2col tensor - the X rows of a k block are the map
// rows of the pixels shifted by the block's tap, zero where that leaves the image.
// (bx, by, bz) of nz: the workgroup's place in the problem's (k tiles, n tiles, splits * groups) grid - blockIdx of the
// one-problem kernel, a slice of a 1-D grid in the batch kernel. lds_raw: 40 KB.
template <bool CONV>
__device__ __forceinline__ void tn_body(unsigned char* lds_raw, const bf16_t* __restrict__ DY, int64_t ldy, const bf16_t* __restrict__ X, int64_t ldx,
                                        int64_t M, int N, int K, int64_t rows_per_split, float* __restrict__ out, int64_t ldo,
                                        int splits, float* __restrict__ bias_part, int cH, int cW, int cC, int bx, int by, int bz, int nz) {
  // staging: [wave][dy | x image][32 rows x 128 B]; after the row loop the same bytes carry the 4 -> 1 reduction of the waves'
  // accumulators (two 20-KB slots: 16 weight tiles + the bias tile of one wave)
  unsigned char (*lds)[2][32 * 128] = reinterpret_cast<unsigned char (*)[2][32 * 128]>(lds_raw);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = by * 64;
  const int k0 = bx * 64;
  // bz = split * n_groups + group: group g owns rows [g*M, (g+1)*M) of both operands; partial tile (split, group)
  const int n_groups = nz / splits;
  const int sp = bz / n_groups, gz = bz - sp * n_groups;
  DY += (int64_t)gz * M * ldy;
  X += (int64_t)gz * M * ldx;
  // the workgroup owns one 64 x 64 output tile over the split's rows; its four waves take a quarter of the rows each and their
  // accumulators are summed through LDS in a fixed order - a quarter of the partial tiles (and of their traffic in the kernel
  // that sums them) for the same number of waves in flight as one wave per (tile, split)
  const int64_t s_begin = (int64_t)sp * rows_per_split;
  const int64_t s_end = s_begin + rows_per_split < M ? s_begin + rows_per_split : M;
  const int64_t rpw = (rows_per_split + 127) / 128 * 32;
  int64_t r_begin = s_begin + wave * rpw;
  if (r_begin > s_end) r_begin = s_end;
  const int64_t r_end = r_begin + rpw < s_end ? r_begin + rpw : s_end;
  out += (int64_t)bz * N * ldo;
  // bias gradient = column sums of dy over the same rows: the wave that owns k-tile 0 multiplies its dy fragments by a
  // fragment of ones as well (4 more MFMAs per step) and writes the partial row [N] of this (split, group)
  const bool want_b = bias_part != nullptr && k0 == 0;
  f32x4_t accb[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) accb[a] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;
  unsigned char* img_a = lds[wave][0];
  unsigned char* img_b = lds[wave][1];
  const int srow = lane >> 3, sch = lane & 7;
  const int acol = min(n0 + sch * 8, N - 8), bcol = min(k0 + sch * 8, K - 8);   // chunks past the edge re-read the last one
  f32x4_t acc[4][4];      // [n tile][k tile]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  u32x4_t av[4], bv[4];
  int py[4], px[4], dyy = 0, dxx = 0, xcol = bcol;                  // CONV: image coordinates of the lane's rows, this k block's tap
  if (CONV) {
    const int tap = k0 / cC, t3 = tap / 3;
    dyy = t3 - 1; dxx = tap - 3 * t3 - 1;
    xcol = k0 - tap * cC + sch * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pix = (int)((r_begin + srow + 8 * i) % ((int64_t)cH * cW));
      py[i] = pix / cW; px[i] = pix - py[i] * cW;
    }
  }
  const int adv_y = CONV ? 32 / cW : 0, adv_x = CONV ? 32 % cW : 0;
  auto issue = [&](int64_t m0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + srow + 8 * i;
      const int64_t mc = m < r_end ? m : r_end - 1;
      av[i] = *reinterpret_cast<const u32x4_t*>(DY + mc * ldy + acol);
      if (CONV) {
        const bool ok = m < r_end && (unsigned)(py[i] + dyy) < (unsigned)cH && (unsigned)(px[i] + dxx) < (unsigned)cW;
        bv[i] = *reinterpret_cast<const u32x4_t*>(X + (ok ? m + dyy * cW + dxx : mc) * ldx + xcol);
        if (!ok) bv[i] = (u32x4_t){0u, 0u, 0u, 0u};
        px[i] += adv_x; py[i] += adv_y;                            // the next step's rows are 32 pixels on
        if (px[i] >= cW) { px[i] -= cW; ++py[i]; }
        while (py[i] >= cH) py[i] -= cH;
      } else {
        bv[i] = *reinterpret_cast<const u32x4_t*>(X + mc * ldx + bcol);
      }
      if (m >= r_end) av[i] = (u32x4_t){0u, 0u, 0u, 0u};          // rows past the split contribute nothing
    }
  };
  if (r_begin < r_end) issue(r_begin);
  for (int64_t m0 = r_begin; m0 < r_end; m0 += 32) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = srow + 8 * i;
      const int off = row * 128 + ((sch ^ (row & 7)) << 4);
      *reinterpret_cast<u32x4_t*>(img_a + off) = av[i];
      *reinterpret_cast<u32x4_t*>(img_b + off) = bv[i];
    }
    if (m0 + 32 < r_end) issue(m0 + 32);
    // k-slot j of lane group fq <-> row 16*(j>>2) + 4*fq + (j&3) of the step, on both operands
    const int trow = 4 * fq + (fr >> 2);
    bf16x8_t af[4], bf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int off = trow * 128 + (((2 * t + ((fr >> 1) & 1)) ^ (trow & 7)) << 4) + ((fr & 1) << 3);
      union { bf16x8_t v; s16x4_t h[2]; } a, b;
      a.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(img_a + off));
      a.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(img_a + off + 16 * 128));
      b.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(img_b + off));
      b.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(img_b + off + 16 * 128));
      af[t] = a.v; bf[t] = b.v;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
    if (want_b) {
#pragma unroll
      for (int a = 0; a < 4; ++a) accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], ones, accb[a], 0, 0, 0);
    }
  }
  // ---- waves 2, 3 -> slots 0, 1; waves 0, 1 add; wave 1 -> slot 0; wave 0 adds and stores
  auto put = [&](int slot) {
    float4* d = reinterpret_cast<float4*>(lds_raw + slot * 20 * 1024) + lane;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int b = 0; b < 4; ++b) d[(a * 4 + b) * 64] = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
      d[(16 + a) * 64] = make_float4(accb[a][0], accb[a][1], accb[a][2], accb[a][3]);
    }
  };
  auto take = [&](int slot) {
    const float4* d = reinterpret_cast<const float4*>(lds_raw + slot * 20 * 1024) + lane;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float4 v = d[(a * 4 + b) * 64];
        acc[a][b][0] += v.x; acc[a][b][1] += v.y; acc[a][b][2] += v.z; acc[a][b][3] += v.w;
      }
      const float4 v = d[(16 + a) * 64];
      accb[a][0] += v.x; accb[a][1] += v.y; accb[a][2] += v.z; accb[a][3] += v.w;
    }
  };
  __syncthreads();                       // every wave is done with its staging images
  if (wave >= 2) put(wave - 2);
  __syncthreads();
  if (wave < 2) take(wave);
  __syncthreads();
  if (wave == 1) put(0);
  __syncthreads();
  if (wave != 0) return;
  take(0);
  if (want_b && fr == 0) {
    float* bo = bias_part + (int64_t)bz * N;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 16 * a + 4 * fq + r;
        if (n < N) bo[n] = accb[a][r];
      }
  }
  // acc[a][b][r] = out[n0 + 16 a + 4 fq + r][k0 + 16 b + fr]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 16 * a + 4 * fq + r;
      if (n >= N) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int k = k0 + 16 * b + fr;
        if (k < K) out[(int64_t)n * ldo + k] = acc[a][b][r];
      }
    }
}

template <bool CONV>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const bf16_t* __restrict__ DY, int64_t ldy, const bf16_t* __restrict__ X, int64_t ldx,
                                                      int64_t M, int N, int K, int64_t rows_per_split, float* __restrict__ out, int64_t ldo,
                                                      int splits, float* __restrict__ bias_part, int cH, int cW, int cC) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[2 * 20 * 1024];
  tn_body<CONV>(lds_raw, DY, ldy, X, ldx, M, N, K, rows_per_split, out, ldo, splits, bias_part, cH, cW, cC, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z);
}

// Several independent weight-gradient products in ONE launch. The decoder / FPN / adapter weight gradients of a step are ~35
// (MMAdapter: ~110) products of 1 to a few hundred workgroups each; launched one by one, the small ones are a serial chain of
// 12-15 us latencies on a mostly idle chip (0.57 ms of the ViT-B MMLoRA step). None of them feeds the backward chain - they only
// have to be done before the gradient delivery - so the host collects them and this kernel runs them side by side: workgroup id
// -> problem by a scan of the start offsets (<= TN_BATCH problems, kernel arguments, no device table).
#define TN_BATCH 24
struct TnProb {
  const bf16_t* DY; const bf16_t* X; float* out; float* bias_part;
  int64_t ldy, ldx, M, rps, ldo;
  int N, K, splits, nz, cH, cW, cC, gx, gy, wg_begin;
};
struct TnBatch { int n; TnProb p[TN_BATCH]; };
__global__ __launch_bounds__(256) void gemm_tn_batch_kernel(const TnBatch b) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[2 * 20 * 1024];
  const int id = blockIdx.x;
  int i = 0;
  for (int j = 1; j < b.n; ++j)
    if (id >= b.p[j].wg_begin) i = j;
  const TnProb& q = b.p[i];
  const int local = id - q.wg_begin;
  const int bx = local % q.gx, t = local / q.gx, by = t % q.gy, bz = t / q.gy;
  if (q.cC) tn_body<true>(lds_raw, q.DY, q.ldy, q.X, q.ldx, q.M, q.N, q.K, q.rps, q.out, q.ldo, q.splits, q.bias_part, q.cH, q.cW, q.cC, bx, by, bz, q.nz);
  else tn_body<false>(lds_raw, q.DY, q.ldy, q.X, q.ldx, q.M, q.N, q.K, q.rps, q.out, q.ldo, q.splits, q.bias_part, 0, 0, 0, bx, by, bz, q.nz);
}

extern "C" int64_t mfnet_gemm_tn_splits(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  const int64_t tiles = ceil_div64(N, 64) * ceil_div64(K, 64);   // one 4-wave workgroup per tile and split
  int64_t s = ceil_div64(768, tiles);                  // ~3 workgroups per CU
  const int64_t smax = ceil_div64(M, 1024);            // at least 8 steps of 32 rows per wave (partials cost traffic too)
  if (s > smax) s = smax;
  if (s > 256) s = 256;
  return s < 1 ? 1 : s;
}

static int gemm_tn_check(const void* dy, int64_t ldy, const void* x, int64_t ldx, int64_t M, int64_t N, int64_t K,
                         float* out, int64_t ldo, int splits, int n_groups, int cC);
// desc: 16 int64 per problem - dy, ldy, x, ldx, M (rows per group), N, K, out, ldo, splits, n_groups, bias_part (0: none),
// conv_H, conv_W, conv_C (0: plain product; else x is a channels-last map and K = 9 * conv_C as in mfnet_gemm_tn_conv3x3), 0.
extern "C" int mfnet_gemm_tn_batch(const int64_t* desc, int64_t n, hipStream_t stream) {
  MF_CHECK_ARG(desc || n == 0, "gemm_tn_batch: null descriptor table");
  MF_CHECK_ARG(n >= 0 && n <= 4096, "gemm_tn_batch: bad problem count");
  for (int64_t i = 0; i < n; ++i) {
    const int64_t* d = desc + 16 * i;
    int rc = gemm_tn_check((const void*)d[0], d[1], (const void*)d[2], d[3], d[4], d[5], d[6], (float*)d[7], d[8], (int)d[9], (int)d[10], (int)d[14]);
    if (rc) return rc;
    MF_CHECK_ARG(d[14] == 0 || (d[12] > 0 && d[13] > 0 && d[14] % 64 == 0 && d[6] == 9 * d[14]), "gemm_tn_batch: bad conv geometry of problem %lld", (long long)i);
  }
  for (int64_t i0 = 0; i0 < n; i0 += TN_BATCH) {
    TnBatch b;
    b.n = (int)(n - i0 < TN_BATCH ? n - i0 : TN_BATCH);
    int64_t total = 0;
    for (int j = 0; j < b.n; ++j) {
      const int64_t* d = desc + 16 * (i0 + j);
      TnProb& q = b.p[j];
      q.DY = (const bf16_t*)d[0]; q.ldy = d[1]; q.X = (const bf16_t*)d[2]; q.ldx = d[3]; q.M = d[4]; q.N = (int)d[5]; q.K = (int)d[6];
      q.out = (float*)d[7]; q.ldo = d[8]; q.splits = (int)d[9]; q.nz = (int)(d[9] * d[10]); q.bias_part = (float*)d[11];
      q.cH = (int)d[12]; q.cW = (int)d[13]; q.cC = (int)d[14];
      q.rps = ceil_div64(ceil_div64(q.M, q.splits), 32) * 32;
      q.gx = (int)ceil_div64(q.K, 64); q.gy = (int)ceil_div64(q.N, 64);
      q.wg_begin = (int)total;
      total += (int64_t)q.gx * q.gy * q.nz;
      MF_CHECK_ARG(total < (1ll << 30), "gemm_tn_batch: too many workgroups");
    }
    for (int j = b.n; j < TN_BATCH; ++j) b.p[j] = b.p[0];
    hipLaunchKernelGGL(gemm_tn_batch_kernel, dim3((unsigned)total), dim3(256), 0, stream, b);
    MF_LAUNCH_CHECK("gemm_tn_batch");
  }
  return MFNET_OK;
}

static int gemm_tn_check(const void* dy, int64_t ldy, const void* x, int64_t ldx, int64_t M, int64_t N, int64_t K,
                         float* out, int64_t ldo, int splits, int n_groups, int cC) {
  MF_CHECK_ARG(dy && x && out, "gemm_tn: null pointer");
  MF_CHECK_ARG(n_groups >= 1 && (int64_t)n_groups * splits <= 65535, "gemm_tn: bad group count");
  MF_CHECK_ARG(M > 0 && N >= 8 && K >= 8 && N % 8 == 0 && K % 8 == 0, "gemm_tn: N=%lld and K=%lld must be multiples of 8", (long long)N, (long long)K);
  MF_CHECK_ARG(ldy % 8 == 0 && ldx % 8 == 0 && ldy >= N && ldx >= (cC ? cC : K) && ldo >= K, "gemm_tn: bad leading dimensions");
  MF_CHECK_ARG(((uintptr_t)dy | (uintptr_t)x) % 16 == 0, "gemm_tn: operands must be 16-byte aligned");
  MF_CHECK_ARG(splits >= 1 && splits <= 65535 && N < (1ll << 31) && K < (1ll << 31), "gemm_tn: bad splits / shape");
  return MFNET_OK;
}

static int gemm_tn_launch(const void* dy, int64_t ldy, const void* x, int64_t ldx, int64_t M, int64_t N, int64_t K,
                          float* out, int64_t ldo, int splits, int n_groups, float* bias_part, hipStream_t stream,
                          int cH = 0, int cW = 0, int cC = 0) {
  MF_CHECK_ARG(dy && x && out, "gemm_tn: null pointer");
  MF_CHECK_ARG(n_groups >= 1 && (int64_t)n_groups * splits <= 65535, "gemm_tn: bad group count");
  MF_CHECK_ARG(M > 0 && N >= 8 && K >= 8 && N % 8 == 0 && K % 8 == 0, "gemm_tn: N=%lld and K=%lld must be multiples of 8", (long long)N, (long long)K);
  MF_CHECK_ARG(ldy % 8 == 0 && ldx % 8 == 0 && ldy >= N && ldx >= (cC ? cC : K) && ldo >= K, "gemm_tn: bad leading dimensions");
  MF_CHECK_ARG(((uintptr_t)dy | (uintptr_t)x) % 16 == 0, "gemm_tn: operands must be 16-byte aligned");
  MF_CHECK_ARG(splits >= 1 && splits <= 65535 && N < (1ll << 31) && K < (1ll << 31), "gemm_tn: bad splits / shape");
  int64_t rps = ceil_div64(ceil_div64(M, splits), 32) * 32;
  dim3 grid((unsigned)ceil_div64(K, 64), (unsigned)ceil_div64(N, 64), (unsigned)(splits * n_groups));
  MF_CHECK_ARG(grid.y <= 65535, "gemm_tn: N too large");
  if (cC) hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, dim3(256), 0, stream, (const bf16_t*)dy, ldy, (const bf16_t*)x, ldx, M, (int)N, (int)K, rps, out, ldo, splits, bias_part, cH, cW, cC);
  else hipLaunchKernelGGL(gemm_tn_kernel<false>, grid, dim3(256), 0, stream, (const bf16_t*)dy, ldy, (const bf16_t*)x, ldx, M, (int)N, (int)K, rps, out, ldo, splits, bias_part, 0, 0, 0);
  MF_LAUNCH_CHECK("gemm_tn");
  return MFNET_OK;
}

extern "C" int mfnet_gemm_tn_bf16(const void* dy, int64_t ldy, const void* x, int64_t ldx, int64_t M, int64_t N, int64_t K,
                                  float* out, int64_t ldo, int splits, float* bias_part, hipStream_t stream) {
  return gemm_tn_launch(dy, ldy, x, ldx, M, N, K, out, ldo, splits, 1, bias_part, stream);
}

// Weight gradient of a 3x3 / pad 1 convolution without the column tensor: out[n][tap*C + c] = sum over pixels p of
// dy[p][n] * x[p shifted by tap][c] (zero outside the image); x [B*H*W, C] channels-last bf16, C % 64 == 0. Splits, partial tiles
// and bias_part as mfnet_gemm_tn_bf16 with M = B*H*W, K = 9*C.
extern "C" int mfnet_gemm_tn_conv3x3(const void* dy, int64_t ldy, const void* x, int64_t ldx, int64_t B, int64_t H, int64_t W, int64_t C,
                                     int64_t N, float* out, int64_t ldo, int splits, float* bias_part, hipStream_t stream) {
  MF_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 64 == 0 && H < (1 << 20) && W < (1 << 20) && H * W < (1ll << 31),
               "gemm_tn_conv3x3: bad geometry (channels must be a multiple of 64, got %lld)", (long long)C);
  return gemm_tn_launch(dy, ldy, x, ldx, B * H * W, N, 9 * C, out, ldo, splits, 1, bias_part, stream, (int)H, (int)W, (int)C);
}

// n_groups independent products in one launch: group g takes rows [g*M, (g+1)*M) of dy and x (M = rows PER GROUP) and
// writes its partial tiles to out + (s*n_groups + g)*N*ldo, s < splits - the two modalities' adapter weight gradients.
extern "C" int mfnet_gemm_tn_bf16_grouped(const void* dy, int64_t ldy, const void* x, int64_t ldx, int64_t M, int64_t N, int64_t K,
                                          float* out, int64_t ldo, int splits, int n_groups, float* bias_part, hipStream_t stream) {
  return gemm_tn_launch(dy, ldy, x, ldx, M, N, K, out, ldo, splits, n_groups, bias_part, stream);
}
