// Small HBM-bound data-movement kernels of the encoder path: dtype casts, patch-embed im2col,
// 3x3 im2col / col2im for the neck conv, the LoRA cross-modal mix (loralib/layers.py:159-160).
#include "common.h"
#include <algorithm>
#include <vector>
#include <string.h>

// ------------------------------------------------------------------------------------------- cast
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(x)[2 * i];
    const float4 b = reinterpret_cast<const float4*>(x)[2 * i + 1];
    u32x4_t w;
    w.x = pack_bf16x2(a.x, a.y); w.y = pack_bf16x2(a.z, a.w); w.z = pack_bf16x2(b.x, b.y); w.w = pack_bf16x2(b.z, b.w);
    reinterpret_cast<u32x4_t*>(y)[i] = w;
  }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const u32x4_t w = reinterpret_cast<const u32x4_t*>(x)[i];
    reinterpret_cast<float4*>(y)[2 * i] = make_float4(bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y));
    reinterpret_cast<float4*>(y)[2 * i + 1] = make_float4(bf16_lo(w.z), bf16_hi(w.z), bf16_lo(w.w), bf16_hi(w.w));
  }
}
static inline unsigned grid_for(int64_t work_items) {
  int64_t g = ceil_div64(work_items, 256);
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (unsigned)g;
}
extern "C" int mfnet_cast(const void* x, int x_dtype, void* y, int y_dtype, int64_t n, hipStream_t stream) {
  MF_CHECK_ARG(x && y, "cast: null pointer");
  MF_CHECK_ARG(n >= 0 && n % 8 == 0, "cast: n=%lld must be a multiple of 8", (long long)n);
  if (n == 0) return MFNET_OK;
  if (x_dtype == MFNET_F32 && y_dtype == MFNET_BF16)
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const float*)x, (bf16_t*)y, n / 8);
  else if (x_dtype == MFNET_BF16 && y_dtype == MFNET_F32)
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16_t*)x, (float*)y, n / 8);
  else { mfnet_set_error("cast: unsupported dtype pair"); return MFNET_ERR_ARG; }
  MF_LAUNCH_CHECK("cast");
  return MFNET_OK;
}

// ------------------------------------------------------------------------- patch-embed im2col
// rgb [B,3,H,W] fp32 and dsm [B,H,W] fp32 (replicated to 3 channels, unetformer_mmsam.py:535) ->
// cols [2*B*T, 3*P*P] bf16, row = (modality, b, ph, pw), k = c*P*P + py*P + px  (conv weight order,
// image_encoder.py:189-197). One thread = 8 consecutive px.
// dsm_ch = 1: the second modality is one plane per image, replicated over the 3 input channels; 3: a real 3-channel image.
__global__ __launch_bounds__(256) void im2col_patch_kernel(const float* __restrict__ rgb, const float* __restrict__ dsm,
                                                           bf16_t* __restrict__ cols, int B, int H, int W, int P, int dsm_ch) {
  const int Sh = H / P, Sw = W / P;
  const int K = 3 * P * P, K8 = K >> 3, P8 = P >> 3;
  const int64_t total = (int64_t)2 * B * Sh * Sw * K8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k8 = (int)(i % K8);
    const int64_t row = i / K8;
    const int px8 = k8 % P8, py = (k8 / P8) % P, c = k8 / (P8 * P);
    const int pw = (int)(row % Sw), ph = (int)((row / Sw) % Sh);
    const int b = (int)((row / ((int64_t)Sw * Sh)) % B), mod = (int)(row / ((int64_t)Sw * Sh * B));
    const int y = ph * P + py, x = pw * P + px8 * 8;
    const float* src = mod == 0 ? rgb + (((int64_t)b * 3 + c) * H + y) * W + x
                                : dsm + (((int64_t)b * dsm_ch + (dsm_ch == 3 ? c : 0)) * H + y) * W + x;
    const float4 a = *reinterpret_cast<const float4*>(src), bb = *reinterpret_cast<const float4*>(src + 4);
    u32x4_t w;
    w.x = pack_bf16x2(a.x, a.y); w.y = pack_bf16x2(a.z, a.w); w.z = pack_bf16x2(bb.x, bb.y); w.w = pack_bf16x2(bb.z, bb.w);
    *reinterpret_cast<u32x4_t*>(cols + row * K + (int64_t)k8 * 8) = w;
  }
}
extern "C" int mfnet_im2col_patch(const float* rgb, const float* dsm, int dsm_channels, void* cols, int64_t B, int64_t H, int64_t W,
                                  int64_t P, hipStream_t stream) {
  MF_CHECK_ARG(rgb && dsm && cols, "im2col_patch: null pointer");
  MF_CHECK_ARG(dsm_channels == 1 || dsm_channels == 3, "im2col_patch: the second modality has 1 or 3 channels, not %d", dsm_channels);
  MF_CHECK_ARG(P > 0 && P % 8 == 0 && H % P == 0 && W % P == 0 && W % 4 == 0, "im2col_patch: bad geometry H=%lld W=%lld P=%lld", (long long)H, (long long)W, (long long)P);
  if (B <= 0) return MFNET_OK;
  const int64_t total = 2 * B * (H / P) * (W / P) * (3 * P * P / 8);
  hipLaunchKernelGGL(im2col_patch_kernel, dim3(grid_for(total)), dim3(256), 0, stream, rgb, dsm, (bf16_t*)cols, (int)B, (int)H, (int)W, (int)P, dsm_channels);
  MF_LAUNCH_CHECK("im2col_patch");
  return MFNET_OK;
}

// ------------------------------------------------------------------------- 3x3 im2col / col2im (NHWC)
// x [G,H,W,C] bf16 -> cols [G*H*W, 9*C] bf16, k = tap*C + c, tap = ky*3+kx, zero padding 1.
__global__ __launch_bounds__(256) void im2col3x3_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ cols,
                                                        int G, int H, int W, int C) {
  const int C8 = C >> 3;
  const int64_t total = (int64_t)G * H * W * 9 * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int tap = (int)((i / C8) % 9);
    const int64_t pix = i / (9 * C8);
    const int xx = (int)(pix % W), yy = (int)((pix / W) % H);
    const int64_t g = pix / ((int64_t)W * H);
    const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
    u32x4_t v = (u32x4_t){0u, 0u, 0u, 0u};
    if (sy >= 0 && sy < H && sx >= 0 && sx < W)
      v = *reinterpret_cast<const u32x4_t*>(x + ((g * H + sy) * W + sx) * C + c8 * 8);
    *reinterpret_cast<u32x4_t*>(cols + pix * 9 * C + (int64_t)tap * C + c8 * 8) = v;
  }
}
// dcols [G*H*W, 9*C] bf16 -> dx [G,H,W,C] fp32 : dx[p] = sum_tap dcols[p - off(tap)][tap]
__global__ __launch_bounds__(256) void col2im3x3_kernel(const bf16_t* __restrict__ dcols, float* __restrict__ dx,
                                                        int G, int H, int W, int C) {
  const int C8 = C >> 3;
  const int64_t total = (int64_t)G * H * W * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t pix = i / C8;
    const int xx = (int)(pix % W), yy = (int)((pix / W) % H);
    const int64_t g = pix / ((int64_t)W * H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // output pixel q read input pixel p through `tap` iff p = q + (tap/3-1, tap%3-1)
      const int qy = yy - (tap / 3 - 1), qx = xx - (tap % 3 - 1);
      if (qy >= 0 && qy < H && qx >= 0 && qx < W) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(dcols + ((g * H + qy) * W + qx) * 9 * C + (int64_t)tap * C + c8 * 8);
        acc[0] += bf16_lo(v.x); acc[1] += bf16_hi(v.x); acc[2] += bf16_lo(v.y); acc[3] += bf16_hi(v.y);
        acc[4] += bf16_lo(v.z); acc[5] += bf16_hi(v.z); acc[6] += bf16_lo(v.w); acc[7] += bf16_hi(v.w);
      }
    }
    float4* d = reinterpret_cast<float4*>(dx + pix * C + c8 * 8);
    d[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    d[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}
extern "C" int mfnet_im2col3x3(const void* x, void* cols, int64_t G, int64_t H, int64_t W, int64_t C, hipStream_t stream) {
  MF_CHECK_ARG(x && cols && C > 0 && C % 8 == 0, "im2col3x3: bad args");
  if (G * H * W <= 0) return MFNET_OK;
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for(G * H * W * 9 * (C / 8))), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)cols, (int)G, (int)H, (int)W, (int)C);
  MF_LAUNCH_CHECK("im2col3x3");
  return MFNET_OK;
}
extern "C" int mfnet_col2im3x3(const void* dcols, float* dx, int64_t G, int64_t H, int64_t W, int64_t C, hipStream_t stream) {
  MF_CHECK_ARG(dcols && dx && C > 0 && C % 8 == 0, "col2im3x3: bad args");
  if (G * H * W <= 0) return MFNET_OK;
  hipLaunchKernelGGL(col2im3x3_kernel, dim3(grid_for(G * H * W * (C / 8))), dim3(256), 0, stream, (const bf16_t*)dcols, dx, (int)G, (int)H, (int)W, (int)C);
  MF_LAUNCH_CHECK("col2im3x3");
  return MFNET_OK;
}

// ------------------------------------------------------------------------- LoRA cross-modal mix
// a [M, r] fp32 holds ax = x@Ax^T for the first M/2 rows and ay = y@Ay^T for the rest.
// U [M, 2r]: x rows -> [wx*ax, (1-wx)*ay] ; y rows -> [(1-wy)*ax, wy*ay]   (loralib/layers.py:159-160)
__global__ __launch_bounds__(256) void lora_mix_fwd_kernel(const float* __restrict__ a, const float* __restrict__ wx,
                                                           const float* __restrict__ wy, float* __restrict__ U,
                                                           int64_t Mh, int r) {
  const float fx = wx[0], fy = wy[0];
  const int64_t total = Mh * r;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / r;
    const int j = (int)(i % r);
    const float ax = a[t * r + j], ay = a[(Mh + t) * r + j];
    U[t * 2 * r + j] = fx * ax;
    U[t * 2 * r + r + j] = (1.f - fx) * ay;
    U[(Mh + t) * 2 * r + j] = (1.f - fy) * ax;
    U[(Mh + t) * 2 * r + r + j] = fy * ay;
  }
}
// P [M, 2r] = g @ [Bx|By].  da [M, r]: x rows <- wx*Px[:, :r] + (1-wy)*Py[:, :r] ; y rows <- (1-wx)*Px[:, r:] + wy*Py[:, r:]
// dw[0] += sum Px[:, :r]*ax - Px[:, r:]*ay ; dw[1] += sum Py[:, r:]*ay - Py[:, :r]*ax
__global__ __launch_bounds__(256) void lora_mix_bwd_kernel(const float* __restrict__ P, int nparts, int64_t part_stride, const float* __restrict__ a,
                                                           const float* __restrict__ wx, const float* __restrict__ wy,
                                                           float* __restrict__ da, float* __restrict__ dw, int64_t Mh, int r) {
  // P is the sum of `nparts` partial matrices part_stride floats apart (the column strips of mfnet_lora_side)
  const float fx = wx[0], fy = wy[0];
  const int64_t total = Mh * r;
  float sx = 0.f, sy = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / r;
    const int j = (int)(i % r);
    const float ax = a[t * r + j], ay = a[(Mh + t) * r + j];
    float pxx = 0.f, pxy = 0.f, pyx = 0.f, pyy = 0.f;
    for (int s = 0; s < nparts; ++s) {
      const float* Ps = P + (int64_t)s * part_stride;
      pxx += Ps[t * 2 * r + j]; pxy += Ps[t * 2 * r + r + j];
      pyx += Ps[(Mh + t) * 2 * r + j]; pyy += Ps[(Mh + t) * 2 * r + r + j];
    }
    da[t * r + j] = fx * pxx + (1.f - fy) * pyx;
    da[(Mh + t) * r + j] = (1.f - fx) * pxy + fy * pyy;
    sx += pxx * ax - pxy * ay;
    sy += pyy * ay - pyx * ax;
  }
  __shared__ float red[2][4];
  sx = wave_sum(sx);
  sy = wave_sum(sy);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sx; red[1][threadIdx.x >> 6] = sy; }
  __syncthreads();
  if (threadIdx.x < 2)     // one add per workgroup and scalar: same-address float atomics serialise
    atomicAdd(dw + threadIdx.x, (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]));
}
extern "C" int mfnet_lora_mix_fwd(const float* a, const float* wx, const float* wy, float* U, int64_t M, int r, hipStream_t stream) {
  MF_CHECK_ARG(a && wx && wy && U && r > 0 && M % 2 == 0, "lora_mix_fwd: bad args");
  if (M == 0) return MFNET_OK;
  hipLaunchKernelGGL(lora_mix_fwd_kernel, dim3(grid_for(M / 2 * r)), dim3(256), 0, stream, a, wx, wy, U, M / 2, r);
  MF_LAUNCH_CHECK("lora_mix_fwd");
  return MFNET_OK;
}
// dw (2 floats: d lora_wx, d lora_wy) is accumulated into; zero it first.
extern "C" int mfnet_lora_mix_bwd(const float* P, int nparts, int64_t part_stride, const float* a, const float* wx, const float* wy, float* da,
                                  float* dw, int64_t M, int r, hipStream_t stream) {
  MF_CHECK_ARG(P && a && wx && wy && da && dw && r > 0 && M % 2 == 0 && nparts >= 1 && (nparts == 1 || part_stride >= M * 2 * r),
               "lora_mix_bwd: bad args");
  if (M == 0) return MFNET_OK;
  unsigned g = grid_for(M / 2 * r);
  if (g > 128) g = 128;
  hipLaunchKernelGGL(lora_mix_bwd_kernel, dim3(g), dim3(256), 0, stream, P, nparts, part_stride, a, wx, wy, da, dw, M / 2, r);
  MF_LAUNCH_CHECK("lora_mix_bwd");
  return MFNET_OK;
}

// ------------------------------------------------------------------------- bf16 transpose (zero-padded)
// x [M, N] bf16 (ld) -> y [N, Mp] bf16 with Mp >= M; columns M..Mp-1 are zero. Operand prep for the
// weight-gradient GEMM dW = dY^T X of the trainable adapters (adapter.py:13-14).
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ x, int64_t ld,
                                                             bf16_t* __restrict__ y, int64_t M, int64_t N, int64_t Mp) {
  __shared__ bf16_t tile[64][66];
  const int64_t m0 = (int64_t)blockIdx.x * 64, n0 = (int64_t)blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int64_t m = m0 + r, n = n0 + tx;
    tile[r][tx] = (m < M && n < N) ? x[m * ld + n] : (bf16_t)0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int64_t n = n0 + r, m = m0 + tx;
    if (n < N && m < Mp) y[n * Mp + m] = tile[tx][r];
  }
}
extern "C" int mfnet_transpose_bf16(const void* x, int64_t ld, void* y, int64_t M, int64_t N, int64_t Mp, hipStream_t stream) {
  MF_CHECK_ARG(x && y && M > 0 && N > 0 && Mp >= M && ld >= N, "transpose_bf16: bad args");
  dim3 grid((unsigned)ceil_div64(Mp, 64), (unsigned)ceil_div64(N, 64));
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, stream, (const bf16_t*)x, ld, (bf16_t*)y, M, N, Mp);
  MF_LAUNCH_CHECK("transpose_bf16");
  return MFNET_OK;
}

// ------------------------------------------------------------------------- split reduction
// out[i] (= | +=) scale * sum_s partial[s*n + i]  : second stage of the split-K weight-gradient GEMMs
__global__ __launch_bounds__(256) void reduce_splits_kernel(const float* __restrict__ partial, float* __restrict__ out, int64_t n4, int splits,
                                                            float scale, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < splits; ++s) {
      const float4 v = reinterpret_cast<const float4*>(partial)[(int64_t)s * n4 + i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    float4 o = make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale);
    if (accumulate) { const float4 p = reinterpret_cast<float4*>(out)[i]; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
    reinterpret_cast<float4*>(out)[i] = o;
  }
}
extern "C" int mfnet_reduce_splits(const float* partial, float* out, int64_t n, int splits, float scale, int accumulate, hipStream_t stream) {
  MF_CHECK_ARG(partial && out && n >= 0 && n % 4 == 0 && splits >= 1, "reduce_splits: bad args");
  if (n == 0) return MFNET_OK;
  hipLaunchKernelGGL(reduce_splits_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, partial, out, n / 4, splits, scale, accumulate);
  MF_LAUNCH_CHECK("reduce_splits");
  return MFNET_OK;
}

// ---------------------------------------------------------------------------------------------------
// Strided gather of many small fp32 tensors in one launch. Each segment copies a logical <=4-D fp32 tensor
// with arbitrary element strides into another (fp32 assign / fp32 accumulate / bf16 assign). The train step
// uses it (a) once per step for every operand derived from trainable weights (bf16 GEMM images of the
// decoder weights in both layouts, the stacked / transposed LoRA factors), and (b) at the end of each
// module's backward to add the parameter gradients, in whatever layout the kernels produced them, into the
// flat gradient buffer - instead of ~700 framework copy / cat / add launches per step.
#define SEG_MAX 48
struct GatherSeg { const float* src; void* dst; int dims[4]; int sstr[4]; int dstr[4]; int red_n; int red_stride; float scale; int pad_; };   // 80 bytes
struct GatherParams { GatherSeg seg[SEG_MAX]; };

__global__ __launch_bounds__(256) void gather_segments_kernel(const GatherParams p, int mode) {
  const GatherSeg& s = p.seg[blockIdx.y];
  const unsigned d1 = s.dims[1], d2 = s.dims[2], d3 = s.dims[3];
  const unsigned n = (unsigned)s.dims[0] * d1 * d2 * d3;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    unsigned t = i / d3;
    const unsigned i3 = i - t * d3;
    unsigned u = t / d2;
    const unsigned i2 = t - u * d2;
    const unsigned i0 = u / d1;
    const unsigned i1 = u - i0 * d1;
    const int64_t so = (int64_t)i0 * s.sstr[0] + (int64_t)i1 * s.sstr[1] + (int64_t)i2 * s.sstr[2] + (int64_t)i3 * s.sstr[3];
    const int64_t dofs = (int64_t)i0 * s.dstr[0] + (int64_t)i1 * s.dstr[1] + (int64_t)i2 * s.dstr[2] + (int64_t)i3 * s.dstr[3];
    float v = s.src[so];
    if (s.red_n > 1) {            // the source is a stack of red_n partial results (split reductions): summed here, in a
      const float* q = s.src + so;   // fixed order, eight loads in flight
      float a[8] = {v, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      int k = 1;
      for (; k + 7 < s.red_n; k += 8) {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = q[(int64_t)(k + e) * s.red_stride];
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += t[e];
      }
      for (; k < s.red_n; ++k) a[0] += q[(int64_t)k * s.red_stride];
      v = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    v *= s.scale;
    if (mode == 2) ((bf16_t*)s.dst)[dofs] = f32_to_bf16(v);
    else if (mode == 1) ((float*)s.dst)[dofs] += v;
    else ((float*)s.dst)[dofs] = v;
  }
}

// desc: HOST array of n x 17 int64: {src ptr, dst ptr, dims[4], src strides[4], dst strides[4], red_n, red_stride, scale}
// (strides in elements; scale = the bits of a float in the low 32 bits). red_n > 1: the value copied / added is
// scale * sum_{k < red_n} src[... + k*red_stride] - the partial tiles of a split reduction are summed on the way.
extern "C" int mfnet_gather_segments(const int64_t* desc, int n, int dst_dtype, int accumulate, hipStream_t stream) {
  MF_CHECK_ARG(n >= 0 && (n == 0 || desc), "gather_segments: bad args");
  MF_CHECK_ARG(dst_dtype == MFNET_F32 || (dst_dtype == MFNET_BF16 && !accumulate), "gather_segments: accumulate needs an fp32 destination");
  const int mode = dst_dtype == MFNET_BF16 ? 2 : (accumulate ? 1 : 0);
  // Segments are independent; a launch's grid is sized by its largest one, so they go out sorted by work: the few big ones
  // (FPN / conv weight gradients with their partial tiles) share the first launch and the hundreds of small ones run in launches
  // of a few workgroups each (unsorted, every launch carried one big segment and 47 columns of idle workgroups).
  std::vector<int> order(n);
  std::vector<int64_t> work(n);
  for (int i = 0; i < n; ++i) {
    const int64_t* d = desc + (int64_t)i * 17;
    int64_t tot = 1;
    for (int k = 0; k < 4; ++k) tot *= (d[2 + k] >= 1 && d[2 + k] < (1ll << 31)) ? d[2 + k] : 1;     // (validated below)
    order[i] = i;
    work[i] = tot * (d[14] < 1 ? 1 : (d[14] < 16 ? d[14] : 16));
  }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return work[a] > work[b]; });
  for (int base = 0; base < n; base += SEG_MAX) {
    const int cnt = n - base < SEG_MAX ? n - base : SEG_MAX;
    GatherParams p;
    int64_t nmax = 0;
    for (int i = 0; i < SEG_MAX; ++i) {
      const int64_t* d = desc + (int64_t)order[base + (i < cnt ? i : 0)] * 17;      // unused slots repeat a valid one
      GatherSeg& s = p.seg[i];
      s.src = (const float*)(uintptr_t)d[0]; s.dst = (void*)(uintptr_t)d[1];
      MF_CHECK_ARG(s.src && s.dst, "gather_segments: null pointer in segment %d", base + i);
      int64_t tot = 1;
      for (int k = 0; k < 4; ++k) {
        MF_CHECK_ARG(d[2 + k] >= 1 && d[2 + k] < (1ll << 31), "gather_segments: bad dim in segment %d", base + i);
        MF_CHECK_ARG(d[6 + k] >= 0 && d[6 + k] < (1ll << 31) && d[10 + k] >= 0 && d[10 + k] < (1ll << 31), "gather_segments: stride out of range in segment %d", base + i);
        s.dims[k] = (int)d[2 + k]; s.sstr[k] = (int)d[6 + k]; s.dstr[k] = (int)d[10 + k];
        tot *= d[2 + k];
      }
      MF_CHECK_ARG(tot < (1ll << 31), "gather_segments: segment %d too large", base + i);
      MF_CHECK_ARG(d[14] >= 1 && d[14] <= 4096 && d[15] >= 0 && d[15] < (1ll << 31), "gather_segments: bad reduction in segment %d", base + i);
      s.red_n = (int)d[14]; s.red_stride = (int)d[15]; s.pad_ = 0;
      if (s.red_n > 1) {
        // a summed stack reads red_n times what it writes: walk the index space with the smallest SOURCE stride fastest
        // (a conv weight gradient arrives as [O, kh, kw, I] tiles and leaves as [O, I, kh, kw]) - same sums, coalesced reads
        for (int a = 1; a < 4; ++a)
          for (int b = a; b > 0 && (s.dims[b - 1] == 1 ? 0 : s.sstr[b - 1]) < (s.dims[b] == 1 ? 0 : s.sstr[b]) ; --b) {
            const int td = s.dims[b], ts = s.sstr[b], tt = s.dstr[b];
            s.dims[b] = s.dims[b - 1]; s.sstr[b] = s.sstr[b - 1]; s.dstr[b] = s.dstr[b - 1];
            s.dims[b - 1] = td; s.sstr[b - 1] = ts; s.dstr[b - 1] = tt;
          }
      }
      { const unsigned int bits = (unsigned int)(d[16] & 0xffffffffll); memcpy(&s.scale, &bits, 4); }
      if (i < cnt && tot * (d[14] < 16 ? d[14] : 16) > nmax) nmax = tot * (d[14] < 16 ? d[14] : 16);   // split sums: more, shorter threads
    }
    int64_t gx = ceil_div64(nmax, 256 * 8);
    gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);       // a summed stack is latency-bound per thread: many short threads
    hipLaunchKernelGGL(gather_segments_kernel, dim3((unsigned)gx, (unsigned)cnt), dim3(256), 0, stream, p, mode);
    MF_LAUNCH_CHECK("gather_segments");
  }
  return MFNET_OK;
}
