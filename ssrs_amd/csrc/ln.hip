// LayerNorm over the last (channel) dimension of a row-major [rows, C] token matrix.
// HBM-bound: one 64-lane wave per row, the row lives in registers (float4 per lane per step),
// two-pass mean / variance with wave shuffles. Algorithmic bytes: read x + write y per call.
//
// Reference semantics: nn.LayerNorm(eps=1e-6) in the ViT blocks (medsam/models/sam/build_sam.py:80),
// LayerNorm2d on NCHW (medsam/models/common/layer_norm.py:12-17) and Norm2d
// (unetformer_mmsam.py:14-22) - all the same op once tokens are channels-last.
#include "common.h"
#include <mutex>

#define LN_MAXV 8   // float4 chunks per lane -> C <= 64*4*8 = 2048

template <typename TIn>
__device__ __forceinline__ float4 ld4(const TIn* p);
template <>
__device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <>
__device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t* p) {
  u32x2_t w = *reinterpret_cast<const u32x2_t*>(p);
  return make_float4(bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y));
}
template <typename TOut>
__device__ __forceinline__ void st4(TOut* p, float4 v);
template <>
__device__ __forceinline__ void st4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <>
__device__ __forceinline__ void st4<bf16_t>(bf16_t* p, float4 v) {
  u32x2_t w;
  w.x = pack_bf16x2(v.x, v.y);
  w.y = pack_bf16x2(v.z, v.w);
  *reinterpret_cast<u32x2_t*>(p) = w;
}

template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TIn* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TOut* __restrict__ y,
                                                     int64_t ldy, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int64_t rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TIn* xr = x + row * ldx;
  const int nv = C >> 2;
  float4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv) {
      v[i] = ld4<TIn>(xr + 4 * c4);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv) {
      const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
  TOut* yr = y + row * ldy;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * c4);
      const float4 b = *reinterpret_cast<const float4*>(beta + 4 * c4);
      float4 o;
      o.x = (v[i].x - mu) * rs * g.x + b.x;
      o.y = (v[i].y - mu) * rs * g.y + b.y;
      o.z = (v[i].z - mu) * rs * g.z + b.z;
      o.w = (v[i].w - mu) * rs * g.w + b.w;
      st4<TOut>(yr + 4 * c4, o);
    }
  }
}

// LayerNorm forward fused with the LoRA down-projection of the layer that consumes it (loralib/layers.py:157-158, 253, 261):
//   y = bf16(LN(x)),  a[row, j] = sum_c y[row, c] * bf16(A_g[j, c])   (A fp32 [G][R][C], g = the row's modality; the same
//   bf16 x bf16 -> fp32 product skinny_nt forms)
// and, for lora.Linear (MIX), the cross-modal mix of layers.py:159-160 on the spot:
//   U[x row t] = [wx*ax[t], (1-wx)*ay[t]],  U[y row t] = [(1-wy)*ax[t], wy*ay[t]].
// The wave already holds its whole row in registers, so the rank-R product is R*C/64 FMAs per lane and a halving
// butterfly (R + log2(64) - 1 shuffles) instead of a second kernel that re-reads y (skinny_nt: 11.6 us per launch at
// 8192 x 768) and a third one for the mix. A workgroup is 16 waves = 16 rows and stages A once in LDS as bf16 (a first
// version read A from global per row: 24 KB per 3 KB row, 200 MB of L2 traffic per launch - as slow as the kernels it
// replaced). With two groups a workgroup takes 8 token positions of BOTH modalities (wave 2p: x row, 2p+1: y row of
// position 8b+p), so the mix needs only LDS.
// A (fp32, tot4 float4) -> bf16 in LDS by the 1024 threads of a workgroup, four loads in flight per thread (one load per
// iteration of a runtime-trip loop is one dependent L2 round trip each: 6 of them at C = 3072, R = 4)
__device__ __forceinline__ void stage_lora_a(const float* __restrict__ A, bf16_t* Al, int tot4) {
  for (int base = threadIdx.x; base < tot4; base += 4096) {
    float4 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + 1024 * u;
      t[u] = *reinterpret_cast<const float4*>(A + 4 * (int64_t)(i < tot4 ? i : base));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + 1024 * u;
      if (i < tot4) {
        u32x2_t wv;
        wv.x = pack_bf16x2(t[u].x, t[u].y);
        wv.y = pack_bf16x2(t[u].z, t[u].w);
        *reinterpret_cast<u32x2_t*>(Al + 4 * (int64_t)i) = wv;
      }
    }
  }
}

template <typename TIn, int R, bool MIX, int NVL>
__global__ __launch_bounds__(1024) void ln_fwd_lora_kernel(const TIn* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, bf16_t* __restrict__ y, int64_t ldy,
                                                           float* __restrict__ mean, float* __restrict__ rstd, int64_t rows, int C, float eps,
                                                           const float* __restrict__ A, int64_t half, int n_groups,
                                                           float* __restrict__ a_out, const float* __restrict__ wx, const float* __restrict__ wy,
                                                           float* __restrict__ U) {
  extern __shared__ __align__(16) unsigned char ln_lds[];
  bf16_t* Al = reinterpret_cast<bf16_t*>(ln_lds);                                 // [G][R][C] bf16
  float* al = reinterpret_cast<float*>(ln_lds + (size_t)n_groups * R * C * 2);    // [8][2][R]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int g = 0;
  int64_t row;
  bool valid;
  if (n_groups == 2) {
    g = w & 1;
    const int64_t t = 8 * (int64_t)blockIdx.x + (w >> 1);
    valid = t < half;
    row = g * half + t;
  } else {
    row = (int64_t)blockIdx.x * 16 + w;
    valid = row < rows;
  }
  const int nv = C >> 2;
  float4 v[NVL];
  if (valid) {
    const TIn* xr = x + row * ldx;
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < nv) v[i] = ld4<TIn>(xr + 4 * c4);
    }
  }
  {   // stage A (fp32 -> bf16) while the row loads are in flight
    const int tot4 = n_groups * R * nv;
    stage_lora_a(A, Al, tot4);
  }
  float p[R];
#pragma unroll
  for (int j = 0; j < R; ++j) p[j] = 0.f;
  float mu = 0.f, rs = 0.f;
  if (valid) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NVL; ++i)
      if (lane + 64 * i < nv) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NVL; ++i)
      if (lane + 64 * i < nv) {
        const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
        q += (a * a + b * b) + (c * c + d * d);
      }
    rs = rsqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0) {
      if (mean) mean[row] = mu;
      if (rstd) rstd[row] = rs;
    }
  }
  __syncthreads();                                             // A staged
  if (valid) {
    bf16_t* yr = y + row * ldy;
    const bf16_t* Ag = Al + (size_t)g * R * C;
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < nv) {
        const float4 gm = *reinterpret_cast<const float4*>(gamma + 4 * c4);
        const float4 bt = *reinterpret_cast<const float4*>(beta + 4 * c4);
        u32x2_t wv;
        wv.x = pack_bf16x2((v[i].x - mu) * rs * gm.x + bt.x, (v[i].y - mu) * rs * gm.y + bt.y);
        wv.y = pack_bf16x2((v[i].z - mu) * rs * gm.z + bt.z, (v[i].w - mu) * rs * gm.w + bt.w);
        *reinterpret_cast<u32x2_t*>(yr + 4 * c4) = wv;
        const float o0 = bf16_lo(wv.x), o1 = bf16_hi(wv.x), o2 = bf16_lo(wv.y), o3 = bf16_hi(wv.y);     // what the GEMM will read
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const u32x2_t av = *reinterpret_cast<const u32x2_t*>(Ag + (size_t)j * C + 4 * c4);
          p[j] = fmaf(o0, bf16_lo(av.x), fmaf(o1, bf16_hi(av.x), fmaf(o2, bf16_lo(av.y), fmaf(o3, bf16_hi(av.y), p[j]))));
        }
      }
    }
  }
  // halving butterfly: after the step with offset `off` a lane keeps the half of its values selected by that lane bit
  int jb = 0, off = 32;
#pragma unroll
  for (int n = R; n > 1; n >>= 1, off >>= 1) {
    const bool up = (lane & off) != 0;
#pragma unroll
    for (int i = 0; i < n / 2; ++i) {
      const float send = up ? p[i] : p[i + n / 2];
      const float recv = __shfl_xor(send, off, 64);
      p[i] = (up ? p[i + n / 2] : p[i]) + recv;
    }
    jb += up ? n / 2 : 0;
  }
  for (; off > 0; off >>= 1) p[0] += __shfl_xor(p[0], off, 64);
  const bool writer = (lane & (64 / R - 1)) == 0;            // one lane per j
  if (valid && writer) a_out[row * R + jb] = p[0];
  if (MIX) {
    if (writer) al[((w >> 1) * 2 + g) * R + jb] = p[0];
    __syncthreads();
    if (valid && lane < 2 * R) {
      const float fx = wx[0], fy = wy[0];
      const int jj = lane < R ? lane : lane - R;
      const float ax = al[((w >> 1) * 2 + 0) * R + jj], ay = al[((w >> 1) * 2 + 1) * R + jj];
      float u;
      if (g == 0) u = lane < R ? fx * ax : (1.f - fx) * ay;
      else u = lane < R ? (1.f - fy) * ax : fy * ay;
      U[row * 2 * R + lane] = u;
    }
  }
}

// The same down-projection + mix for a layer whose input is NOT a LayerNorm output (lin2 of the MLP reads GELU(lin1)):
// x bf16 [rows, C] -> a [rows, R], U [rows, 2R]. One wave per row (16-byte loads, the row is read once, contiguous 1 KB per
// wave-instruction), A as bf16 in LDS, halving butterfly, mix through LDS - replaces skinny_nt (18.7 us at 8192 x 3072: its
// 16-row MFMA tiles touch 16 rows x 128 B per request) + lora_mix_fwd.
template <int R, bool MIX>
__global__ __launch_bounds__(1024) void lora_down_mix_kernel(const bf16_t* __restrict__ x, int64_t ldx, int64_t rows, int C, const float* __restrict__ A,
                                                             int64_t half, int n_groups, float* __restrict__ a_out, const float* __restrict__ wx,
                                                             const float* __restrict__ wy, float* __restrict__ U, int64_t nblk) {
  extern __shared__ __align__(16) unsigned char ln_lds[];
  bf16_t* Al = reinterpret_cast<bf16_t*>(ln_lds);                                 // [G][R][C] bf16
  float* al = reinterpret_cast<float*>(ln_lds + (size_t)n_groups * R * C * 2);    // [8][2][R]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = n_groups == 2 ? (w & 1) : 0;
  const int n8 = C >> 3;                       // 16-byte chunks per row
  constexpr int NCH = 8;                       // up to 8 chunks per lane: C <= 4096
  // Persistent: a workgroup stages A once (it is n_groups*R*C floats - 98 KB at C = 3072 - and with one 16-row block per
  // workgroup every block paid that read and its barrier) and then walks 16-row blocks blockIdx.x, + gridDim.x, ...
  auto locate = [&](int64_t blk, int64_t* row) -> bool {
    if (n_groups == 2) {
      const int64_t t = 8 * blk + (w >> 1);
      *row = g * half + t;
      return blk < nblk && t < half;
    }
    *row = blk * 16 + w;
    return blk < nblk && *row < rows;
  };
  auto request = [&](bool ok, int64_t row, u32x4_t (&v)[NCH]) {
    if (ok) {
      const bf16_t* xr = x + row * ldx;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c8 = lane + 64 * i;
        if (c8 < n8) v[i] = *reinterpret_cast<const u32x4_t*>(xr + 8 * c8);
      }
    }
  };
  u32x4_t v[NCH];
  int64_t row;
  bool valid = locate(blockIdx.x, &row);
  request(valid, row, v);
  {
    const int tot4 = n_groups * R * (C >> 2);
    stage_lora_a(A, Al, tot4);
  }
  __syncthreads();
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    float p[R];
#pragma unroll
    for (int j = 0; j < R; ++j) p[j] = 0.f;
    if (valid) {
      const bf16_t* Ag = Al + (size_t)g * R * C;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c8 = lane + 64 * i;
        if (c8 < n8) {
          const float o[8] = {bf16_lo(v[i].x), bf16_hi(v[i].x), bf16_lo(v[i].y), bf16_hi(v[i].y), bf16_lo(v[i].z), bf16_hi(v[i].z), bf16_lo(v[i].w), bf16_hi(v[i].w)};
#pragma unroll
          for (int j = 0; j < R; ++j) {
            const u32x4_t av = *reinterpret_cast<const u32x4_t*>(Ag + (size_t)j * C + 8 * c8);
            p[j] = fmaf(o[0], bf16_lo(av.x), fmaf(o[1], bf16_hi(av.x), fmaf(o[2], bf16_lo(av.y), fmaf(o[3], bf16_hi(av.y),
                   fmaf(o[4], bf16_lo(av.z), fmaf(o[5], bf16_hi(av.z), fmaf(o[6], bf16_lo(av.w), fmaf(o[7], bf16_hi(av.w), p[j]))))))));
          }
        }
      }
    }
    int jb = 0, off = 32;
#pragma unroll
    for (int n = R; n > 1; n >>= 1, off >>= 1) {
      const bool up = (lane & off) != 0;
#pragma unroll
      for (int i = 0; i < n / 2; ++i) {
        const float send = up ? p[i] : p[i + n / 2];
        const float recv = __shfl_xor(send, off, 64);
        p[i] = (up ? p[i + n / 2] : p[i]) + recv;
      }
      jb += up ? n / 2 : 0;
    }
    for (; off > 0; off >>= 1) p[0] += __shfl_xor(p[0], off, 64);
    const bool writer = (lane & (64 / R - 1)) == 0;
    if (valid && writer) a_out[row * R + jb] = p[0];
    if (MIX) {
      if (writer) al[((w >> 1) * 2 + g) * R + jb] = p[0];
      __syncthreads();
      if (valid && lane < 2 * R) {
        const float fx = wx[0], fy = wy[0];
        const int jj = lane < R ? lane : lane - R;
        const float ax = al[((w >> 1) * 2 + 0) * R + jj], ay = al[((w >> 1) * 2 + 1) * R + jj];
        float u;
        if (g == 0) u = lane < R ? fx * ax : (1.f - fx) * ay;
        else u = lane < R ? (1.f - fy) * ax : fy * ay;
        U[row * 2 * R + lane] = u;
      }
      __syncthreads();                       // `al` is rewritten by the next block
    }
    valid = locate(blk + gridDim.x, &row);     // (a second register set for a prefetch spills at 16 waves per workgroup: 128 VGPRs)
    request(valid, row, v);
  }
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)) [+ dres]; optional bf16 copy of dx;
// optional dgamma/dbeta accumulation (float atomics, one add per column per wave).
// NVL = float4 chunks per lane (C <= 256 * NVL): the per-row register arrays are sized for the row at hand, which keeps
// enough waves resident for the loads in flight to cover HBM latency (sized for C = 2048 the kernel ran 2 waves / SIMD).
template <typename TDy, typename TX, int NVL, bool WANT_W>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDy* __restrict__ dy, int64_t lddy,
                                                     const TX* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma,
                                                     const float* dres, float* dx,
                                                     bf16_t* __restrict__ dx_bf16, int64_t lddx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     int64_t rows, int C) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nv = C >> 2;
  float4 ag[NVL], ab[NVL];
  constexpr bool want_w = WANT_W;
  if (want_w) {
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      ag[i] = make_float4(0, 0, 0, 0);
      ab[i] = make_float4(0, 0, 0, 0);
    }
  }
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    float4 xh[NVL], gd[NVL], rr[NVL];
    float s1 = 0.f, s2 = 0.f;
    if (dres) {        // requested with x and dy: the row costs one memory round trip, not two
#pragma unroll
      for (int i = 0; i < NVL; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < nv) rr[i] = *reinterpret_cast<const float4*>(dres + row * lddx + 4 * c4);
      }
    }
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < nv) {
        const float4 xv = ld4<TX>(x + row * ldx + 4 * c4);
        const float4 d = ld4<TDy>(dy + row * lddy + 4 * c4);
        const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * c4);
        xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        gd[i] = make_float4(g.x * d.x, g.y * d.y, g.z * d.z, g.w * d.w);
        s1 += (gd[i].x + gd[i].y) + (gd[i].z + gd[i].w);
        s2 += (gd[i].x * xh[i].x + gd[i].y * xh[i].y) + (gd[i].z * xh[i].z + gd[i].w * xh[i].w);
        if (want_w) {
          ag[i].x += d.x * xh[i].x; ag[i].y += d.y * xh[i].y; ag[i].z += d.z * xh[i].z; ag[i].w += d.w * xh[i].w;
          ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
        }
      }
    }
    const float m1 = wave_sum(s1) / (float)C;
    const float m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < nv) {
        float4 o;
        o.x = rs * (gd[i].x - m1 - xh[i].x * m2);
        o.y = rs * (gd[i].y - m1 - xh[i].y * m2);
        o.z = rs * (gd[i].z - m1 - xh[i].z * m2);
        o.w = rs * (gd[i].w - m1 - xh[i].w * m2);
        if (dres) { o.x += rr[i].x; o.y += rr[i].y; o.z += rr[i].z; o.w += rr[i].w; }
        if (dx) *reinterpret_cast<float4*>(dx + row * lddx + 4 * c4) = o;
        if (dx_bf16) st4<bf16_t>(dx_bf16 + row * lddx + 4 * c4, o);
      }
    }
  }
  if (want_w) {
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < nv) {
        atomicAdd(dgamma + 4 * c4 + 0, ag[i].x); atomicAdd(dgamma + 4 * c4 + 1, ag[i].y);
        atomicAdd(dgamma + 4 * c4 + 2, ag[i].z); atomicAdd(dgamma + 4 * c4 + 3, ag[i].w);
        atomicAdd(dbeta + 4 * c4 + 0, ab[i].x); atomicAdd(dbeta + 4 * c4 + 1, ab[i].y);
        atomicAdd(dbeta + 4 * c4 + 2, ab[i].z); atomicAdd(dbeta + 4 * c4 + 3, ab[i].w);
      }
    }
  }
}

extern "C" int mfnet_layernorm_fwd(const void* x, int x_dtype, int64_t ldx, const float* gamma,
                                   const float* beta, void* y, int y_dtype, int64_t ldy, float* mean,
                                   float* rstd, int64_t rows, int64_t C, float eps, hipStream_t stream) {
  MF_CHECK_ARG(x && y && gamma && beta, "layernorm_fwd: null pointer");
  MF_CHECK_ARG(rows >= 0 && C > 0 && C % 4 == 0 && C <= 64 * 4 * LN_MAXV,
               "layernorm_fwd: C=%lld must be a multiple of 4 and <= %d", (long long)C, 64 * 4 * LN_MAXV);
  MF_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C, "layernorm_fwd: bad leading dims");
  if (rows == 0) return MFNET_OK;
  dim3 grid((unsigned)ceil_div64(rows, 4)), block(256);
#define LN_FWD(TI, TO)                                                                                      \
  hipLaunchKernelGGL((ln_fwd_kernel<TI, TO>), grid, block, 0, stream, (const TI*)x, ldx, gamma, beta, (TO*)y, \
                     ldy, mean, rstd, rows, (int)C, eps)
  if (x_dtype == MFNET_F32 && y_dtype == MFNET_BF16) LN_FWD(float, bf16_t);
  else if (x_dtype == MFNET_F32 && y_dtype == MFNET_F32) LN_FWD(float, float);
  else if (x_dtype == MFNET_BF16 && y_dtype == MFNET_BF16) LN_FWD(bf16_t, bf16_t);
  else if (x_dtype == MFNET_BF16 && y_dtype == MFNET_F32) LN_FWD(bf16_t, float);
  else { mfnet_set_error("layernorm_fwd: bad dtype"); return MFNET_ERR_ARG; }
#undef LN_FWD
  MF_LAUNCH_CHECK("layernorm_fwd");
  return MFNET_OK;
}

extern "C" int mfnet_layernorm_fwd_lora(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy,
                                        float* mean, float* rstd, int64_t rows, int64_t C, float eps, const float* A, int64_t a_group_stride,
                                        int n_groups, int R, float* a_out, const float* wx, const float* wy, float* U, hipStream_t stream) {
  MF_CHECK_ARG(x && y && gamma && beta && A && a_out, "layernorm_fwd_lora: null pointer");
  MF_CHECK_ARG(rows >= 0 && C > 0 && C % 4 == 0 && C <= 64 * 4 * LN_MAXV, "layernorm_fwd_lora: C=%lld must be a multiple of 4 and <= %d",
               (long long)C, 64 * 4 * LN_MAXV);
  MF_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C, "layernorm_fwd_lora: bad leading dims");
  MF_CHECK_ARG(R == 4 || R == 8, "layernorm_fwd_lora: R=%d (4 and 8 are built)", R);
  MF_CHECK_ARG(n_groups == 1 || (n_groups == 2 && rows % 2 == 0 && a_group_stride == (int64_t)R * C),
               "layernorm_fwd_lora: one group, or two contiguous groups [2][R][C] of rows/2 rows each");
  MF_CHECK_ARG(!U || (wx && wy && n_groups == 2), "layernorm_fwd_lora: the cross-modal mix needs wx, wy and two groups");
  MF_CHECK_ARG(((uintptr_t)A % 16) == 0, "layernorm_fwd_lora: A must be 16-byte aligned");
  const size_t lds = (size_t)n_groups * R * C * 2 + 8 * 2 * R * sizeof(float);
  MF_CHECK_ARG(lds <= 64 * 1024, "layernorm_fwd_lora: G*R*C = %lld does not fit the 64 KB LDS image", (long long)n_groups * R * C);
  if (rows == 0) return MFNET_OK;
  const int64_t half = rows / 2;
  const int64_t nblk = n_groups == 2 ? ceil_div64(half, 8) : ceil_div64(rows, 16);
  MF_CHECK_ARG(nblk < (1ll << 31), "layernorm_fwd_lora: too many rows");
  dim3 grid((unsigned)nblk), block(1024);
#define LN_LORA(TI, RR, MX, NV)                                                                                                \
  hipLaunchKernelGGL((ln_fwd_lora_kernel<TI, RR, MX, NV>), grid, block, lds, stream, (const TI*)x, ldx, gamma, beta, (bf16_t*)y, ldy, mean, \
                     rstd, rows, (int)C, eps, A, half, n_groups, a_out, wx, wy, U)
#define LN_LORA_N(TI, RR, MX)                                   \
  do {                                                          \
    if (C <= 768) LN_LORA(TI, RR, MX, 3);                       \
    else if (C <= 1280) LN_LORA(TI, RR, MX, 5);                 \
    else LN_LORA(TI, RR, MX, LN_MAXV);                          \
  } while (0)
#define LN_LORA_T(TI)                                           \
  do {                                                          \
    if (R == 8 && U) LN_LORA_N(TI, 8, true);                    \
    else if (R == 8) LN_LORA_N(TI, 8, false);                   \
    else if (U) LN_LORA_N(TI, 4, true);                         \
    else LN_LORA_N(TI, 4, false);                               \
  } while (0)
  if (x_dtype == MFNET_F32) LN_LORA_T(float);
  else if (x_dtype == MFNET_BF16) LN_LORA_T(bf16_t);
  else { mfnet_set_error("layernorm_fwd_lora: bad dtype"); return MFNET_ERR_ARG; }
#undef LN_LORA_T
#undef LN_LORA_N
#undef LN_LORA
  MF_LAUNCH_CHECK("layernorm_fwd_lora");
  return MFNET_OK;
}

extern "C" int mfnet_lora_down_mix(const void* x, int64_t ldx, int64_t rows, int64_t C, const float* A, int64_t a_group_stride, int n_groups, int R,
                                   float* a_out, const float* wx, const float* wy, float* U, hipStream_t stream) {
  MF_CHECK_ARG(x && A && a_out, "lora_down_mix: null pointer");
  MF_CHECK_ARG(rows >= 0 && C > 0 && C % 8 == 0 && C <= 4096 && ldx % 8 == 0 && ldx >= C && (uintptr_t)x % 16 == 0 && (uintptr_t)A % 16 == 0,
               "lora_down_mix: C=%lld must be a multiple of 8 and <= 4096 (16-byte aligned rows)", (long long)C);
  MF_CHECK_ARG(R == 4 || R == 8, "lora_down_mix: R=%d (4 and 8 are built)", R);
  MF_CHECK_ARG(n_groups == 1 || (n_groups == 2 && rows % 2 == 0 && a_group_stride == (int64_t)R * C),
               "lora_down_mix: one group, or two contiguous groups [2][R][C] of rows/2 rows each");
  MF_CHECK_ARG(!U || (wx && wy && n_groups == 2), "lora_down_mix: the cross-modal mix needs wx, wy and two groups");
  const size_t lds = (size_t)n_groups * R * C * 2 + 8 * 2 * R * sizeof(float);
  MF_CHECK_ARG(lds <= 96 * 1024, "lora_down_mix: G*R*C = %lld does not fit the 96 KB LDS image", (long long)n_groups * R * C);
  if (lds > 64 * 1024) {
    // more than 64 KB of dynamic LDS needs the attribute once per device (C = 4096 with two groups of rank 4: ViT-L's lin2); the
    // per-device flag is the only state kept (guarded: entry points may be called from any thread)
    static std::mutex mu;
    static bool done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { mfnet_set_error("lora_down_mix: hipGetDevice failed"); return MFNET_ERR_HIP; }
    std::lock_guard<std::mutex> lk(mu);
    if (!done[dev]) {
      const void* fns[4] = {(const void*)lora_down_mix_kernel<8, true>, (const void*)lora_down_mix_kernel<8, false>,
                            (const void*)lora_down_mix_kernel<4, true>, (const void*)lora_down_mix_kernel<4, false>};
      for (int i = 0; i < 4; ++i) {
        hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) { mfnet_set_error("lora_down_mix: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MFNET_ERR_HIP; }
      }
      done[dev] = true;
    }
  }
  if (rows == 0) return MFNET_OK;
  const int64_t half = rows / 2;
  const int64_t nblk = n_groups == 2 ? ceil_div64(half, 8) : ceil_div64(rows, 16);
  MF_CHECK_ARG(nblk < (1ll << 31), "lora_down_mix: too many rows");
  dim3 grid((unsigned)(nblk < 256 ? nblk : 256)), block(1024);      // persistent: one 16-wave workgroup per CU walks the blocks
#define LDM(RR, MX) hipLaunchKernelGGL((lora_down_mix_kernel<RR, MX>), grid, block, lds, stream, (const bf16_t*)x, ldx, rows, (int)C, A, half, n_groups, a_out, wx, wy, U, nblk)
  if (R == 8 && U) LDM(8, true);
  else if (R == 8) LDM(8, false);
  else if (U) LDM(4, true);
  else LDM(4, false);
#undef LDM
  MF_LAUNCH_CHECK("lora_down_mix");
  return MFNET_OK;
}

extern "C" int mfnet_layernorm_bwd(const void* dy, int dy_dtype, int64_t lddy, const void* x, int x_dtype,
                                   int64_t ldx, const float* mean, const float* rstd, const float* gamma,
                                   const float* dres, float* dx, void* dx_bf16, int64_t lddx, float* dgamma,
                                   float* dbeta, int64_t rows, int64_t C, hipStream_t stream) {
  MF_CHECK_ARG(dy && x && mean && rstd && gamma && (dx || dx_bf16), "layernorm_bwd: null pointer");
  MF_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 64 * 4 * LN_MAXV, "layernorm_bwd: bad C=%lld", (long long)C);
  MF_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "layernorm_bwd: dgamma/dbeta must come together");
  MF_CHECK_ARG(lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0, "layernorm_bwd: bad leading dims");
  if (rows == 0) return MFNET_OK;
  int64_t nblk = ceil_div64(rows, 4);
  if (dgamma && nblk > 512) nblk = 512;       // fewer waves -> fewer column atomics
  if (nblk > 65535 * 16) nblk = 65535 * 16;
  dim3 grid((unsigned)nblk), block(256);
#define LN_BWD_N(TD, TX, NVL)                                                                                      \
  do {                                                                                                             \
    if (dgamma)                                                                                                    \
      hipLaunchKernelGGL((ln_bwd_kernel<TD, TX, NVL, true>), grid, block, 0, stream, (const TD*)dy, lddy, (const TX*)x, ldx, \
                         mean, rstd, gamma, dres, dx, (bf16_t*)dx_bf16, lddx, dgamma, dbeta, rows, (int)C);        \
    else                                                                                                           \
      hipLaunchKernelGGL((ln_bwd_kernel<TD, TX, NVL, false>), grid, block, 0, stream, (const TD*)dy, lddy, (const TX*)x, ldx, \
                         mean, rstd, gamma, dres, dx, (bf16_t*)dx_bf16, lddx, dgamma, dbeta, rows, (int)C);        \
  } while (0)
#define LN_BWD(TD, TX)                                                                                             \
  do {                                                                                                             \
    if (C <= 256) LN_BWD_N(TD, TX, 1);                                                                             \
    else if (C <= 768) LN_BWD_N(TD, TX, 3);                                                                        \
    else if (C <= 1280) LN_BWD_N(TD, TX, 5);                                                                       \
    else LN_BWD_N(TD, TX, LN_MAXV);                                                                                \
  } while (0)
  if (dy_dtype == MFNET_BF16 && x_dtype == MFNET_F32) LN_BWD(bf16_t, float);
  else if (dy_dtype == MFNET_F32 && x_dtype == MFNET_F32) LN_BWD(float, float);
  else if (dy_dtype == MFNET_BF16 && x_dtype == MFNET_BF16) LN_BWD(bf16_t, bf16_t);
  else if (dy_dtype == MFNET_F32 && x_dtype == MFNET_BF16) LN_BWD(float, bf16_t);
  else { mfnet_set_error("layernorm_bwd: bad dtype"); return MFNET_ERR_ARG; }
#undef LN_BWD_N
#undef LN_BWD
  MF_LAUNCH_CHECK("layernorm_bwd");
  return MFNET_OK;
}
