// HBM-bound kernels of the FPN / SE-fusion / UNetFormer decoder (unetformer_mmsam.py:24-563), all on
// channels-last maps [rows = B*H*W, C] so that every access is coalesced over channels. The dense
// contractions of the decoder (1x1 convs, ConvTranspose k2s2 as a GEMM with N = 4*C, 3x3 convs through
// im2col) run on mfnet_gemm_bf16; everything else is here:
//
//   colreduce  : per-(group, channel) sums over rows - BatchNorm statistics, BatchNorm backward sums,
//                SE / FRH average pools and their backward dot products
//   bn_*       : BatchNorm2d train / eval forward (+ReLU6 / GELU, Dropout2d channel mask) and backward
//   dwconv_*   : depthwise k x k conv (k = 3 pa / proj of FRH, k = 8 GLA proj with the reflect
//                "pad_out" row/column folded into the indexing), forward, dgrad, wgrad
//   gla_attn_* : 8x8-window, 8-head (hd 8) attention with the relative-position bias table
//   gla_pool_* : AvgPool(8,1) + AvgPool(1,8) on reflect-extended maps + the local branch
//   upfuse_*   : WF / FRH weighted fusion  fw0*skip + fw1*bilinear_x2(x), skip read through the
//                un-shuffled ConvTranspose row order (no pixel-shuffle pass ever runs)
//   up_final_* : bilinear x4 of the class maps into NCHW logits, and its adjoint
//   maxpool2_*, se_*, frh_gate_* : fpn4, SqueezeAndExcitation / SEFusion, FRH pixel+channel attention
#include "common.h"

// small clears run as kernels, not hipMemsetAsync: as a memset NODE of a captured graph a clear was seen to lose its order against
// the kernels around it on the first replay after an idle second
__global__ void zero_f32_kernel(float* __restrict__ p, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0.f;
}
#include <mutex>
#include <stdlib.h>
#include <float.h>

static inline unsigned grid_for(int64_t work_items, int64_t cap = 16384) {
  int64_t g = ceil_div64(work_items, 256);
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}
#define GS_LOOP(i, total) for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (total); i += (int64_t)gridDim.x * 256)

__device__ __forceinline__ float4 ld4(const void* p, int dt, int64_t i4) {
  if (dt == MFNET_F32) return reinterpret_cast<const float4*>(p)[i4];
  const u32x2_t w = reinterpret_cast<const u32x2_t*>(p)[i4];
  return make_float4(bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y));
}
// compile-time dtype: inside loops that keep several loads in flight the runtime switch of ld4 is a branch plus a full
// memory wait per load (the bf16 arm unpacks at once), which serialises them
template <bool BF>
__device__ __forceinline__ float4 ld4t(const void* p, int64_t i4) {
  if (!BF) return reinterpret_cast<const float4*>(p)[i4];
  const u32x2_t w = reinterpret_cast<const u32x2_t*>(p)[i4];
  return make_float4(bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y));
}
__device__ __forceinline__ void st4(void* p, int dt, int64_t i4, float4 v) {
  if (dt == MFNET_F32) { reinterpret_cast<float4*>(p)[i4] = v; return; }
  u32x2_t w;
  w.x = pack_bf16x2(v.x, v.y); w.y = pack_bf16x2(v.z, v.w);
  reinterpret_cast<u32x2_t*>(p)[i4] = w;
}
__device__ __forceinline__ float relu6f(float x) { return fminf(fmaxf(x, 0.f), 6.f); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 f4fma(float4 a, float4 b, float4 c) { return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w)); }
#define F4ZERO make_float4(0.f, 0.f, 0.f, 0.f)

// activation codes shared by bn_apply / bn_bwd
#define ACT_NONE 0
#define ACT_RELU6 1
#define ACT_GELU 2
__device__ __forceinline__ float act_f(float x, int act) { return act == ACT_RELU6 ? relu6f(x) : (act == ACT_GELU ? gelu_f(x) : x); }
__device__ __forceinline__ float act_grad_f(float x, int act) {
  return act == ACT_RELU6 ? ((x > 0.f && x < 6.f) ? 1.f : 0.f) : (act == ACT_GELU ? gelu_grad_f(x) : 1.f);
}

// =============================================================================== column reductions
// a, b: [G*P, C]; out0/out1: [G, C] fp32, ADDED into (zero first). grid = (splits, G); a block covers
// rows [split*rpb, ...) of group g; thread = (row lane, channel group of 4).
//   mode 0: out0 += sum a            out1 += sum a*a            (BatchNorm statistics; pools use out0)
//   mode 1: out0 += sum a*b                                      (SE / FRH gate gradients)
//   mode 2: BatchNorm backward: a = dy, b = x (pre-norm input, fp32); y = g*xhat+be; dy' = dy*cs*act'(y)
//           out0 += sum dy'          out1 += sum dy'*xhat
struct ColRedParams {
  const void* a; int a_dt; const void* b; int b_dt;
  float* out0; float* out1;
  int64_t P; int C; int64_t rpb; int mode;
  const float* mean; const float* rstd; const float* gamma; const float* beta; int act;
  const float* chan_scale; int64_t rows_per_image;       // optional [B, C] multiplier (Dropout2d mask)
  float* partial; unsigned int* counter;                 // workspace: [G][splits][2][C] floats, [G] zeroed tickets
};
// Finish of a column reduction: red0 / red1 hold the workgroup's per-row-lane partial sums as [nrl][C/4] float4.
__device__ __forceinline__ void colreduce_finish(const ColRedParams& p, const int64_t g, float4* red0, float4* red1, const int nrl) {
  const int C4 = p.C >> 2;
  __syncthreads();
  // Deterministic finish without float atomics: every workgroup stores its partial row sums, the one that
  // draws the last ticket of its group adds all of them in a fixed order (agent-scope hand-off:
  // write-through slab stores -> vmcnt(0) -> barrier -> relaxed ticket; the last arriver acquires, then reads
  // the slabs with plain loads). The forward pass - BatchNorm statistics, SE / FRH pools -
  // is therefore bit-reproducible, and so are the ReLU6 / max-pool decisions the backward pass follows.
  const int splits = gridDim.x;
  float* slab = p.partial + ((int64_t)g * splits + blockIdx.x) * 2 * p.C;
  if (threadIdx.x < C4) {
    const int cg = threadIdx.x;
    float4 s0 = F4ZERO, s1 = F4ZERO;
    for (int k = 0; k < nrl; ++k) { s0 = f4add(s0, red0[k * C4 + cg]); s1 = f4add(s1, red1[k * C4 + cg]); }
    // write-through (sc1) 8-byte stores: the slab needs no release fence, which would write back every dirty
    // line of this XCD's L2 - typically the map the previous kernel just produced
    typedef __attribute__((address_space(1))) unsigned long long gu64_t;
    union { float2 f; unsigned long long u; } w;
    gu64_t* d0 = (gu64_t*)(slab + 4 * cg);
    gu64_t* d1 = (gu64_t*)(slab + p.C + 4 * cg);
    w.f = make_float2(s0.x, s0.y); __hip_atomic_store(d0, w.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w.f = make_float2(s0.z, s0.w); __hip_atomic_store(d0 + 1, w.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w.f = make_float2(s1.x, s1.y); __hip_atomic_store(d1, w.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w.f = make_float2(s1.z, s1.w); __hip_atomic_store(d1 + 1, w.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int last_arriver;
  if (threadIdx.x == 0) {
    const unsigned int ticket = __hip_atomic_fetch_add(p.counter + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_arriver = ticket == (unsigned)splits - 1;
    if (last_arriver) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(p.counter + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
    }
  }
  __syncthreads();
  if (!last_arriver) return;
  // The slabs were written through to memory by other XCDs: every load here is a long-latency miss. Threads
  // read float4s, 16 in flight each, and share a column group between `parts` threads (fixed partition and
  // summation order: still deterministic), so the tail is one or a few memory round trips.
  const float* slabs = p.partial + (int64_t)g * splits * 2 * p.C;
  const int ncg = (p.out1 ? 2 * p.C : p.C) >> 2;     // float4 column groups to finish
  const int cpp = ncg < 256 ? ncg : 256;             // column groups per pass
  const int parts = 256 / cpp;
  for (int c0 = 0; c0 < ncg; c0 += cpp) {
    const int cl = threadIdx.x % cpp, part = threadIdx.x / cpp, c4 = c0 + cl;
    float4 tot = F4ZERO;
    const bool on = part < parts;                      // 256 % cpp != 0: the threads past the last whole part idle
    for (int sb = part; on && sb < splits; sb += 16 * parts) {
      float4 v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int si = sb + k * parts;
        v[k] = si < splits ? reinterpret_cast<const float4*>(slabs + (int64_t)si * 2 * p.C)[c4] : F4ZERO;
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) tot = f4add(tot, v[k]);
    }
    __syncthreads();
    if (on) red0[part * cpp + cl] = tot;
    __syncthreads();
    if (part == 0) {
      for (int k = 1; k < parts; ++k) tot = f4add(tot, red0[k * cpp + cl]);
      float* o = 4 * c4 < p.C ? p.out0 + g * p.C + 4 * c4 : p.out1 + g * p.C + 4 * c4 - p.C;
      o[0] += tot.x; o[1] += tot.y; o[2] += tot.z; o[3] += tot.w;
    }
  }
}

template <bool ABF, bool BBF>
__global__ __launch_bounds__(256) void colreduce_kernel(const ColRedParams p) {
  __shared__ float4 red0[256], red1[256];
  const int C4 = p.C >> 2;
  const int cg = threadIdx.x % C4, rl = threadIdx.x / C4, nrl = 256 / C4;
  const int64_t g = blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.x * p.rpb;
  int64_t r1 = r0 + p.rpb;
  if (r1 > p.P) r1 = p.P;
  float4 s0 = F4ZERO, s1 = F4ZERO;
  float4 mu = F4ZERO, rs = F4ZERO, ga = F4ZERO, be = F4ZERO;
  if (p.mode == 2) {
    mu = reinterpret_cast<const float4*>(p.mean)[cg]; rs = reinterpret_cast<const float4*>(p.rstd)[cg];
    ga = reinterpret_cast<const float4*>(p.gamma)[cg]; be = reinterpret_cast<const float4*>(p.beta)[cg];
  }
  if (rl < nrl) {
    // eight rows per thread are requested before any is consumed: the kernel streams the map once and is bound
    // by the bytes in flight (rows past the end re-read the last row and are discarded)
    for (int64_t rb = r0 + rl; rb < r1; rb += 8 * nrl) {
      float4 av[8], bv[8];
      float mur[8], rsr[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        int64_t r = rb + (int64_t)k * nrl;
        r = r < r1 ? r : r1 - 1;
        const int64_t row = g * p.P + r;
        av[k] = ld4t<ABF>(p.a, row * C4 + cg);
        if (p.mode != 0) bv[k] = ld4t<BBF>(p.b, row * C4 + cg);
        if (p.mode == 3) { mur[k] = p.mean[row]; rsr[k] = p.rstd[row]; }       // LayerNorm: per-ROW statistics
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int64_t r = rb + (int64_t)k * nrl;
        if (r >= r1) break;
        const float4 a = av[k];
        if (p.mode == 0) {
          s0 = f4add(s0, a); s1 = f4fma(a, a, s1);
        } else if (p.mode == 1) {
          s0 = f4fma(a, bv[k], s0);
        } else if (p.mode == 3) {
          const float4 x = bv[k];
          const float4 xh = make_float4((x.x - mur[k]) * rsr[k], (x.y - mur[k]) * rsr[k], (x.z - mur[k]) * rsr[k], (x.w - mur[k]) * rsr[k]);
          s0 = f4add(s0, a); s1 = f4fma(a, xh, s1);
        } else {
          const float4 x = bv[k];
          const float4 xh = make_float4((x.x - mu.x) * rs.x, (x.y - mu.y) * rs.y, (x.z - mu.z) * rs.z, (x.w - mu.w) * rs.w);
          float4 d = a;
          if (p.chan_scale) d = f4mul(d, reinterpret_cast<const float4*>(p.chan_scale)[((g * p.P + r) / p.rows_per_image) * C4 + cg]);
          if (p.act != ACT_NONE) {
            d.x *= act_grad_f(fmaf(ga.x, xh.x, be.x), p.act); d.y *= act_grad_f(fmaf(ga.y, xh.y, be.y), p.act);
            d.z *= act_grad_f(fmaf(ga.z, xh.z, be.z), p.act); d.w *= act_grad_f(fmaf(ga.w, xh.w, be.w), p.act);
          }
          s0 = f4add(s0, d); s1 = f4fma(d, xh, s1);
        }
      }
    }
  }
  red0[threadIdx.x] = s0; red1[threadIdx.x] = s1;
  colreduce_finish(p, g, red0, red1, nrl);
}

// bf16 maps (SE pools over the FPN levels, their gate gradients, ConvTranspose bias gradients): the same
// reduction with 16-byte loads - eight columns per thread, modes 0 (sum, sum of squares) and 1 (dot) only.
__global__ __launch_bounds__(256) void colreduce_wide_kernel(const ColRedParams p) {
  __shared__ float4 red0[512], red1[512];
  const int CG = p.C >> 3;
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG, nrl = 256 / CG;
  const int64_t g = blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.x * p.rpb;
  int64_t r1 = r0 + p.rpb;
  if (r1 > p.P) r1 = p.P;
  float s0[8], s1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
  const u32x4_t* A = reinterpret_cast<const u32x4_t*>(p.a);
  const u32x4_t* B = reinterpret_cast<const u32x4_t*>(p.b);
  for (int64_t rb = r0 + rl; rl < nrl && rb < r1; rb += 8 * nrl) {
    u32x4_t av[8], bv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int64_t r = rb + (int64_t)k * nrl;
      r = r < r1 ? r : r1 - 1;
      av[k] = A[(g * p.P + r) * CG + cg];
      if (p.mode == 1) bv[k] = B[(g * p.P + r) * CG + cg];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (rb + (int64_t)k * nrl >= r1) break;
      const unsigned int aw[4] = {av[k].x, av[k].y, av[k].z, av[k].w};
      const unsigned int bw[4] = {bv[k].x, bv[k].y, bv[k].z, bv[k].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = bf16_lo(aw[e]), hi = bf16_hi(aw[e]);
        if (p.mode == 0) {
          s0[2 * e] += lo; s0[2 * e + 1] += hi;
          s1[2 * e] = fmaf(lo, lo, s1[2 * e]); s1[2 * e + 1] = fmaf(hi, hi, s1[2 * e + 1]);
        } else {
          s0[2 * e] = fmaf(lo, bf16_lo(bw[e]), s0[2 * e]); s0[2 * e + 1] = fmaf(hi, bf16_hi(bw[e]), s0[2 * e + 1]);
        }
      }
    }
  }
  // [rl][C/4] float4 layout: this thread owns float4 columns 2*cg and 2*cg + 1 of row lane rl
  red0[2 * threadIdx.x] = make_float4(s0[0], s0[1], s0[2], s0[3]); red0[2 * threadIdx.x + 1] = make_float4(s0[4], s0[5], s0[6], s0[7]);
  red1[2 * threadIdx.x] = make_float4(s1[0], s1[1], s1[2], s1[3]); red1[2 * threadIdx.x + 1] = make_float4(s1[4], s1[5], s1[6], s1[7]);
  colreduce_finish(p, g, red0, red1, nrl);
}

static int64_t colreduce_splits(int64_t G, int64_t P, int C, int64_t* rpb_out) {
  int64_t splits = ceil_div64(512, G);
  if (splits > 128) splits = 128;                    // the last arriver reads every slab (long-latency loads)
  int64_t rpb = ceil_div64(P, splits);
  const int nrl = 256 / (C / 4);
  if (rpb < 8 * nrl) rpb = 8 * nrl;
  if (rpb_out) *rpb_out = rpb;
  return ceil_div64(P, rpb);
}
// floats of workspace for one column reduction over G groups of P rows x C columns (partial slabs)
extern "C" int64_t mfnet_colreduce_workspace_floats(int64_t G, int64_t P, int64_t C) {
  if (G <= 0 || P <= 0 || C <= 0 || C % 4 || C > 1024) return 0;
  return G * colreduce_splits(G, P, (int)C, nullptr) * 2 * C;
}
static int launch_colreduce(ColRedParams& p, int64_t G, float* workspace, unsigned int* counters, hipStream_t stream, const char* who) {
  MF_CHECK_ARG(p.C > 0 && p.C % 4 == 0 && p.C <= 1024, "%s: C=%d must be a multiple of 4, at most 1024", who, p.C);
  MF_CHECK_ARG(G > 0 && G < 65536 && p.P > 0, "%s: bad group geometry", who);
  MF_CHECK_ARG(workspace && counters, "%s: needs a workspace of mfnet_colreduce_workspace_floats(G, P, C) floats and G zeroed tickets", who);
  int64_t rpb;
  const int64_t splits = colreduce_splits(G, p.P, p.C, &rpb);
  p.rpb = rpb; p.partial = workspace; p.counter = counters;
  const bool wide = p.a_dt == MFNET_BF16 && (p.mode == 0 || (p.mode == 1 && p.b_dt == MFNET_BF16)) && p.C % 8 == 0 && ((uintptr_t)p.a % 16 == 0) &&
                    (p.mode == 0 || (uintptr_t)p.b % 16 == 0);
  if (wide) hipLaunchKernelGGL(colreduce_wide_kernel, dim3((unsigned)splits, (unsigned)G), dim3(256), 0, stream, p);
  else {
    const dim3 grid((unsigned)splits, (unsigned)G);
    const bool abf = p.a_dt == MFNET_BF16, bbf = p.mode != 0 && p.b_dt == MFNET_BF16;
    if (abf && bbf) hipLaunchKernelGGL((colreduce_kernel<true, true>), grid, dim3(256), 0, stream, p);
    else if (abf) hipLaunchKernelGGL((colreduce_kernel<true, false>), grid, dim3(256), 0, stream, p);
    else if (bbf) hipLaunchKernelGGL((colreduce_kernel<false, true>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((colreduce_kernel<false, false>), grid, dim3(256), 0, stream, p);
  }
  MF_LAUNCH_CHECK(who);
  return MFNET_OK;
}
// x [G*P, C] -> sum [G, C] (+= ) and optionally sumsq [G, C]
extern "C" int mfnet_colsum(const void* x, int x_dtype, float* sum, float* sumsq, int64_t G, int64_t P, int64_t C,
                            float* workspace, unsigned int* tickets, hipStream_t stream) {
  MF_CHECK_ARG(x && sum, "colsum: null pointer");
  ColRedParams p = {};
  p.a = x; p.a_dt = x_dtype; p.out0 = sum; p.out1 = sumsq; p.P = P; p.C = (int)C; p.mode = 0;
  return launch_colreduce(p, G, workspace, tickets, stream, "colsum");
}
// ---- mfnet_colsum for callers that cannot allocate (HIP graph capture, tight serving loops, harnesses that forbid
// hipMalloc): the same kernels, with the partial slabs and the tickets in a file-scope device array instead of a
// caller workspace. `scratch` is a host-side stand-in in host code, never a device address: its per-device address is
// taken ONCE per device with hipGetSymbolAddress and only that pointer is handed to kernels.
#define COLSUM_STATIC_MAX_FLOATS (1 << 20)     /* 4 MiB of partial slabs: G * splits * 2 * C floats */
#define COLSUM_STATIC_MAX_GROUPS 1024          /* tickets, one per group */
#define COLSUM_STATIC_MAX_DEVICES 64
static __device__ __align__(256) float scratch[COLSUM_STATIC_MAX_FLOATS + COLSUM_STATIC_MAX_GROUPS];
static float* g_scratch_ptr[COLSUM_STATIC_MAX_DEVICES];        // per device: each GPU holds its own copy at its own address
static std::mutex g_scratch_mu;

extern "C" int mfnet_colsum_static(const void* x, int x_dtype, float* sum, float* sumsq, int64_t G, int64_t P, int64_t C, hipStream_t stream) {
  MF_CHECK_ARG(x && sum, "colsum_static: null pointer");
  MF_CHECK_ARG(G > 0 && P > 0 && C > 0 && C % 4 == 0 && C <= 1024, "colsum_static: bad shape (C a multiple of 4, at most 1024)");
  const int64_t need = mfnet_colreduce_workspace_floats(G, P, C);
  MF_CHECK_ARG(G <= COLSUM_STATIC_MAX_GROUPS && need > 0 && need <= COLSUM_STATIC_MAX_FLOATS,
               "colsum_static: G=%lld P=%lld C=%lld needs %lld floats of partial sums / %lld tickets, the static scratch holds %d / %d - "
               "use mfnet_colsum with a caller workspace", (long long)G, (long long)P, (long long)C, (long long)need, (long long)G,
               COLSUM_STATIC_MAX_FLOATS, COLSUM_STATIC_MAX_GROUPS);
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= COLSUM_STATIC_MAX_DEVICES) {
    mfnet_set_error("colsum_static: no current device (or device index >= %d)", COLSUM_STATIC_MAX_DEVICES);
    return MFNET_ERR_HIP;
  }
  float* ptr;
  {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    ptr = g_scratch_ptr[dev];
    if (!ptr) {          // set-up, once per device, with that device current
      void* sym = nullptr;
      const hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));
      if (e != hipSuccess || !sym) {
        mfnet_set_error("colsum_static: hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(e));
        return MFNET_ERR_HIP;
      }
      ptr = g_scratch_ptr[dev] = (float*)sym;
    }
  }
  unsigned int* tickets = (unsigned int*)(ptr + COLSUM_STATIC_MAX_FLOATS);
  // the array keeps the last call's contents: the slabs are fully rewritten before they are read, the tickets must start at
  // zero - cleared here by a one-workgroup kernel on the caller's stream
  hipLaunchKernelGGL(zero_f32_kernel, dim3(1), dim3(256), 0, stream, (float*)tickets, (int)((G + 3) / 4 * 4));      // (0.f is all-zero bits)
  ColRedParams p = {};
  p.a = x; p.a_dt = x_dtype; p.out0 = sum; p.out1 = sumsq; p.P = P; p.C = (int)C; p.mode = 0;
  return launch_colreduce(p, G, ptr, tickets, stream, "colsum_static");
}

// out [G, C] += sum_p a[g,p,c] * b[g,p,c]
extern "C" int mfnet_coldot(const void* a, int a_dtype, const void* b, int b_dtype, float* out, int64_t G, int64_t P, int64_t C,
                            float* workspace, unsigned int* tickets, hipStream_t stream) {
  MF_CHECK_ARG(a && b && out, "coldot: null pointer");
  ColRedParams p = {};
  p.a = a; p.a_dt = a_dtype; p.b = b; p.b_dt = b_dtype; p.out0 = out; p.out1 = nullptr; p.P = P; p.C = (int)C; p.mode = 1;
  return launch_colreduce(p, G, workspace, tickets, stream, "coldot");
}

// LayerNorm weight gradients (nn.LayerNorm / Norm2d over the channel axis): dbeta [C] += sum_rows dy, dgamma [C] += sum_rows
// dy * (x - mean[row]) * rstd[row]. One deterministic pass instead of float atomics from every wave of the dx kernel
// (which serialised ~2000 adds per address: 0.2 ms for the FPN's Norm2d).
extern "C" int mfnet_layernorm_bwd_weights(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* rstd,
                                           float* dgamma, float* dbeta, int64_t rows, int64_t C, float* workspace, unsigned int* tickets,
                                           hipStream_t stream) {
  MF_CHECK_ARG(dy && x && mean && rstd && dgamma && dbeta, "layernorm_bwd_weights: null pointer");
  ColRedParams p = {};
  p.a = dy; p.a_dt = dy_dtype; p.b = x; p.b_dt = x_dtype; p.mean = mean; p.rstd = rstd; p.out0 = dbeta; p.out1 = dgamma;
  p.P = rows; p.C = (int)C; p.mode = 3;
  return launch_colreduce(p, 1, workspace, tickets, stream, "layernorm_bwd_weights");
}

// =============================================================================== BatchNorm2d
// finalize: training: mean/var from the sums (biased var for normalisation), running stats updated with the
// unbiased var (nn.BatchNorm2d, momentum 0.1), num_batches_tracked += 1. eval: mean/rstd from running stats.
__global__ void bn_finalize_kernel(const float* sum, const float* sumsq, float* mean, float* rstd, float* run_mean,
                                   float* run_var, int64_t* nbt, int C, float n, float eps, float momentum, int training) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  if (training) {
    const float m = sum[c] / n;
    float v = sumsq[c] / n - m * m;
    v = v > 0.f ? v : 0.f;
    mean[c] = m; rstd[c] = rsqrtf(v + eps);
    if (run_mean) {
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * m;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (n > 1.f ? v * n / (n - 1.f) : v);
    }
    if (nbt && c == 0) nbt[0] += 1;
  } else {
    mean[c] = run_mean[c]; rstd[c] = rsqrtf(run_var[c] + eps);
  }
}
extern "C" int mfnet_bn_finalize(const float* sum, const float* sumsq, float* mean, float* rstd, float* run_mean, float* run_var,
                                 int64_t* num_batches_tracked, int64_t C, int64_t n_rows, float eps, float momentum,
                                 int training, hipStream_t stream) {
  MF_CHECK_ARG(mean && rstd && C > 0 && (training ? (sum && sumsq && n_rows > 0) : (run_mean && run_var)), "bn_finalize: bad args");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)ceil_div64(C, 256)), dim3(256), 0, stream, sum, sumsq, mean, rstd, run_mean,
                     run_var, num_batches_tracked, (int)C, (float)n_rows, eps, momentum, training);
  MF_LAUNCH_CHECK("bn_finalize");
  return MFNET_OK;
}
// y = act((x - mean) * rstd * gamma + beta) * chan_scale[row / rows_per_image]; x [rows, C] fp32|bf16
__global__ __launch_bounds__(256) void bn_apply_kernel(const void* __restrict__ x, int x_dt, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int act, const float* __restrict__ cs,
                                                       int64_t rpi, void* __restrict__ y, int y_dt, void* __restrict__ y2, int y2_dt,
                                                       int64_t rows, int C4) {
  const int64_t total = rows * C4;
  GS_LOOP(i, total) {
    const int cg = (int)(i % C4);
    const int64_t row = i / C4;
    const float4 v = ld4(x, x_dt, i);
    const float4 mu = reinterpret_cast<const float4*>(mean)[cg], rs = reinterpret_cast<const float4*>(rstd)[cg];
    const float4 ga = reinterpret_cast<const float4*>(gamma)[cg], be = reinterpret_cast<const float4*>(beta)[cg];
    float4 o = make_float4(act_f(fmaf((v.x - mu.x) * rs.x, ga.x, be.x), act), act_f(fmaf((v.y - mu.y) * rs.y, ga.y, be.y), act),
                           act_f(fmaf((v.z - mu.z) * rs.z, ga.z, be.z), act), act_f(fmaf((v.w - mu.w) * rs.w, ga.w, be.w), act));
    if (cs) o = f4mul(o, reinterpret_cast<const float4*>(cs)[(row / rpi) * C4 + cg]);
    st4(y, y_dt, i, o);
    if (y2) st4(y2, y2_dt, i, o);
  }
}
extern "C" int mfnet_bn_apply(const void* x, int x_dtype, const float* mean, const float* rstd, const float* gamma, const float* beta,
                              int act, const float* chan_scale, int64_t rows_per_image, void* y, int y_dtype, void* y2,
                              int y2_dtype, int64_t rows, int64_t C, hipStream_t stream) {
  MF_CHECK_ARG(x && mean && rstd && gamma && beta && y && C > 0 && C % 4 == 0 && act >= 0 && act <= 2, "bn_apply: bad args");
  MF_CHECK_ARG(!chan_scale || rows_per_image > 0, "bn_apply: chan_scale needs rows_per_image");
  if (rows <= 0) return MFNET_OK;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(rows * C / 4)), dim3(256), 0, stream, x, x_dtype, mean, rstd, gamma, beta, act,
                     chan_scale, rows_per_image, y, y_dtype, y2, y2_dtype, rows, (int)(C / 4));
  MF_LAUNCH_CHECK("bn_apply");
  return MFNET_OK;
}
// Training forward in one launch: every workgroup derives mean / rstd of all channels from the column sums (the arithmetic of
// bn_finalize_kernel, kept in LDS), workgroup 0 also stores them for the backward pass and updates the running statistics.
#define BN_MAXC 1024
__global__ __launch_bounds__(256) void bn_train_apply_kernel(const void* __restrict__ x, int x_dt, const float* __restrict__ sum, const float* __restrict__ sumsq,
                                                             float n, float eps, float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                             float* __restrict__ run_mean, float* __restrict__ run_var, int64_t* __restrict__ nbt,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                             const float* __restrict__ cs, int64_t rpi, void* __restrict__ y, int y_dt,
                                                             void* __restrict__ y2, int y2_dt, int64_t rows, int C4) {
  __shared__ __attribute__((aligned(16))) float mu_s[BN_MAXC], rs_s[BN_MAXC];
  const int C = 4 * C4;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float m = sum[c] / n;
    float v = sumsq[c] / n - m * m;
    v = v > 0.f ? v : 0.f;
    const float r = rsqrtf(v + eps);
    mu_s[c] = m; rs_s[c] = r;
    if (blockIdx.x == 0) {
      mean[c] = m; rstd[c] = r;
      if (run_mean) {
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * m;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (n > 1.f ? v * n / (n - 1.f) : v);
      }
      if (nbt && c == 0) nbt[0] += 1;
    }
  }
  __syncthreads();
  const int64_t total = rows * C4;
  GS_LOOP(i, total) {
    const int cg = (int)(i % C4);
    const int64_t row = i / C4;
    const float4 v = ld4(x, x_dt, i);
    const float4 mu = reinterpret_cast<const float4*>(mu_s)[cg], rs = reinterpret_cast<const float4*>(rs_s)[cg];
    const float4 ga = reinterpret_cast<const float4*>(gamma)[cg], be = reinterpret_cast<const float4*>(beta)[cg];
    float4 o = make_float4(act_f(fmaf((v.x - mu.x) * rs.x, ga.x, be.x), act), act_f(fmaf((v.y - mu.y) * rs.y, ga.y, be.y), act),
                           act_f(fmaf((v.z - mu.z) * rs.z, ga.z, be.z), act), act_f(fmaf((v.w - mu.w) * rs.w, ga.w, be.w), act));
    if (cs) o = f4mul(o, reinterpret_cast<const float4*>(cs)[(row / rpi) * C4 + cg]);
    st4(y, y_dt, i, o);
    if (y2) st4(y2, y2_dt, i, o);
  }
}
extern "C" int mfnet_bn_train_apply(const void* x, int x_dtype, const float* sum, const float* sumsq, float* mean, float* rstd, float* run_mean,
                                    float* run_var, int64_t* num_batches_tracked, float eps, float momentum, const float* gamma,
                                    const float* beta, int act, const float* chan_scale, int64_t rows_per_image, void* y, int y_dtype, void* y2,
                                    int y2_dtype, int64_t rows, int64_t C, hipStream_t stream) {
  MF_CHECK_ARG(x && sum && sumsq && mean && rstd && gamma && beta && y && C > 0 && C % 4 == 0 && C <= BN_MAXC && act >= 0 && act <= 2 && rows > 0,
               "bn_train_apply: bad args (C a multiple of 4, at most %d; rows > 0)", BN_MAXC);
  MF_CHECK_ARG((run_mean == nullptr) == (run_var == nullptr), "bn_train_apply: running mean and variance come together");
  MF_CHECK_ARG(!chan_scale || rows_per_image > 0, "bn_train_apply: chan_scale needs rows_per_image");
  hipLaunchKernelGGL(bn_train_apply_kernel, dim3(grid_for(rows * C / 4)), dim3(256), 0, stream, x, x_dtype, sum, sumsq, (float)rows, eps, momentum,
                     mean, rstd, run_mean, run_var, num_batches_tracked, gamma, beta, act, chan_scale, rows_per_image, y, y_dtype, y2, y2_dtype,
                     rows, (int)(C / 4));
  MF_LAUNCH_CHECK("bn_train_apply");
  return MFNET_OK;
}
// sums for the backward: dsum[c] += sum dy', dxh[c] += sum dy'*xhat  (= dbeta, dgamma)
extern "C" int mfnet_bn_bwd_reduce(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, int act, const float* chan_scale, int64_t rows_per_image,
                                   float* dsum, float* dxh, int64_t rows, int64_t C, float* workspace, unsigned int* tickets,
                                   hipStream_t stream) {
  MF_CHECK_ARG(dy && x && mean && rstd && gamma && beta && dsum && dxh && act >= 0 && act <= 2, "bn_bwd_reduce: bad args");
  ColRedParams p = {};
  p.a = dy; p.a_dt = dy_dtype; p.b = x; p.b_dt = x_dtype; p.out0 = dsum; p.out1 = dxh; p.P = rows; p.C = (int)C; p.mode = 2;
  p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.beta = beta; p.act = act; p.chan_scale = chan_scale; p.rows_per_image = rows_per_image;
  return launch_colreduce(p, 1, workspace, tickets, stream, "bn_bwd_reduce");
}
// dx = gamma*rstd*(dy' - [training](dsum/n + xhat*dxh/n))
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const void* __restrict__ dy, int dy_dt, const void* __restrict__ x, int x_dt,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                           const float* __restrict__ cs, int64_t rpi, const float* __restrict__ dsum,
                                                           const float* __restrict__ dxh, float inv_n, void* __restrict__ dx, int dx_dt,
                                                           int64_t rows, int C4) {
  const int64_t total = rows * C4;
  GS_LOOP(i, total) {
    const int cg = (int)(i % C4);
    const int64_t row = i / C4;
    const float4 v = ld4(x, x_dt, i);
    float4 d = ld4(dy, dy_dt, i);
    const float4 mu = reinterpret_cast<const float4*>(mean)[cg], rs = reinterpret_cast<const float4*>(rstd)[cg];
    const float4 ga = reinterpret_cast<const float4*>(gamma)[cg], be = reinterpret_cast<const float4*>(beta)[cg];
    const float4 xh = make_float4((v.x - mu.x) * rs.x, (v.y - mu.y) * rs.y, (v.z - mu.z) * rs.z, (v.w - mu.w) * rs.w);
    if (cs) d = f4mul(d, reinterpret_cast<const float4*>(cs)[(row / rpi) * C4 + cg]);
    if (act != ACT_NONE) {
      d.x *= act_grad_f(fmaf(ga.x, xh.x, be.x), act); d.y *= act_grad_f(fmaf(ga.y, xh.y, be.y), act);
      d.z *= act_grad_f(fmaf(ga.z, xh.z, be.z), act); d.w *= act_grad_f(fmaf(ga.w, xh.w, be.w), act);
    }
    if (inv_n > 0.f) {
      const float4 s = reinterpret_cast<const float4*>(dsum)[cg], t = reinterpret_cast<const float4*>(dxh)[cg];
      d.x -= (s.x + xh.x * t.x) * inv_n; d.y -= (s.y + xh.y * t.y) * inv_n;
      d.z -= (s.z + xh.z * t.z) * inv_n; d.w -= (s.w + xh.w * t.w) * inv_n;
    }
    st4(dx, dx_dt, i, make_float4(d.x * ga.x * rs.x, d.y * ga.y * rs.y, d.z * ga.z * rs.z, d.w * ga.w * rs.w));
  }
}
extern "C" int mfnet_bn_bwd_apply(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* rstd,
                                  const float* gamma, const float* beta, int act, const float* chan_scale, int64_t rows_per_image,
                                  const float* dsum, const float* dxh, int training, void* dx, int dx_dtype, int64_t rows,
                                  int64_t C, hipStream_t stream) {
  MF_CHECK_ARG(dy && x && mean && rstd && gamma && beta && dx && C % 4 == 0 && (!training || (dsum && dxh)), "bn_bwd_apply: bad args");
  if (rows <= 0) return MFNET_OK;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(rows * C / 4)), dim3(256), 0, stream, dy, dy_dtype, x, x_dtype, mean, rstd, gamma,
                     beta, act, chan_scale, rows_per_image, dsum, dxh, training ? 1.f / (float)rows : 0.f, dx, dx_dtype, rows, (int)(C / 4));
  MF_LAUNCH_CHECK("bn_bwd_apply");
  return MFNET_OK;
}

// =============================================================================== elementwise activation (bf16 / fp32)
__global__ __launch_bounds__(256) void act_kernel(const void* __restrict__ x, int x_dt, void* __restrict__ y, int y_dt, int act, int64_t n4) {
  GS_LOOP(i, n4) {
    const float4 v = ld4(x, x_dt, i);
    st4(y, y_dt, i, make_float4(act_f(v.x, act), act_f(v.y, act), act_f(v.z, act), act_f(v.w, act)));
  }
}
extern "C" int mfnet_act(const void* x, int x_dtype, void* y, int y_dtype, int act, int64_t n, hipStream_t stream) {
  MF_CHECK_ARG(x && y && n % 4 == 0 && act >= 0 && act <= 2, "act: bad args");
  if (n <= 0) return MFNET_OK;
  hipLaunchKernelGGL(act_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, x, x_dtype, y, y_dtype, act, n / 4);
  MF_LAUNCH_CHECK("act");
  return MFNET_OK;
}

// dx = dy * act'(ref); for ReLU6 `ref` may be the activation's input or its output (same mask 0 < ref < 6)
__global__ __launch_bounds__(256) void act_bwd_kernel(const void* __restrict__ dy, int dy_dt, const void* __restrict__ ref, int ref_dt, void* __restrict__ dx,
                                                      int dx_dt, int act, int64_t n4) {
  GS_LOOP(i, n4) {
    const float4 d = ld4(dy, dy_dt, i), r = ld4(ref, ref_dt, i);
    st4(dx, dx_dt, i, make_float4(d.x * act_grad_f(r.x, act), d.y * act_grad_f(r.y, act), d.z * act_grad_f(r.z, act), d.w * act_grad_f(r.w, act)));
  }
}
extern "C" int mfnet_act_bwd(const void* dy, int dy_dtype, const void* ref, int ref_dtype, void* dx, int dx_dtype, int act, int64_t n, hipStream_t stream) {
  MF_CHECK_ARG(dy && ref && dx && n % 4 == 0 && act >= 0 && act <= 2, "act_bwd: bad args");
  if (n <= 0) return MFNET_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, dy, dy_dtype, ref, ref_dtype, dx, dx_dtype, act, n / 4);
  MF_LAUNCH_CHECK("act_bwd");
  return MFNET_OK;
}

// =============================================================================== depthwise conv
// x [B,H,W,C] fp32, w [C, k*k] (nn.Conv2d weight [C,1,k,k]), pad = (k-1)/2 zeros. ext = 1: the input is first
// extended by one reflected row / column at the bottom / right (F.pad(.., (0,1,0,1), 'reflect'),
// unetformer_mmsam.py:192-196) and the output cropped back to H x W: ext coordinate H -> H-2, W -> W-2.
struct DwParams { int B, H, W, C, k, pad, ext; };
__device__ __forceinline__ int ext_map(int r, int n, int ext) {   // -> source index or -1 (zero)
  if (r < 0) return -1;
  if (r < n) return r;
  return (ext && r == n) ? n - 2 : -1;
}
// K = 3 / 8: the taps of one kernel row are loaded together (clamped address, masked after the load) - with a runtime k and
// `continue` around each tap the loop was one dependent L2 round trip per tap (64 of them for the GLA 8x8 kernel: 24 us on a
// 64 KB map). K = 0: any k, the plain loop.
template <int K>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ y, const DwParams p) {
  extern __shared__ float wl[];      // [k*k][C]
  const int k = K ? K : p.k, kk = k * k;
  for (int i = threadIdx.x; i < kk * p.C; i += 256) { const int tap = i / p.C, c = i - tap * p.C; wl[i] = w[c * kk + tap]; }
  __syncthreads();
  const int C4 = p.C >> 2;
  const int64_t total = (int64_t)p.B * p.H * p.W * C4;
  GS_LOOP(i, total) {
    const int cg = (int)(i % C4);
    const int64_t pix = i / C4;
    const int xx = (int)(pix % p.W), yy = (int)((pix / p.W) % p.H);
    const int64_t b = pix / ((int64_t)p.W * p.H);
    float4 acc = bias ? reinterpret_cast<const float4*>(bias)[cg] : F4ZERO;
    if (K) {
      constexpr int KU = K ? K : 1;
#pragma unroll 2
      for (int ky = 0; ky < KU; ++ky) {
        const int sy = ext_map(yy + ky - p.pad, p.H, p.ext);
        const int64_t rowb = (b * p.H + (sy < 0 ? 0 : sy)) * p.W;
        float4 v[KU];
        bool ok[KU];
#pragma unroll
        for (int kx = 0; kx < KU; ++kx) {
          const int sx = ext_map(xx + kx - p.pad, p.W, p.ext);
          ok[kx] = sy >= 0 && sx >= 0;
          v[kx] = reinterpret_cast<const float4*>(x)[(rowb + (sx < 0 ? 0 : sx)) * C4 + cg];
        }
#pragma unroll
        for (int kx = 0; kx < KU; ++kx) {
          const float4 wv = reinterpret_cast<const float4*>(wl)[(ky * KU + kx) * C4 + cg];
          if (ok[kx]) acc = f4fma(v[kx], wv, acc);
        }
      }
    } else {
      for (int ky = 0; ky < k; ++ky) {
        const int sy = ext_map(yy + ky - p.pad, p.H, p.ext);
        if (sy < 0) continue;
        for (int kx = 0; kx < k; ++kx) {
          const int sx = ext_map(xx + kx - p.pad, p.W, p.ext);
          if (sx < 0) continue;
          const float4 v = reinterpret_cast<const float4*>(x)[((b * p.H + sy) * p.W + sx) * C4 + cg];
          acc = f4fma(v, reinterpret_cast<const float4*>(wl)[(ky * k + kx) * C4 + cg], acc);
        }
      }
    }
    reinterpret_cast<float4*>(y)[i] = acc;
  }
}
// dx[y,x] = sum over the ext positions e that map to (y,x) of  sum_taps dy[e - tap + pad] * w[tap]
template <bool DY_BF16, int K>
__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(const void* __restrict__ dy, int dy_dt, const float* __restrict__ w,
                                                              float* __restrict__ dx, const DwParams p) {
  extern __shared__ float wl[];
  const int k = K ? K : p.k, kk = k * k;
  for (int i = threadIdx.x; i < kk * p.C; i += 256) { const int tap = i / p.C, c = i - tap * p.C; wl[i] = w[c * kk + tap]; }
  __syncthreads();
  const int C4 = p.C >> 2;
  const int64_t total = (int64_t)p.B * p.H * p.W * C4;
  GS_LOOP(i, total) {
    const int cg = (int)(i % C4);
    const int64_t pix = i / C4;
    const int xx = (int)(pix % p.W), yy = (int)((pix / p.W) % p.H);
    const int64_t b = pix / ((int64_t)p.W * p.H);
    float4 acc = F4ZERO;
    const int ny = (p.ext && yy == p.H - 2) ? 2 : 1, nx = (p.ext && xx == p.W - 2) ? 2 : 1;
    for (int ey = 0; ey < ny; ++ey) {
      const int py = ey ? p.H : yy;
      for (int ex = 0; ex < nx; ++ex) {
        const int px = ex ? p.W : xx;
        if (K) {
          constexpr int KU = K ? K : 1;
#pragma unroll 2
          for (int ky = 0; ky < KU; ++ky) {
            const int oy = py - ky + p.pad;
            const bool oky = oy >= 0 && oy < p.H;
            const int64_t rowb = (b * p.H + (oky ? oy : 0)) * p.W;
            float4 v[KU];
            bool ok[KU];
#pragma unroll
            for (int kx = 0; kx < KU; ++kx) {
              const int ox = px - kx + p.pad;
              const bool okx = ox >= 0 && ox < p.W;
              ok[kx] = oky && okx;
              v[kx] = ld4t<DY_BF16>(dy, (rowb + (okx ? ox : 0)) * C4 + cg);
            }
#pragma unroll
            for (int kx = 0; kx < KU; ++kx) {
              const float4 wv = reinterpret_cast<const float4*>(wl)[(ky * KU + kx) * C4 + cg];
              if (ok[kx]) acc = f4fma(v[kx], wv, acc);
            }
          }
        } else {
          for (int ky = 0; ky < k; ++ky) {
            const int oy = py - ky + p.pad;
            if (oy < 0 || oy >= p.H) continue;
            for (int kx = 0; kx < k; ++kx) {
              const int ox = px - kx + p.pad;
              if (ox < 0 || ox >= p.W) continue;
              acc = f4fma(ld4t<DY_BF16>(dy, ((b * p.H + oy) * p.W + ox) * C4 + cg), reinterpret_cast<const float4*>(wl)[(ky * k + kx) * C4 + cg], acc);
            }
          }
        }
      }
    }
    reinterpret_cast<float4*>(dx)[i] = acc;
  }
}
// dw[c, tap] += sum_pixels dy[pix, c] * x_ext[pix + tap - pad, c]; thread = (tap slot, channel group); block = chunk of pixels
#define DW_PU 4
template <bool DY_BF16>      // the gradient's dtype is a template argument: a runtime switch around each load put a branch and a
                             // full memory wait between the loads of one iteration
__device__ __forceinline__ void dw_tap_body(const void* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw,
                                            float* __restrict__ dbias, const DwParams& p, int64_t pix_per_block, int slots, int bid) {
  const int C4 = p.C >> 2, kk = p.k * p.k;
  const int cg = threadIdx.x % C4, slot = threadIdx.x / C4, nslot = 256 / C4;
  const int64_t npix = (int64_t)p.B * p.H * p.W;
  const int64_t p0 = (int64_t)bid * pix_per_block;
  int64_t p1 = p0 + pix_per_block;
  if (p1 > npix) p1 = npix;
  for (int tap = slot; tap < kk; tap += nslot) {
    const int ky = tap / p.k, kx = tap - ky * p.k;
    float4 acc = F4ZERO, bacc = F4ZERO;
    // DW_PU pixels per iteration, their 2 * DW_PU loads issued together (unconditionally: a pixel past the chunk or a tap
    // that falls into the zero padding reads element 0 and is masked): the loop was one dependent round trip per pixel
    // pixel coordinates advance by increments: the three 64-bit divisions per pixel this loop started with were most of its time
    int xx = (int)(p0 % p.W), yy = (int)((p0 / p.W) % p.H);
    int64_t rowb = (p0 / p.W) - yy;                  // b * H
    for (int64_t pix = p0; pix < p1; pix += DW_PU) {
      float4 d[DW_PU], xv[DW_PU];
      bool ok[DW_PU];
#pragma unroll
      for (int u = 0; u < DW_PU; ++u) {
        const bool in = pix + u < p1;
        const int sy = ext_map(yy + ky - p.pad, p.H, p.ext), sx = ext_map(xx + kx - p.pad, p.W, p.ext);
        ok[u] = in && sy >= 0 && sx >= 0;
        const int64_t di = (in ? pix + u : p0) * C4 + cg;
        if (DY_BF16) {
          const u32x2_t w2 = reinterpret_cast<const u32x2_t*>(dy)[di];
          d[u] = make_float4(bf16_lo(w2.x), bf16_hi(w2.x), bf16_lo(w2.y), bf16_hi(w2.y));
        } else {
          d[u] = reinterpret_cast<const float4*>(dy)[di];
        }
        if (!in) d[u] = F4ZERO;
        xv[u] = reinterpret_cast<const float4*>(x)[ok[u] ? ((rowb + sy) * p.W + sx) * C4 + cg : cg];
        if (++xx == p.W) { xx = 0; if (++yy == p.H) { yy = 0; rowb += p.H; } }
      }
#pragma unroll
      for (int u = 0; u < DW_PU; ++u) {
        if (tap == 0) bacc = f4add(bacc, d[u]);
        if (ok[u]) acc = f4fma(d[u], xv[u], acc);
      }
    }
    // same-address float atomics cost ~140 ns each on this chip: the workgroups spread over `slots` copies of dw / dbias
    // that the caller sums (512 workgroups on one copy: 70 of the kernel's 110 us)
    const int sl = bid % slots;
    float* o = dw + ((int64_t)sl * p.C + 4 * cg) * kk + tap;
    atomicAdd(o, acc.x); atomicAdd(o + kk, acc.y); atomicAdd(o + 2 * kk, acc.z); atomicAdd(o + 3 * kk, acc.w);
    if (dbias && tap == 0) {
      float* ob = dbias + (int64_t)sl * p.C;
      atomicAdd(ob + 4 * cg, bacc.x); atomicAdd(ob + 4 * cg + 1, bacc.y); atomicAdd(ob + 4 * cg + 2, bacc.z); atomicAdd(ob + 4 * cg + 3, bacc.w);
    }
  }
}
template <bool DY_BF16>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const void* __restrict__ dy, int dy_dt, const float* __restrict__ x,
                                                                float* __restrict__ dw, float* __restrict__ dbias, const DwParams p,
                                                                int64_t pix_per_block, int slots) {
  dw_tap_body<DY_BF16>(dy, x, dw, dbias, p, pix_per_block, slots, blockIdx.x);
}
// K = 3 / 8: thread = (pixel slot, channel group) instead of (tap slot, channel group). Above, every tap slot streams the whole
// chunk of dy again (k*k reads of dy and of x per pixel: 0.5 GB through L2 for the 8x8 kernel on the 32x32 maps, 59 us); here a
// thread reads dy of a pixel once per pass of TG taps (one kernel row for K = 8, all nine taps for K = 3), holds TG float4
// sums, and the 256 / C4 pixel slots of the workgroup meet in LDS before the (unchanged number of) atomics.
template <bool DY_BF16, int K>
__device__ __forceinline__ void dw_px_body(float4* red, const void* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw,
                                           float* __restrict__ dbias, const DwParams& p, int64_t pix_per_block, int slots, int bid) {
  constexpr int TG = K == 3 ? 9 : K, NPASS = K * K / TG;      // red: [tap of the pass | bias][pixel slot][channel group], (TG + 1) * 256 float4
  const int C4 = p.C >> 2, kk = K * K;
  const int cg = threadIdx.x % C4, ps = threadIdx.x / C4, nps = 256 / C4;
  const int64_t npix = (int64_t)p.B * p.H * p.W;
  const int64_t p0 = (int64_t)bid * pix_per_block;
  int64_t p1 = p0 + pix_per_block;
  if (p1 > npix) p1 = npix;
  const int sl = bid % slots;
  for (int pass = 0; pass < NPASS; ++pass) {
    float4 acc[TG], bacc = F4ZERO;
#pragma unroll
    for (int t = 0; t < TG; ++t) acc[t] = F4ZERO;
    for (int64_t pix = p0 + ps; pix < p1; pix += nps) {
      const int xx = (int)(pix % p.W), yy = (int)((pix / p.W) % p.H);
      const int64_t rowb = (pix / p.W) - yy;               // b * H
      float4 d;
      if (DY_BF16) {
        const u32x2_t w2 = reinterpret_cast<const u32x2_t*>(dy)[pix * C4 + cg];
        d = make_float4(bf16_lo(w2.x), bf16_hi(w2.x), bf16_lo(w2.y), bf16_hi(w2.y));
      } else {
        d = reinterpret_cast<const float4*>(dy)[pix * C4 + cg];
      }
      float4 xv[TG];
      bool ok[TG];
#pragma unroll
      for (int t = 0; t < TG; ++t) {
        const int tap = pass * TG + t, ky = tap / K, kx = tap - ky * K;
        const int sy = ext_map(yy + ky - p.pad, p.H, p.ext), sx = ext_map(xx + kx - p.pad, p.W, p.ext);
        ok[t] = sy >= 0 && sx >= 0;
        xv[t] = reinterpret_cast<const float4*>(x)[ok[t] ? ((rowb + sy) * p.W + sx) * C4 + cg : cg];
      }
#pragma unroll
      for (int t = 0; t < TG; ++t)
        if (ok[t]) acc[t] = f4fma(d, xv[t], acc[t]);
      if (pass == 0) bacc = f4add(bacc, d);
    }
    __syncthreads();                                        // (the previous pass's sums have been read)
#pragma unroll
    for (int t = 0; t < TG; ++t) red[t * 256 + threadIdx.x] = acc[t];
    if (pass == 0) red[TG * 256 + threadIdx.x] = bacc;
    __syncthreads();
    // thread -> (entry e of the pass, channel group): sums the pixel slots in a fixed order
    const int nent = TG + ((pass == 0 && dbias) ? 1 : 0);
    for (int e = ps; e < nent; e += nps) {
      float4 v = F4ZERO;
      for (int q = 0; q < nps; ++q) v = f4add(v, red[e * 256 + q * C4 + cg]);
      if (e < TG) {
        float* o = dw + ((int64_t)sl * p.C + 4 * cg) * kk + pass * TG + e;
        atomicAdd(o, v.x); atomicAdd(o + kk, v.y); atomicAdd(o + 2 * kk, v.z); atomicAdd(o + 3 * kk, v.w);
      } else {
        float* ob = dbias + (int64_t)sl * p.C + 4 * cg;
        atomicAdd(ob, v.x); atomicAdd(ob + 1, v.y); atomicAdd(ob + 2, v.z); atomicAdd(ob + 3, v.w);
      }
    }
  }
}
template <bool DY_BF16, int K>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_px_kernel(const void* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw,
                                                                   float* __restrict__ dbias, const DwParams p, int64_t pix_per_block, int slots) {
  __shared__ float4 red[10 * 256];
  dw_px_body<DY_BF16, K>(red, dy, x, dw, dbias, p, pix_per_block, slots, blockIdx.x);
}
// Several depthwise weight gradients in one launch (they feed nothing but the gradient delivery: collected by the host like the
// weight-gradient GEMMs, mfnet_gemm_tn_batch). kind: 0 tap slots, 1 pixel slots k = 3, 2 pixel slots k = 8.
#define DW_BATCH 16
struct DwProb { const void* dy; const float* x; float* dw; float* dbias; DwParams p; int slots; int64_t ppb; int kind, bf16, wg_begin, pad_; };
struct DwBatch { int n; DwProb q[DW_BATCH]; };
__global__ __launch_bounds__(256) void dwconv_bwd_weight_batch_kernel(const DwBatch b) {
  __shared__ float4 red[10 * 256];
  const int id = blockIdx.x;
  int i = 0;
  for (int j = 1; j < b.n; ++j)
    if (id >= b.q[j].wg_begin) i = j;
  const DwProb& q = b.q[i];
  const int bid = id - q.wg_begin;
  if (q.kind == 0) { if (q.bf16) dw_tap_body<true>(q.dy, q.x, q.dw, q.dbias, q.p, q.ppb, q.slots, bid); else dw_tap_body<false>(q.dy, q.x, q.dw, q.dbias, q.p, q.ppb, q.slots, bid); }
  else if (q.kind == 1) { if (q.bf16) dw_px_body<true, 3>(red, q.dy, q.x, q.dw, q.dbias, q.p, q.ppb, q.slots, bid); else dw_px_body<false, 3>(red, q.dy, q.x, q.dw, q.dbias, q.p, q.ppb, q.slots, bid); }
  else { if (q.bf16) dw_px_body<true, 8>(red, q.dy, q.x, q.dw, q.dbias, q.p, q.ppb, q.slots, bid); else dw_px_body<false, 8>(red, q.dy, q.x, q.dw, q.dbias, q.p, q.ppb, q.slots, bid); }
}
static int dw_check(DwParams* p, int64_t B, int64_t H, int64_t W, int64_t C, int64_t k, int ext, const char* who) {
  MF_CHECK_ARG(B > 0 && H > 1 && W > 1 && C > 0 && C % 4 == 0 && C <= 256 && 256 % (C / 4) == 0, "%s: bad geometry", who);
  MF_CHECK_ARG(k >= 1 && k <= 9 && (size_t)(k * k * C * 4) <= 64 * 1024, "%s: kernel size %lld unsupported", who, (long long)k);
  p->B = (int)B; p->H = (int)H; p->W = (int)W; p->C = (int)C; p->k = (int)k; p->pad = (int)((k - 1) / 2); p->ext = ext ? 1 : 0;
  return MFNET_OK;
}
extern "C" int mfnet_dwconv_fwd(const float* x, const float* w, const float* bias, float* y, int64_t B, int64_t H, int64_t W, int64_t C,
                                int64_t k, int ext, hipStream_t stream) {
  MF_CHECK_ARG(x && w && y, "dwconv_fwd: null pointer");
  DwParams p;
  int rc = dw_check(&p, B, H, W, C, k, ext, "dwconv_fwd");
  if (rc) return rc;
  const dim3 grid(grid_for(B * H * W * C / 4));
  const size_t lds = (size_t)(k * k * C * 4);
  if (k == 8) hipLaunchKernelGGL(dwconv_fwd_kernel<8>, grid, dim3(256), lds, stream, x, w, bias, y, p);
  else if (k == 3) hipLaunchKernelGGL(dwconv_fwd_kernel<3>, grid, dim3(256), lds, stream, x, w, bias, y, p);
  else hipLaunchKernelGGL(dwconv_fwd_kernel<0>, grid, dim3(256), lds, stream, x, w, bias, y, p);
  MF_LAUNCH_CHECK("dwconv_fwd");
  return MFNET_OK;
}
extern "C" int mfnet_dwconv_bwd_data(const void* dy, int dy_dtype, const float* w, float* dx, int64_t B, int64_t H, int64_t W, int64_t C,
                                     int64_t k, int ext, hipStream_t stream) {
  MF_CHECK_ARG(dy && w && dx, "dwconv_bwd_data: null pointer");
  DwParams p;
  int rc = dw_check(&p, B, H, W, C, k, ext, "dwconv_bwd_data");
  if (rc) return rc;
  const dim3 grid(grid_for(B * H * W * C / 4));
  const size_t lds = (size_t)(k * k * C * 4);
#define DW_BWD_DATA(BF) do {                                                                                              \
    if (k == 8) hipLaunchKernelGGL((dwconv_bwd_data_kernel<BF, 8>), grid, dim3(256), lds, stream, dy, dy_dtype, w, dx, p);      \
    else if (k == 3) hipLaunchKernelGGL((dwconv_bwd_data_kernel<BF, 3>), grid, dim3(256), lds, stream, dy, dy_dtype, w, dx, p); \
    else hipLaunchKernelGGL((dwconv_bwd_data_kernel<BF, 0>), grid, dim3(256), lds, stream, dy, dy_dtype, w, dx, p);             \
  } while (0)
  if (dy_dtype == MFNET_BF16) DW_BWD_DATA(true); else DW_BWD_DATA(false);
#undef DW_BWD_DATA
  MF_LAUNCH_CHECK("dwconv_bwd_data");
  return MFNET_OK;
}
// dw [slots][C, k*k] and dbias [slots][C] (optional) are ADDED into; the gradient is the sum over the slots.
// kernel variant, pixels per workgroup and workgroup count of one depthwise weight gradient
static void dw_weight_geometry(int64_t npix, int64_t k, int* kind, int64_t* ppb, int64_t* nwg) {
  *ppb = ceil_div64(npix, 1024);
  if (*ppb < 16) *ppb = 16;
  *kind = 0;
  if ((k == 3 || k == 8) && npix >= 4096) {
    *ppb = (ceil_div64(npix, 256) + 15) / 16 * 16;      // one workgroup per CU (see the launcher below)
    *kind = k == 3 ? 1 : 2;
  }
  *nwg = ceil_div64(npix, *ppb);
}
// desc: 12 int64 per problem - dy, dy_dtype, x, dw, dbias (0: none), B, H, W, C, k, ext, slots; per problem exactly
// mfnet_dwconv_bwd_weight (dw / dbias are ADDED into).
extern "C" int mfnet_dwconv_bwd_weight_batch(const int64_t* desc, int64_t n, hipStream_t stream) {
  MF_CHECK_ARG(desc || n == 0, "dwconv_bwd_weight_batch: null descriptor table");
  MF_CHECK_ARG(n >= 0 && n <= 1024, "dwconv_bwd_weight_batch: bad problem count");
  for (int64_t i0 = 0; i0 < n; i0 += DW_BATCH) {
    DwBatch b;
    b.n = (int)(n - i0 < DW_BATCH ? n - i0 : DW_BATCH);
    int64_t total = 0;
    for (int j = 0; j < b.n; ++j) {
      const int64_t* d = desc + 12 * (i0 + j);
      DwProb& q = b.q[j];
      MF_CHECK_ARG(d[0] && d[2] && d[3] && d[11] >= 1 && d[11] <= 64 && (d[1] == MFNET_BF16 || d[1] == MFNET_F32), "dwconv_bwd_weight_batch: bad problem %lld", (long long)(i0 + j));
      int rc = dw_check(&q.p, d[5], d[6], d[7], d[8], d[9], (int)d[10], "dwconv_bwd_weight_batch");
      if (rc) return rc;
      q.dy = (const void*)d[0]; q.bf16 = d[1] == MFNET_BF16; q.x = (const float*)d[2]; q.dw = (float*)d[3]; q.dbias = (float*)d[4]; q.slots = (int)d[11];
      int64_t nwg;
      dw_weight_geometry(d[5] * d[6] * d[7], d[9], &q.kind, &q.ppb, &nwg);
      q.wg_begin = (int)total; q.pad_ = 0;
      total += nwg;
    }
    for (int j = b.n; j < DW_BATCH; ++j) b.q[j] = b.q[0];
    hipLaunchKernelGGL(dwconv_bwd_weight_batch_kernel, dim3((unsigned)total), dim3(256), 0, stream, b);
    MF_LAUNCH_CHECK("dwconv_bwd_weight_batch");
  }
  return MFNET_OK;
}
extern "C" int mfnet_dwconv_bwd_weight(const void* dy, int dy_dtype, const float* x, float* dw, float* dbias, int64_t B, int64_t H, int64_t W,
                                       int64_t C, int64_t k, int ext, int slots, hipStream_t stream) {
  MF_CHECK_ARG(dy && x && dw && slots >= 1 && slots <= 64, "dwconv_bwd_weight: null pointer / slots not in [1, 64]");
  DwParams p;
  int rc = dw_check(&p, B, H, W, C, k, ext, "dwconv_bwd_weight");
  if (rc) return rc;
  const int64_t npix = B * H * W;
  int64_t ppb = ceil_div64(npix, 1024);
  if (ppb < 16) ppb = 16;
  if ((k == 3 || k == 8) && npix >= 4096) {
    // one workgroup per CU: the float atomics at the end (C * k * k per workgroup) are what the small maps pay for
    ppb = (ceil_div64(npix, 256) + 15) / 16 * 16;
    const dim3 gridw((unsigned)ceil_div64(npix, ppb));
#define DW_PX(BF, KK) hipLaunchKernelGGL((dwconv_bwd_weight_px_kernel<BF, KK>), gridw, dim3(256), 0, stream, dy, x, dw, dbias, p, ppb, slots)
    if (dy_dtype == MFNET_BF16) { if (k == 3) DW_PX(true, 3); else DW_PX(true, 8); }
    else { if (k == 3) DW_PX(false, 3); else DW_PX(false, 8); }
#undef DW_PX
  } else if (dy_dtype == MFNET_BF16) hipLaunchKernelGGL(dwconv_bwd_weight_kernel<true>, dim3((unsigned)ceil_div64(npix, ppb)), dim3(256), 0, stream, dy, dy_dtype, x, dw, dbias, p, ppb, slots);
  else hipLaunchKernelGGL(dwconv_bwd_weight_kernel<false>, dim3((unsigned)ceil_div64(npix, ppb)), dim3(256), 0, stream, dy, dy_dtype, x, dw, dbias, p, ppb, slots);
  MF_LAUNCH_CHECK("dwconv_bwd_weight");
  return MFNET_OK;
}

// =============================================================================== GLA window attention
// qkv [B*H*W, 3*heads*8] bf16, channel = (which*heads + head)*8 + d (the 1x1 qkv conv's output order,
// unetformer_mmsam.py:169-170); 8x8 windows; dots = q.k*scale + table[idx(i,j)][head]; softmax; .v
// (:172-184). out [B*H*W, heads*8] fp32, channel = head*8 + d. One wave = one (window, head), lane = token.
#define GLA_WS 8
#define GLA_HD 8
#define GLA_NB ((2 * GLA_WS - 1) * (2 * GLA_WS - 1))
struct GlaParams { int B, H, W, heads; float scale; };
__device__ __forceinline__ void gla_ld8(const bf16_t* p, float* o) {
  const u32x4_t v = *reinterpret_cast<const u32x4_t*>(p);
  o[0] = bf16_lo(v.x); o[1] = bf16_hi(v.x); o[2] = bf16_lo(v.y); o[3] = bf16_hi(v.y);
  o[4] = bf16_lo(v.z); o[5] = bf16_hi(v.z); o[6] = bf16_lo(v.w); o[7] = bf16_hi(v.w);
}
__device__ __forceinline__ void gla_ld8(const float* p, float* o) {          // fp32-accurate forward path: qkv kept in fp32
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
template <typename T>
__global__ __launch_bounds__(256) void gla_attn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ table,
                                                           float* __restrict__ out, const GlaParams p) {
  __shared__ float4 kl[4][64][2], vl[4][64][2];      // rows as two float4: broadcast ds_read_b128 in the key loops
  __shared__ float tl[4][GLA_NB];             // this head's bias column: the inner loops gather from LDS, not from global memory
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int head = blockIdx.y * 4 + wv;
  if (head < p.heads)
    for (int i = lane; i < GLA_NB; i += 64) tl[wv][i] = table[i * p.heads + head];
  const int nww = p.W / GLA_WS, nwh = p.H / GLA_WS;
  const int win = blockIdx.x;
  const int b = win / (nwh * nww), wy = (win / nww) % nwh, wx = win % nww;
  const int ty = lane >> 3, tx = lane & 7;
  const int64_t row = ((int64_t)b * p.H + wy * GLA_WS + ty) * p.W + wx * GLA_WS + tx;
  const int Cq = 3 * p.heads * GLA_HD;
  const bool on = head < p.heads;
  float q[8], k[8], v[8];
  if (on) {
    gla_ld8(qkv + row * Cq + head * GLA_HD, q);
    gla_ld8(qkv + row * Cq + (p.heads + head) * GLA_HD, k);
    gla_ld8(qkv + row * Cq + (2 * p.heads + head) * GLA_HD, v);
    kl[wv][lane][0] = make_float4(k[0], k[1], k[2], k[3]); kl[wv][lane][1] = make_float4(k[4], k[5], k[6], k[7]);
    vl[wv][lane][0] = make_float4(v[0], v[1], v[2], v[3]); vl[wv][lane][1] = make_float4(v[4], v[5], v[6], v[7]);
  }
  __syncthreads();
  if (!on) return;
  float s[64];
  float mx = -FLT_MAX;
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    const float4 k0 = kl[wv][j][0], k1 = kl[wv][j][1];
    float a = fmaf(q[0], k0.x, fmaf(q[1], k0.y, fmaf(q[2], k0.z, fmaf(q[3], k0.w, fmaf(q[4], k1.x, fmaf(q[5], k1.y, fmaf(q[6], k1.z, q[7] * k1.w)))))));
    const int idx = (ty - (j >> 3) + GLA_WS - 1) * (2 * GLA_WS - 1) + (tx - (j & 7) + GLA_WS - 1);
    a = a * p.scale + tl[wv][idx];
    s[j] = a;
    mx = fmaxf(mx, a);
  }
  float l = 0.f, o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    const float e = __expf(s[j] - mx);
    l += e;
    const float4 v0 = vl[wv][j][0], v1 = vl[wv][j][1];
    o[0] = fmaf(e, v0.x, o[0]); o[1] = fmaf(e, v0.y, o[1]); o[2] = fmaf(e, v0.z, o[2]); o[3] = fmaf(e, v0.w, o[3]);
    o[4] = fmaf(e, v1.x, o[4]); o[5] = fmaf(e, v1.y, o[5]); o[6] = fmaf(e, v1.z, o[6]); o[7] = fmaf(e, v1.w, o[7]);
  }
  const float inv = 1.f / l;
  float* op = out + row * (p.heads * GLA_HD) + head * GLA_HD;
  reinterpret_cast<float4*>(op)[0] = make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
  reinterpret_cast<float4*>(op)[1] = make_float4(o[4] * inv, o[5] * inv, o[6] * inv, o[7] * inv);
}
// backward: pass 1 (lane = query) -> max, sum, delta (one online-softmax sweep), then dq and the table-gradient adds (second
// sweep); pass 2 (lane = key) -> dk, dv. dtable [225, heads] is ADDED into: a block loops over several windows and keeps its
// share in LDS. K / V / Q / dO rows sit in LDS as two float4 each and are read as broadcasts (all lanes the same row): four
// ds_read_b128 per key instead of sixteen ds_read_b32 - the kernel is a chain of LDS reads, not of arithmetic.
struct GlaRow { float4 lo, hi; };
__device__ __forceinline__ float gla_dot8(const float* a, const GlaRow& r) {
  return fmaf(a[0], r.lo.x, fmaf(a[1], r.lo.y, fmaf(a[2], r.lo.z, fmaf(a[3], r.lo.w,
         fmaf(a[4], r.hi.x, fmaf(a[5], r.hi.y, fmaf(a[6], r.hi.z, a[7] * r.hi.w)))))));
}
__device__ __forceinline__ void gla_axpy8(float* acc, float s, const GlaRow& r) {
  acc[0] = fmaf(s, r.lo.x, acc[0]); acc[1] = fmaf(s, r.lo.y, acc[1]); acc[2] = fmaf(s, r.lo.z, acc[2]); acc[3] = fmaf(s, r.lo.w, acc[3]);
  acc[4] = fmaf(s, r.hi.x, acc[4]); acc[5] = fmaf(s, r.hi.y, acc[5]); acc[6] = fmaf(s, r.hi.z, acc[6]); acc[7] = fmaf(s, r.hi.w, acc[7]);
}
__device__ __forceinline__ GlaRow gla_row(const float* v) {
  GlaRow r;
  r.lo = make_float4(v[0], v[1], v[2], v[3]);
  r.hi = make_float4(v[4], v[5], v[6], v[7]);
  return r;
}
// One workgroup = one (window, head); its four waves split the 64 keys (pass 1) / the 64 queries (pass 2) and meet in LDS.
// (One wave per (window, head) executed ~10 k dependent wave64 VALU instructions back to back: 45-120 us per launch whatever
// the map size.)
__device__ __forceinline__ void gla_st8(bf16_t* p, const float* r) {
  u32x4_t w;
  w.x = pack_bf16x2(r[0], r[1]); w.y = pack_bf16x2(r[2], r[3]); w.z = pack_bf16x2(r[4], r[5]); w.w = pack_bf16x2(r[6], r[7]);
  *reinterpret_cast<u32x4_t*>(p) = w;
}
__device__ __forceinline__ void gla_st8(float* p, const float* r) {          // fp32 training path
  reinterpret_cast<float4*>(p)[0] = make_float4(r[0], r[1], r[2], r[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(r[4], r[5], r[6], r[7]);
}
template <typename T>
__global__ __launch_bounds__(256) void gla_attn_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ table,
                                                           const float* __restrict__ dout, T* __restrict__ dqkv,
                                                           float* __restrict__ dtable, const GlaParams p, int nwin) {
  __shared__ GlaRow ql[64], kl[64], vl[64], dol[64];
  __shared__ float pm[4][64], pl[4][64], pd[4][64];             // per-wave partial max / sum / delta sum of each query
  __shared__ float ml[64], ll[64], dl[64];
  __shared__ float part[4][64][17];                             // per-wave partial dq (8) or dk | dv (16) of each token
  __shared__ float dtl[GLA_NB], tl[GLA_NB];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int head = blockIdx.y;
  for (int i = threadIdx.x; i < GLA_NB; i += 256) { tl[i] = table[i * p.heads + head]; dtl[i] = 0.f; }
  const int nww = p.W / GLA_WS, nwh = p.H / GLA_WS;
  const int ty = lane >> 3, tx = lane & 7;
  const int Cq = 3 * p.heads * GLA_HD, Co = p.heads * GLA_HD;
  const int ib = (ty + GLA_WS - 1) * (2 * GLA_WS - 1) + (tx + GLA_WS - 1);       // bias index of (this lane as query, key 0)
  const int kb = (GLA_WS - 1 - ty) * (2 * GLA_WS - 1) + (GLA_WS - 1 - tx);       // bias index of (query 0, this lane as key)
  const int j0 = 16 * wv;
  for (int win = blockIdx.x; win < nwin; win += gridDim.x) {
    const int b = win / (nwh * nww), wy = (win / nww) % nwh, wx = win % nww;
    const int64_t row = ((int64_t)b * p.H + wy * GLA_WS + ty) * p.W + wx * GLA_WS + tx;
    __syncthreads();                      // previous window's LDS fully consumed
    {
      float t[8];
      if (wv < 3) {
        gla_ld8(qkv + row * Cq + (wv * p.heads + head) * GLA_HD, t);
      } else {
        const float4 d0 = reinterpret_cast<const float4*>(dout + row * Co + head * GLA_HD)[0];
        const float4 d1 = reinterpret_cast<const float4*>(dout + row * Co + head * GLA_HD)[1];
        t[0] = d0.x; t[1] = d0.y; t[2] = d0.z; t[3] = d0.w; t[4] = d1.x; t[5] = d1.y; t[6] = d1.z; t[7] = d1.w;
      }
      (wv == 0 ? ql : wv == 1 ? kl : wv == 2 ? vl : dol)[lane] = gla_row(t);
    }
    __syncthreads();
    float q[8], dO[8];
    {
      const GlaRow a = ql[lane], c = dol[lane];
      q[0] = a.lo.x; q[1] = a.lo.y; q[2] = a.lo.z; q[3] = a.lo.w; q[4] = a.hi.x; q[5] = a.hi.y; q[6] = a.hi.z; q[7] = a.hi.w;
      dO[0] = c.lo.x; dO[1] = c.lo.y; dO[2] = c.lo.z; dO[3] = c.lo.w; dO[4] = c.hi.x; dO[5] = c.hi.y; dO[6] = c.hi.z; dO[7] = c.hi.w;
    }
    // ---- pass 1a: this lane's query against this wave's 16 keys: scores kept, partial max / sum / delta sum
    float sc[16], dpv[16];
    float mx = -FLT_MAX;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int j = j0 + jj;
      const int idx = ib - (j >> 3) * (2 * GLA_WS - 1) - (j & 7);
      sc[jj] = gla_dot8(q, kl[j]) * p.scale + tl[idx];
      dpv[jj] = gla_dot8(dO, vl[j]);
      mx = fmaxf(mx, sc[jj]);
    }
    float l = 0.f, dsum = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const float e = __expf(sc[jj] - mx);
      l += e;
      dsum = fmaf(e, dpv[jj], dsum);
    }
    pm[wv][lane] = mx; pl[wv][lane] = l; pd[wv][lane] = dsum;
    __syncthreads();
    float M = fmaxf(fmaxf(pm[0][lane], pm[1][lane]), fmaxf(pm[2][lane], pm[3][lane]));
    float L = 0.f, Dm = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float c = __expf(pm[w][lane] - M);
      L = fmaf(pl[w][lane], c, L);
      Dm = fmaf(pd[w][lane], c, Dm);
    }
    const float inv = 1.f / L, delta = Dm * inv;
    if (wv == 0) { ml[lane] = M; ll[lane] = inv; dl[lane] = delta; }
    // ---- pass 1b: partial dq over this wave's keys, table-gradient adds
    float dq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int j = j0 + jj;
      const int idx = ib - (j >> 3) * (2 * GLA_WS - 1) - (j & 7);
      const float ds = __expf(sc[jj] - M) * inv * (dpv[jj] - delta);
      atomicAdd(&dtl[idx], ds);
      gla_axpy8(dq, ds * p.scale, kl[j]);
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) part[wv][lane][d] = dq[d];
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int d = 0; d < 8; ++d) dq[d] = (part[0][lane][d] + part[1][lane][d]) + (part[2][lane][d] + part[3][lane][d]);
      gla_st8(dqkv + row * Cq + head * GLA_HD, dq);
    }
    __syncthreads();                      // ml / ll / dl visible, part[] free again
    // ---- pass 2: this lane's key / value against this wave's 16 queries
    float k[8], v[8];
    {
      const GlaRow a = kl[lane], c = vl[lane];
      k[0] = a.lo.x; k[1] = a.lo.y; k[2] = a.lo.z; k[3] = a.lo.w; k[4] = a.hi.x; k[5] = a.hi.y; k[6] = a.hi.z; k[7] = a.hi.w;
      v[0] = c.lo.x; v[1] = c.lo.y; v[2] = c.lo.z; v[3] = c.lo.w; v[4] = c.hi.x; v[5] = c.hi.y; v[6] = c.hi.z; v[7] = c.hi.w;
    }
    float dk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ii = 0; ii < 16; ++ii) {
      const int i = j0 + ii;
      const GlaRow qr = ql[i], dr = dol[i];
      const int idx = kb + (i >> 3) * (2 * GLA_WS - 1) + (i & 7);
      const float a = gla_dot8(k, qr) * p.scale + tl[idx];
      const float dp = gla_dot8(v, dr);
      const float pr = __expf(a - ml[i]) * ll[i];
      const float ds = pr * (dp - dl[i]) * p.scale;
      gla_axpy8(dk, ds, qr);
      gla_axpy8(dv, pr, dr);
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) { part[wv][lane][d] = dk[d]; part[wv][lane][8 + d] = dv[d]; }
    __syncthreads();
    if (wv < 2) {                          // wave 0 sums and stores dk, wave 1 dv
      float r[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) r[d] = (part[0][lane][8 * wv + d] + part[1][lane][8 * wv + d]) + (part[2][lane][8 * wv + d] + part[3][lane][8 * wv + d]);
      gla_st8(dqkv + row * Cq + ((1 + wv) * p.heads + head) * GLA_HD, r);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < GLA_NB; i += 256) atomicAdd(dtable + i * p.heads + head, dtl[i]);
}
static int gla_check(GlaParams* p, int64_t B, int64_t H, int64_t W, int64_t heads, int64_t hd, int64_t ws, float scale, const char* who) {
  MF_CHECK_ARG(hd == GLA_HD && ws == GLA_WS, "%s: only head dim 8 / window 8 (decode_channels 64, 8 heads) are implemented", who);
  MF_CHECK_ARG(B > 0 && heads > 0 && H > 0 && W > 0 && H % GLA_WS == 0 && W % GLA_WS == 0,
               "%s: map %lldx%lld must be a multiple of the 8x8 window (input tiles must be multiples of 256)", who, (long long)H, (long long)W);
  MF_CHECK_ARG(B * (H / GLA_WS) * (W / GLA_WS) < (1ll << 31), "%s: too many windows", who);
  p->B = (int)B; p->H = (int)H; p->W = (int)W; p->heads = (int)heads; p->scale = scale;
  return MFNET_OK;
}
extern "C" int mfnet_gla_attn_fwd(const void* qkv, const float* table, float* out, int64_t B, int64_t H, int64_t W, int64_t heads,
                                  int64_t hd, int64_t ws, float scale, hipStream_t stream) {
  MF_CHECK_ARG(qkv && table && out, "gla_attn_fwd: null pointer");
  GlaParams p;
  int rc = gla_check(&p, B, H, W, heads, hd, ws, scale, "gla_attn_fwd");
  if (rc) return rc;
  const int64_t nwin = B * (H / GLA_WS) * (W / GLA_WS);
  hipLaunchKernelGGL(gla_attn_fwd_kernel<bf16_t>, dim3((unsigned)nwin, (unsigned)ceil_div64(heads, 4)), dim3(256), 0, stream, (const bf16_t*)qkv, table, out, p);
  MF_LAUNCH_CHECK("gla_attn_fwd");
  return MFNET_OK;
}
extern "C" int mfnet_gla_attn_fwd_f32(const float* qkv, const float* table, float* out, int64_t B, int64_t H, int64_t W, int64_t heads,
                                      int64_t hd, int64_t ws, float scale, hipStream_t stream) {
  MF_CHECK_ARG(qkv && table && out, "gla_attn_fwd_f32: null pointer");
  GlaParams p;
  int rc = gla_check(&p, B, H, W, heads, hd, ws, scale, "gla_attn_fwd_f32");
  if (rc) return rc;
  const int64_t nwin = B * (H / GLA_WS) * (W / GLA_WS);
  hipLaunchKernelGGL(gla_attn_fwd_kernel<float>, dim3((unsigned)nwin, (unsigned)ceil_div64(heads, 4)), dim3(256), 0, stream, qkv, table, out, p);
  MF_LAUNCH_CHECK("gla_attn_fwd_f32");
  return MFNET_OK;
}
extern "C" int mfnet_gla_attn_bwd(const void* qkv, const float* table, const float* dout, void* dqkv, float* dtable, int64_t B, int64_t H,
                                  int64_t W, int64_t heads, int64_t hd, int64_t ws, float scale, hipStream_t stream) {
  MF_CHECK_ARG(qkv && table && dout && dqkv && dtable, "gla_attn_bwd: null pointer");
  GlaParams p;
  int rc = gla_check(&p, B, H, W, heads, hd, ws, scale, "gla_attn_bwd");
  if (rc) return rc;
  const int64_t nwin = B * (H / GLA_WS) * (W / GLA_WS);
  const unsigned gx = (unsigned)(nwin < 512 ? nwin : 512);
  hipLaunchKernelGGL(gla_attn_bwd_kernel<bf16_t>, dim3(gx, (unsigned)heads), dim3(256), 0, stream, (const bf16_t*)qkv, table, dout,
                     (bf16_t*)dqkv, dtable, p, (int)nwin);
  MF_LAUNCH_CHECK("gla_attn_bwd");
  return MFNET_OK;
}
extern "C" int mfnet_gla_attn_bwd_f32(const float* qkv, const float* table, const float* dout, float* dqkv, float* dtable, int64_t B, int64_t H,
                                      int64_t W, int64_t heads, int64_t hd, int64_t ws, float scale, hipStream_t stream) {
  MF_CHECK_ARG(qkv && table && dout && dqkv && dtable, "gla_attn_bwd_f32: null pointer");
  GlaParams p;
  int rc = gla_check(&p, B, H, W, heads, hd, ws, scale, "gla_attn_bwd_f32");
  if (rc) return rc;
  const int64_t nwin = B * (H / GLA_WS) * (W / GLA_WS);
  const unsigned gx = (unsigned)(nwin < 512 ? nwin : 512);
  hipLaunchKernelGGL(gla_attn_bwd_kernel<float>, dim3(gx, (unsigned)heads), dim3(256), 0, stream, qkv, table, dout, dqkv, dtable, p, (int)nwin);
  MF_LAUNCH_CHECK("gla_attn_bwd_f32");
  return MFNET_OK;
}

// =============================================================================== GLA pooling
// out = AvgPool(8,1)(pad_bottom_reflect(o)) + AvgPool(1,8)(pad_right_reflect(o)) + l1 + l2   (unetformer_mmsam.py:187-191;
// kernel 8, stride 1, zero padding 3, count_include_pad => divisor 8). o, l1, l2, out [B,H,W,C] fp32.
__global__ __launch_bounds__(256) void gla_pool_fwd_kernel(const float* __restrict__ o, const float* __restrict__ l1, const float* __restrict__ l2,
                                                           float* __restrict__ out, int B, int H, int W, int C4) {
  const int64_t total = (int64_t)B * H * W * C4;
  GS_LOOP(i, total) {
    const int cg = (int)(i % C4);
    const int64_t pix = i / C4;
    const int xx = (int)(pix % W), yy = (int)((pix / W) % H);
    const int64_t b = pix / ((int64_t)W * H);
    float4 acc = F4ZERO;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int sy = ext_map(yy - 3 + t, H, 1), sx = ext_map(xx - 3 + t, W, 1);
      if (sy >= 0) acc = f4add(acc, reinterpret_cast<const float4*>(o)[((b * H + sy) * W + xx) * C4 + cg]);
      if (sx >= 0) acc = f4add(acc, reinterpret_cast<const float4*>(o)[((b * H + yy) * W + sx) * C4 + cg]);
    }
    reinterpret_cast<float4*>(out)[i] = f4add(f4scale(acc, 0.125f), f4add(reinterpret_cast<const float4*>(l1)[i], reinterpret_cast<const float4*>(l2)[i]));
  }
}
// do[r,x] = 1/8 * ( sum_{y in [r-4, r+3]} ds[y,x] + [r == H-2] sum_{y in [H-4, H-1]} ds[y,x] + same along x )
__global__ __launch_bounds__(256) void gla_pool_bwd_kernel(const float* __restrict__ ds, float* __restrict__ d_o, int B, int H, int W, int C4) {
  const int64_t total = (int64_t)B * H * W * C4;
  GS_LOOP(i, total) {
    const int cg = (int)(i % C4);
    const int64_t pix = i / C4;
    const int xx = (int)(pix % W), yy = (int)((pix / W) % H);
    const int64_t b = pix / ((int64_t)W * H);
    float4 acc = F4ZERO;
    for (int e = 0; e < 2; ++e) {           // e = 1: the reflected ext row / column H (W) that maps onto H-2 (W-2)
      if (e == 1 && yy != H - 2) break;
      const int r = e ? H : yy;
      for (int y = r - 4; y <= r + 3; ++y)
        if (y >= 0 && y < H) acc = f4add(acc, reinterpret_cast<const float4*>(ds)[((b * H + y) * W + xx) * C4 + cg]);
    }
    for (int e = 0; e < 2; ++e) {
      if (e == 1 && xx != W - 2) break;
      const int r = e ? W : xx;
      for (int x = r - 4; x <= r + 3; ++x)
        if (x >= 0 && x < W) acc = f4add(acc, reinterpret_cast<const float4*>(ds)[((b * H + yy) * W + x) * C4 + cg]);
    }
    reinterpret_cast<float4*>(d_o)[i] = f4scale(acc, 0.125f);
  }
}
extern "C" int mfnet_gla_pool_fwd(const float* o, const float* l1, const float* l2, float* out, int64_t B, int64_t H, int64_t W, int64_t C, hipStream_t stream) {
  MF_CHECK_ARG(o && l1 && l2 && out && C % 4 == 0 && H >= 2 && W >= 2, "gla_pool_fwd: bad args");
  hipLaunchKernelGGL(gla_pool_fwd_kernel, dim3(grid_for(B * H * W * C / 4)), dim3(256), 0, stream, o, l1, l2, out, (int)B, (int)H, (int)W, (int)(C / 4));
  MF_LAUNCH_CHECK("gla_pool_fwd");
  return MFNET_OK;
}
extern "C" int mfnet_gla_pool_bwd(const float* ds, float* d_o, int64_t B, int64_t H, int64_t W, int64_t C, hipStream_t stream) {
  MF_CHECK_ARG(ds && d_o && C % 4 == 0 && H >= 2 && W >= 2, "gla_pool_bwd: bad args");
  hipLaunchKernelGGL(gla_pool_bwd_kernel, dim3(grid_for(B * H * W * C / 4)), dim3(256), 0, stream, ds, d_o, (int)B, (int)H, (int)W, (int)(C / 4));
  MF_LAUNCH_CHECK("gla_pool_bwd");
  return MFNET_OK;
}

// =============================================================================== WF / FRH fusion with bilinear x2
// out[b,Y,X,:] = fw0 * skip[perm(b,Y,X),:] + fw1 * bilinear_x2(x)[b,Y,X,:]   fw = relu(w) / (sum relu(w) + eps)
// (unetformer_mmsam.py:242-246, 342-345; F.interpolate scale 2, align_corners=False). x [B,h,w,C] fp32 with
// H = 2h, W = 2w. `skip` rows are in the un-shuffled ConvTranspose order of depth `perm` (0 row-major,
// 1 one k2s2 ConvT above the (h,w) grid... see skip_row()).
__device__ __forceinline__ int64_t skip_row(int perm, int64_t b, int Y, int X, int H, int W) {
  if (perm == 0) return (b * H + Y) * W + X;
  if (perm == 1) return (((b * (H >> 1) + (Y >> 1)) * (W >> 1) + (X >> 1)) << 2) + ((Y & 1) << 1) + (X & 1);
  return (((((b * (H >> 2) + (Y >> 2)) * (W >> 2) + (X >> 2)) << 2) + (((Y >> 1) & 1) << 1) + ((X >> 1) & 1)) << 2) + ((Y & 1) << 1) + (X & 1);
}
__device__ __forceinline__ void bilin_src(int O, int n_in, float inv_scale, int* i0, int* i1, float* l) {
  float s = (O + 0.5f) * inv_scale - 0.5f;
  s = s < 0.f ? 0.f : s;
  const int a = (int)s;
  *i0 = a; *i1 = a + 1 < n_in ? a + 1 : n_in - 1; *l = s - (float)a;
}
__device__ __forceinline__ void fuse_w(const float* w, float eps, float* f0, float* f1) {
  const float a = fmaxf(w[0], 0.f), b = fmaxf(w[1], 0.f), inv = 1.f / (a + b + eps);
  *f0 = a * inv; *f1 = b * inv;
}
__global__ __launch_bounds__(256) void upfuse_fwd_kernel(const float* __restrict__ x, const void* __restrict__ skip, int skip_dt, int perm,
                                                         const float* __restrict__ w, float eps, void* __restrict__ out, int out_dt,
                                                         int B, int H, int W, int C4) {
  float f0, f1;
  fuse_w(w, eps, &f0, &f1);
  const int h = H >> 1, wd = W >> 1;
  const int64_t total = (int64_t)B * H * W * C4;
  GS_LOOP(i, total) {
    const int cg = (int)(i % C4);
    const int64_t pix = i / C4;
    const int X = (int)(pix % W), Y = (int)((pix / W) % H);
    const int64_t b = pix / ((int64_t)W * H);
    int y0, y1, x0, x1; float ly, lx;
    bilin_src(Y, h, 0.5f, &y0, &y1, &ly); bilin_src(X, wd, 0.5f, &x0, &x1, &lx);
    const float4* xb = reinterpret_cast<const float4*>(x) + b * h * wd * C4 + cg;
    const float4 v00 = xb[(int64_t)(y0 * wd + x0) * C4], v01 = xb[(int64_t)(y0 * wd + x1) * C4];
    const float4 v10 = xb[(int64_t)(y1 * wd + x0) * C4], v11 = xb[(int64_t)(y1 * wd + x1) * C4];
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    float4 up = f4scale(v00, w00);
    up = f4fma(v01, make_float4(w01, w01, w01, w01), up); up = f4fma(v10, make_float4(w10, w10, w10, w10), up);
    up = f4fma(v11, make_float4(w11, w11, w11, w11), up);
    const float4 sk = ld4(skip, skip_dt, skip_row(perm, b, Y, X, H, W) * C4 + cg);
    st4(out, out_dt, i, f4add(f4scale(sk, f0), f4scale(up, f1)));
  }
}
#define UPFUSE_SLOTS 32
// dskip[perm row] = fw0*df ; sums[slot][0] += sum df*skip ; sums[slot][1] += sum df*up(x)   (for the 2 fusion weights)
__global__ __launch_bounds__(256) void upfuse_bwd_skip_kernel(const float* __restrict__ df, const float* __restrict__ x, const void* __restrict__ skip,
                                                              int skip_dt, int perm, const float* __restrict__ w, float eps,
                                                              void* __restrict__ dskip, int dskip_dt, float* __restrict__ sums,
                                                              int B, int H, int W, int C4) {
  float f0, f1;
  fuse_w(w, eps, &f0, &f1);
  const int h = H >> 1, wd = W >> 1;
  const int64_t total = (int64_t)B * H * W * C4;
  float s0 = 0.f, s1 = 0.f;
  GS_LOOP(i, total) {
    const int cg = (int)(i % C4);
    const int64_t pix = i / C4;
    const int X = (int)(pix % W), Y = (int)((pix / W) % H);
    const int64_t b = pix / ((int64_t)W * H);
    int y0, y1, x0, x1; float ly, lx;
    bilin_src(Y, h, 0.5f, &y0, &y1, &ly); bilin_src(X, wd, 0.5f, &x0, &x1, &lx);
    const float4* xb = reinterpret_cast<const float4*>(x) + b * h * wd * C4 + cg;
    const float4 v00 = xb[(int64_t)(y0 * wd + x0) * C4], v01 = xb[(int64_t)(y0 * wd + x1) * C4];
    const float4 v10 = xb[(int64_t)(y1 * wd + x0) * C4], v11 = xb[(int64_t)(y1 * wd + x1) * C4];
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    float4 up = f4scale(v00, w00);
    up = f4fma(v01, make_float4(w01, w01, w01, w01), up); up = f4fma(v10, make_float4(w10, w10, w10, w10), up);
    up = f4fma(v11, make_float4(w11, w11, w11, w11), up);
    const float4 d = reinterpret_cast<const float4*>(df)[i];
    const int64_t sr = skip_row(perm, b, Y, X, H, W) * C4 + cg;
    const float4 sk = ld4(skip, skip_dt, sr);
    st4(dskip, dskip_dt, sr, f4scale(d, f0));
    s0 += d.x * sk.x + d.y * sk.y + d.z * sk.z + d.w * sk.w;
    s1 += d.x * up.x + d.y * up.y + d.z * up.z + d.w * up.w;
  }
  // float atomics on one address serialise (~25 ns each): one add per workgroup, spread over UPFUSE_SLOTS addresses
  __shared__ float red[2][4];
  s0 = wave_sum(s0); s1 = wave_sum(s1);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; }
  __syncthreads();
  if (threadIdx.x < 2) {
    const float t = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    atomicAdd(sums + 2 * (blockIdx.x % UPFUSE_SLOTS) + threadIdx.x, t);
  }
}
// generic adjoint of bilinear upsampling by integer factor s (gather form): dx[b,y,x,:] = fw * sum_{Y,X} wy(Y->y) wx(X->x) df[b,Y,X,:]
__device__ __forceinline__ float bilin_wt(int O, int i, int n_in, float inv_scale) {
  int i0, i1; float l;
  bilin_src(O, n_in, inv_scale, &i0, &i1, &l);
  return (i0 == i ? 1.f - l : 0.f) + (i1 == i ? l : 0.f);
}
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ df, const float* __restrict__ w, float eps, float* __restrict__ dx,
                                                           int B, int h, int wd, int C4, int s) {
  float f1 = 1.f;
  if (w) { float f0; fuse_w(w, eps, &f0, &f1); }
  const int H = h * s, W = wd * s;
  const float inv = 1.f / (float)s;
  const int64_t total = (int64_t)B * h * wd * C4;
  GS_LOOP(i, total) {
    const int cg = (int)(i % C4);
    const int64_t pix = i / C4;
    const int x = (int)(pix % wd), y = (int)((pix / wd) % h);
    const int64_t b = pix / ((int64_t)wd * h);
    float4 acc = F4ZERO;
    const int Y0 = max(0, s * y - s - s / 2 - 1), Y1 = min(H - 1, s * y + 2 * s + s / 2);
    const int X0 = max(0, s * x - s - s / 2 - 1), X1 = min(W - 1, s * x + 2 * s + s / 2);
    for (int Y = Y0; Y <= Y1; ++Y) {
      const float wy = bilin_wt(Y, y, h, inv);
      if (wy == 0.f) continue;
      for (int X = X0; X <= X1; ++X) {
        const float wx = bilin_wt(X, x, wd, inv);
        if (wx == 0.f) continue;
        const float ww = wy * wx;
        acc = f4fma(reinterpret_cast<const float4*>(df)[((b * H + Y) * W + X) * C4 + cg], make_float4(ww, ww, ww, ww), acc);
      }
    }
    reinterpret_cast<float4*>(dx)[i] = f4scale(acc, f1);
  }
}
// d weights (2): fw = relu(w)/(S+eps), S = relu(w0)+relu(w1); sums = (dL/dfw0, dL/dfw1)
__global__ void fuse_w_bwd_kernel(const float* w, float eps, const float* sums, float* dw) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float g0 = 0.f, g1 = 0.f;
  for (int k = 0; k < UPFUSE_SLOTS; ++k) { g0 += sums[2 * k]; g1 += sums[2 * k + 1]; }
  const float a = fmaxf(w[0], 0.f), b = fmaxf(w[1], 0.f), S = a + b + eps;
  const float common = (g0 * a + g1 * b) / (S * S);
  dw[0] += (w[0] > 0.f) ? (g0 / S - common) : 0.f;
  dw[1] += (w[1] > 0.f) ? (g1 / S - common) : 0.f;
}
extern "C" int mfnet_upfuse_fwd(const float* x, const void* skip, int skip_dtype, int perm, const float* w, float eps, void* out, int out_dtype,
                                int64_t B, int64_t H, int64_t W, int64_t C, hipStream_t stream) {
  MF_CHECK_ARG(x && skip && w && out && C % 4 == 0 && H % 2 == 0 && W % 2 == 0 && perm >= 0 && perm <= 2, "upfuse_fwd: bad args");
  MF_CHECK_ARG(perm < 2 || (H % 4 == 0 && W % 4 == 0), "upfuse_fwd: perm 2 needs H, W multiples of 4");
  hipLaunchKernelGGL(upfuse_fwd_kernel, dim3(grid_for(B * H * W * C / 4)), dim3(256), 0, stream, x, skip, skip_dtype, perm, w, eps, out, out_dtype,
                     (int)B, (int)H, (int)W, (int)(C / 4));
  MF_LAUNCH_CHECK("upfuse_fwd");
  return MFNET_OK;
}
// df [B,H,W,C] fp32 -> dx [B,H/2,W/2,C] fp32, dskip (skip's row order), dw[2] += gradient of the fusion weights.
// sums: scratch of 64 floats (32 slots x 2) owned by the caller (zeroed here).
extern "C" int mfnet_upfuse_bwd(const float* df, const float* x, const void* skip, int skip_dtype, int perm, const float* w, float eps,
                                float* dx, void* dskip, int dskip_dtype, float* dw, float* sums, int64_t B, int64_t H, int64_t W, int64_t C,
                                hipStream_t stream) {
  MF_CHECK_ARG(df && x && skip && w && dx && dskip && dw && sums && C % 4 == 0 && H % 2 == 0 && W % 2 == 0 && perm >= 0 && perm <= 2, "upfuse_bwd: bad args");
  // (a kernel, not hipMemsetAsync: as a memset NODE of a captured graph the clear was seen to lose its order against the kernels
  // around it on the first replay after an idle second - garbage fusion-weight gradients)
  hipLaunchKernelGGL(zero_f32_kernel, dim3(1), dim3(64), 0, stream, sums, 2 * UPFUSE_SLOTS);
  hipLaunchKernelGGL(upfuse_bwd_skip_kernel, dim3(grid_for(B * H * W * C / 4, 1024)), dim3(256), 0, stream, df, x, skip, skip_dtype, perm, w, eps, dskip,
                     dskip_dtype, sums, (int)B, (int)H, (int)W, (int)(C / 4));
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_for(B * (H / 2) * (W / 2) * C / 4)), dim3(256), 0, stream, df, w, eps, dx, (int)B, (int)(H / 2), (int)(W / 2),
                     (int)(C / 4), 2);
  hipLaunchKernelGGL(fuse_w_bwd_kernel, dim3(1), dim3(64), 0, stream, w, eps, sums, dw);
  MF_LAUNCH_CHECK("upfuse_bwd");
  return MFNET_OK;
}

// =============================================================================== final bilinear xS into NCHW logits
// x [B,h,w,ldx] fp32 (first ncls channels used) -> logits [B,ncls,h*s,w*s] fp32 (unetformer_mmsam.py:455)
__global__ __launch_bounds__(256) void up_final_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out, int B, int h, int wd, int ncls, int s) {
  const int H = h * s, W = wd * s;
  const float inv = 1.f / (float)s;
  const int64_t total = (int64_t)B * ncls * H * W;
  GS_LOOP(i, total) {
    const int X = (int)(i % W), Y = (int)((i / W) % H);
    const int c = (int)((i / ((int64_t)W * H)) % ncls);
    const int64_t b = i / ((int64_t)W * H * ncls);
    int y0, y1, x0, x1; float ly, lx;
    bilin_src(Y, h, inv, &y0, &y1, &ly); bilin_src(X, wd, inv, &x0, &x1, &lx);
    const float* xb = x + b * h * wd * ldx + c;
    const float v = (1.f - ly) * ((1.f - lx) * xb[(int64_t)(y0 * wd + x0) * ldx] + lx * xb[(int64_t)(y0 * wd + x1) * ldx]) +
                    ly * ((1.f - lx) * xb[(int64_t)(y1 * wd + x0) * ldx] + lx * xb[(int64_t)(y1 * wd + x1) * ldx]);
    out[i] = v;
  }
}
// s == 4 (the decoder's head): one thread per (image, output row, SOURCE pixel) produces the 4 output columns of that pixel for
// every class: the 2 x 3 source pixels it needs are loaded once as contiguous class rows, the 4 results per class leave as one
// float4. (One thread per output element gathered 4 strided words each: 48 us for a 25 MB map.) Same expression per element.
__global__ __launch_bounds__(256) void up_final_fwd4_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out, int B, int h, int wd, int ncls) {
  const int H = h * 4, W = wd * 4;
  const int64_t total = (int64_t)B * H * wd;
  GS_LOOP(i, total) {
    const int xs = (int)(i % wd), Y = (int)((i / wd) % H);
    const int64_t b = i / ((int64_t)wd * H);
    int y0, y1; float ly;
    bilin_src(Y, h, 0.25f, &y0, &y1, &ly);
    const int cols[3] = {xs > 0 ? xs - 1 : 0, xs, xs + 1 < wd ? xs + 1 : wd - 1};
    const float* r0 = x + (b * h + y0) * (int64_t)wd * ldx;
    const float* r1 = x + (b * h + y1) * (int64_t)wd * ldx;
    int i0[4], i1[4]; float lx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int x0, x1;
      bilin_src(4 * xs + k, wd, 0.25f, &x0, &x1, &lx[k]);
      i0[k] = x0 - (xs - 1); i1[k] = x1 - (xs - 1);            // which of the three columns (x0, x1 are within xs-1 .. xs+1)
    }
    for (int c = 0; c < ncls; ++c) {
      float a[3], d[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) { a[t] = r0[(int64_t)cols[t] * ldx + c]; d[t] = r1[(int64_t)cols[t] * ldx + c]; }
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a0 = i0[k] == 0 ? a[0] : (i0[k] == 1 ? a[1] : a[2]), a1 = i1[k] == 0 ? a[0] : (i1[k] == 1 ? a[1] : a[2]);
        const float d0 = i0[k] == 0 ? d[0] : (i0[k] == 1 ? d[1] : d[2]), d1 = i1[k] == 0 ? d[0] : (i1[k] == 1 ? d[1] : d[2]);
        v[k] = (1.f - ly) * ((1.f - lx[k]) * a0 + lx[k] * a1) + ly * ((1.f - lx[k]) * d0 + lx[k] * d1);
      }
      *reinterpret_cast<float4*>(out + ((b * ncls + c) * H + Y) * (int64_t)W + 4 * xs) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}
// dlogits [B,ncls,H,W] fp32 -> dx [B,h,w,ldx] bf16 (channels >= ncls are written as zero). One thread per low-resolution
// pixel: the adjoint of the xS bilinear upsample reads the <= (2S+2)^2 logit gradients whose interpolation touches the
// pixel (exact weights from bilin_wt, separable: 2*(2S+2) weight evaluations), for every class, and writes the
// pixel's whole channel row. (The first version ran one thread per (pixel, channel) of the 64-wide row - 6 of 64
// lanes busy - over a (3.5S)^2 window: 147 us; the gradient map is 25 MB.)
#define UPF_MAX_S 8
#define UPF_MAX_CLS 16
// One thread per (image, class, low-resolution pixel), x fastest: neighbouring lanes read neighbouring windows of one class plane.
// (One thread per pixel looping over the classes: 65 k threads of 600 loads each, 58 us.) Same sum order per element.
__device__ __forceinline__ void upf_store(bf16_t* o, float v) { *o = f32_to_bf16(v); }
__device__ __forceinline__ void upf_store(float* o, float v) { *o = v; }
template <typename TO>
__global__ __launch_bounds__(256) void up_final_bwd_cls_kernel(const float* __restrict__ dl, TO* __restrict__ dx, int ldx, int B, int h, int wd, int ncls, int s) {
  const int H = h * s, W = wd * s;
  const float inv = 1.f / (float)s;
  const int64_t total = (int64_t)B * ncls * h * wd;
  const int span = 2 * s + 2;
  GS_LOOP(i, total) {
    const int x = (int)(i % wd), y = (int)((i / wd) % h);
    const int c = (int)((i / ((int64_t)wd * h)) % ncls);
    const int64_t b = i / ((int64_t)wd * h * ncls);
    const int Yb = s * y - s / 2 - 1 - (s & 1), Xb = s * x - s / 2 - 1 - (s & 1);
    float wy[2 * UPF_MAX_S + 2], wx[2 * UPF_MAX_S + 2];
    for (int k = 0; k < span; ++k) {
      const int Y = Yb + k, X = Xb + k;
      wy[k] = (Y >= 0 && Y < H) ? bilin_wt(Y, y, h, inv) : 0.f;
      wx[k] = (X >= 0 && X < W) ? bilin_wt(X, x, wd, inv) : 0.f;
    }
    float acc = 0.f;
    const float* plane = dl + (b * ncls + c) * (int64_t)H * W;
    for (int ky = 0; ky < span; ++ky) {
      if (wy[ky] == 0.f) continue;
      const float* p = plane + (int64_t)(Yb + ky) * W + Xb;
      for (int kx = 0; kx < span; ++kx) {
        if (wx[kx] == 0.f) continue;
        acc = fmaf(wy[ky] * wx[kx], p[kx], acc);
      }
    }
    TO* o = dx + ((b * h + y) * (int64_t)wd + x) * ldx;
    upf_store(o + c, acc);
    if (c == 0)
      for (int z = ncls; z < ldx; ++z) upf_store(o + z, 0.f);
  }
}
extern "C" int mfnet_up_final_fwd(const float* x, int64_t ldx, float* logits, int64_t B, int64_t h, int64_t w, int64_t ncls, int64_t s, hipStream_t stream) {
  MF_CHECK_ARG(x && logits && ncls > 0 && ncls <= ldx && s >= 1 && s <= 8, "up_final_fwd: bad args");
  if (s == 4) hipLaunchKernelGGL(up_final_fwd4_kernel, dim3(grid_for(B * h * 4 * w)), dim3(256), 0, stream, x, (int)ldx, logits, (int)B, (int)h, (int)w, (int)ncls);
  else hipLaunchKernelGGL(up_final_fwd_kernel, dim3(grid_for(B * ncls * h * s * w * s)), dim3(256), 0, stream, x, (int)ldx, logits, (int)B, (int)h, (int)w, (int)ncls, (int)s);
  MF_LAUNCH_CHECK("up_final_fwd");
  return MFNET_OK;
}
extern "C" int mfnet_up_final_bwd(const float* dlogits, void* dx, int64_t ldx, int64_t B, int64_t h, int64_t w, int64_t ncls, int64_t s, hipStream_t stream) {
  MF_CHECK_ARG(dlogits && dx && ncls > 0 && ncls <= ldx && ncls <= UPF_MAX_CLS && s >= 1 && s <= UPF_MAX_S, "up_final_bwd: bad args (classes <= 16, scale <= 8)");
  hipLaunchKernelGGL(up_final_bwd_cls_kernel<bf16_t>, dim3(grid_for(B * ncls * h * w)), dim3(256), 0, stream, dlogits, (bf16_t*)dx, (int)ldx, (int)B, (int)h, (int)w, (int)ncls, (int)s);
  MF_LAUNCH_CHECK("up_final_bwd");
  return MFNET_OK;
}
extern "C" int mfnet_up_final_bwd_f32(const float* dlogits, float* dx, int64_t ldx, int64_t B, int64_t h, int64_t w, int64_t ncls, int64_t s, hipStream_t stream) {
  MF_CHECK_ARG(dlogits && dx && ncls > 0 && ncls <= ldx && ncls <= UPF_MAX_CLS && s >= 1 && s <= UPF_MAX_S, "up_final_bwd_f32: bad args (classes <= 16, scale <= 8)");
  hipLaunchKernelGGL(up_final_bwd_cls_kernel<float>, dim3(grid_for(B * ncls * h * w)), dim3(256), 0, stream, dlogits, dx, (int)ldx, (int)B, (int)h, (int)w, (int)ncls, (int)s);
  MF_LAUNCH_CHECK("up_final_bwd_f32");
  return MFNET_OK;
}

// =============================================================================== fpn4: MaxPool2d(2,2)
// x [B,H,W,C] fp32 -> y [B,H/2,W/2,C] fp32 + arg (0..3, first maximum wins like ATen) uint8
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ arg, int B, int H, int W, int C) {
  const int h = H >> 1, wd = W >> 1;
  const int64_t total = (int64_t)B * h * wd * C;
  GS_LOOP(i, total) {
    const int c = (int)(i % C);
    const int64_t pix = i / C;
    const int xx = (int)(pix % wd), yy = (int)((pix / wd) % h);
    const int64_t b = pix / ((int64_t)wd * h);
    const float* p = x + ((b * H + 2 * yy) * W + 2 * xx) * C + c;
    float m = p[0]; int a = 0;
    float v = p[C]; if (v > m) { m = v; a = 1; }
    v = p[(int64_t)W * C]; if (v > m) { m = v; a = 2; }
    v = p[(int64_t)W * C + C]; if (v > m) { m = v; a = 3; }
    y[i] = m; arg[i] = (unsigned char)a;
  }
}
// dx[b,2y+dy,2x+dx,c] (+= if accumulate) (arg == dy*2+dx) * dy_[b,y,x,c]
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const void* __restrict__ dy, int dy_dt, const unsigned char* __restrict__ arg, float* __restrict__ dx,
                                                           int B, int H, int W, int C, int accumulate) {
  const int h = H >> 1, wd = W >> 1;
  const int64_t total = (int64_t)B * H * W * C;
  GS_LOOP(i, total) {
    const int c = (int)(i % C);
    const int64_t pix = i / C;
    const int X = (int)(pix % W), Y = (int)((pix / W) % H);
    const int64_t b = pix / ((int64_t)W * H);
    const int64_t j = ((b * h + (Y >> 1)) * wd + (X >> 1)) * C + c;
    const float g = (arg[j] == ((Y & 1) * 2 + (X & 1))) ? (dy_dt == MFNET_F32 ? reinterpret_cast<const float*>(dy)[j] : bf16_to_f32(reinterpret_cast<const bf16_t*>(dy)[j])) : 0.f;
    dx[i] = accumulate ? dx[i] + g : g;
  }
}
extern "C" int mfnet_maxpool2_fwd(const float* x, float* y, unsigned char* arg, int64_t B, int64_t H, int64_t W, int64_t C, hipStream_t stream) {
  MF_CHECK_ARG(x && y && arg && H % 2 == 0 && W % 2 == 0, "maxpool2_fwd: bad args");
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid_for(B * H * W * C / 4)), dim3(256), 0, stream, x, y, arg, (int)B, (int)H, (int)W, (int)C);
  MF_LAUNCH_CHECK("maxpool2_fwd");
  return MFNET_OK;
}
extern "C" int mfnet_maxpool2_bwd(const void* dy, int dy_dtype, const unsigned char* arg, float* dx, int64_t B, int64_t H, int64_t W, int64_t C, int accumulate, hipStream_t stream) {
  MF_CHECK_ARG(dy && arg && dx && H % 2 == 0 && W % 2 == 0, "maxpool2_bwd: bad args");
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(B * H * W * C)), dim3(256), 0, stream, dy, dy_dtype, arg, dx, (int)B, (int)H, (int)W, (int)C, accumulate);
  MF_LAUNCH_CHECK("maxpool2_bwd");
  return MFNET_OK;
}

// =============================================================================== squeeze-excite gates
// One block per image: mean = pooled_sum/P ; z = act1(W1 mean + b1) [Cr] ; g = sigmoid(W2 z + b2) [C].
// SqueezeAndExcitation (unetformer_mmsam.py:251-266): C=256, Cr=16, biases, ReLU. FRH channel attention
// (:333-337): C=64, Cr=4, no biases, ReLU6. W1 [Cr,C], W2 [C,Cr] (1x1 conv weights).
#define SE_MAXC 256
#define SE_MAXR 16
__global__ __launch_bounds__(256) void se_gate_fwd_kernel(const float* __restrict__ pooled, float inv_p, const float* __restrict__ W1, const float* __restrict__ b1,
                                                          const float* __restrict__ W2, const float* __restrict__ b2, int relu6, float* __restrict__ z_out,
                                                          float* __restrict__ g_out, int C, int Cr) {
  __shared__ float m[SE_MAXC], z[SE_MAXR];
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < C) m[t] = pooled[(int64_t)b * C + t] * inv_p;
  __syncthreads();
  const int lane = t & 63, wv = t >> 6;
  for (int r = wv; r < Cr; r += 4) {
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a = fmaf(W1[r * C + c], m[c], a);
    a = wave_sum(a);
    if (lane == 0) {
      a += b1 ? b1[r] : 0.f;
      a = relu6 ? relu6f(a) : fmaxf(a, 0.f);
      z[r] = a; z_out[(int64_t)b * Cr + r] = a;
    }
  }
  __syncthreads();
  if (t < C) {
    float a = b2 ? b2[t] : 0.f;
    for (int r = 0; r < Cr; ++r) a = fmaf(W2[t * Cr + r], z[r], a);
    g_out[(int64_t)b * C + t] = sigmoidf_(a);
  }
}
// dg [B,C] -> dpool [B,C] = d(pooled mean)/P (ready to be added to every pixel) ; weight grads ADDED into.
__global__ __launch_bounds__(256) void se_gate_bwd_kernel(const float* __restrict__ dg, const float* __restrict__ pooled, float inv_p, const float* __restrict__ W1,
                                                          const float* __restrict__ W2, int relu6, const float* __restrict__ z_in, const float* __restrict__ g_in,
                                                          float* __restrict__ dpool, float* __restrict__ dW1, float* __restrict__ db1, float* __restrict__ dW2,
                                                          float* __restrict__ db2, int C, int Cr) {
  __shared__ float d2[SE_MAXC], dz[SE_MAXR], z[SE_MAXR];
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < Cr) z[t] = z_in[(int64_t)b * Cr + t];
  __syncthreads();
  if (t < C) {
    const float g = g_in[(int64_t)b * C + t];
    const float d = dg[(int64_t)b * C + t] * g * (1.f - g);
    d2[t] = d;
    if (db2) atomicAdd(db2 + t, d);
    for (int r = 0; r < Cr; ++r) atomicAdd(dW2 + t * Cr + r, d * z[r]);
  }
  __syncthreads();
  const int lane = t & 63, wv = t >> 6;
  for (int r = wv; r < Cr; r += 4) {
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a = fmaf(W2[c * Cr + r], d2[c], a);
    a = wave_sum(a);
    if (lane == 0) {
      const float zz = z[r];
      const float d1 = relu6 ? ((zz > 0.f && zz < 6.f) ? a : 0.f) : (zz > 0.f ? a : 0.f);
      dz[r] = d1;
      if (db1) atomicAdd(db1 + r, d1);
    }
  }
  __syncthreads();
  if (t < C) {
    const float mean = pooled[(int64_t)b * C + t] * inv_p;
    float a = 0.f;
    for (int r = 0; r < Cr; ++r) { a = fmaf(W1[r * C + t], dz[r], a); atomicAdd(dW1 + r * C + t, dz[r] * mean); }
    dpool[(int64_t)b * C + t] = a * inv_p;
  }
}
extern "C" int mfnet_se_gate_fwd(const float* pooled_sum, int64_t P, const float* W1, const float* b1, const float* W2, const float* b2, int relu6,
                                 float* z, float* g, int64_t B, int64_t C, int64_t Cr, hipStream_t stream) {
  MF_CHECK_ARG(pooled_sum && W1 && W2 && z && g && C > 0 && C <= SE_MAXC && Cr > 0 && Cr <= SE_MAXR && P > 0 && B > 0, "se_gate_fwd: bad args");
  hipLaunchKernelGGL(se_gate_fwd_kernel, dim3((unsigned)B), dim3(256), 0, stream, pooled_sum, 1.f / (float)P, W1, b1, W2, b2, relu6, z, g, (int)C, (int)Cr);
  MF_LAUNCH_CHECK("se_gate_fwd");
  return MFNET_OK;
}
extern "C" int mfnet_se_gate_bwd(const float* dg, const float* pooled_sum, int64_t P, const float* W1, const float* W2, int relu6, const float* z, const float* g,
                                 float* dpool, float* dW1, float* db1, float* dW2, float* db2, int64_t B, int64_t C, int64_t Cr, hipStream_t stream) {
  MF_CHECK_ARG(dg && pooled_sum && W1 && W2 && z && g && dpool && dW1 && dW2 && C > 0 && C <= SE_MAXC && Cr > 0 && Cr <= SE_MAXR && P > 0 && B > 0, "se_gate_bwd: bad args");
  hipLaunchKernelGGL(se_gate_bwd_kernel, dim3((unsigned)B), dim3(256), 0, stream, dg, pooled_sum, 1.f / (float)P, W1, W2, relu6, z, g, dpool, dW1, db1, dW2, db2, (int)C, (int)Cr);
  MF_LAUNCH_CHECK("se_gate_bwd");
  return MFNET_OK;
}
// SEFusion (unetformer_mmsam.py:268-281): res = rx*gx[b] + ry*gy[b] ; rows of image b = [b*P, (b+1)*P)
__global__ __launch_bounds__(256) void se_scale_sum_kernel(const void* __restrict__ rx, int rx_dt, const void* __restrict__ ry, int ry_dt, const float* __restrict__ gx,
                                                           const float* __restrict__ gy, void* __restrict__ out, int out_dt, int64_t rows, int64_t P, int C4) {
  const int64_t total = rows * C4;
  GS_LOOP(i, total) {
    const int cg = (int)(i % C4);
    const int64_t b = (i / C4) / P;
    const float4 a = f4mul(ld4(rx, rx_dt, i), reinterpret_cast<const float4*>(gx)[b * C4 + cg]);
    st4(out, out_dt, i, f4fma(ld4(ry, ry_dt, i), reinterpret_cast<const float4*>(gy)[b * C4 + cg], a));
  }
}
// drx = dres*gx[b] + dpx[b] ; dry = dres*gy[b] + dpy[b]
__global__ __launch_bounds__(256) void se_bwd_apply_kernel(const void* __restrict__ dres, int dres_dt, const float* __restrict__ gx, const float* __restrict__ gy,
                                                           const float* __restrict__ dpx, const float* __restrict__ dpy, void* __restrict__ drx, int drx_dt,
                                                           void* __restrict__ dry, int dry_dt, int64_t rows, int64_t P, int C4) {
  const int64_t total = rows * C4;
  GS_LOOP(i, total) {
    const int cg = (int)(i % C4);
    const int64_t b = (i / C4) / P;
    const float4 d = ld4(dres, dres_dt, i);
    st4(drx, drx_dt, i, f4fma(d, reinterpret_cast<const float4*>(gx)[b * C4 + cg], reinterpret_cast<const float4*>(dpx)[b * C4 + cg]));
    st4(dry, dry_dt, i, f4fma(d, reinterpret_cast<const float4*>(gy)[b * C4 + cg], reinterpret_cast<const float4*>(dpy)[b * C4 + cg]));
  }
}
extern "C" int mfnet_se_scale_sum(const void* rx, int rx_dtype, const void* ry, int ry_dtype, const float* gx, const float* gy, void* out, int out_dtype,
                                  int64_t B, int64_t P, int64_t C, hipStream_t stream) {
  MF_CHECK_ARG(rx && ry && gx && gy && out && C % 4 == 0 && P > 0, "se_scale_sum: bad args");
  hipLaunchKernelGGL(se_scale_sum_kernel, dim3(grid_for(B * P * C / 4)), dim3(256), 0, stream, rx, rx_dtype, ry, ry_dtype, gx, gy, out, out_dtype, B * P, P, (int)(C / 4));
  MF_LAUNCH_CHECK("se_scale_sum");
  return MFNET_OK;
}
extern "C" int mfnet_se_bwd_apply(const void* dres, int dres_dtype, const float* gx, const float* gy, const float* dpx, const float* dpy, void* drx, int drx_dtype,
                                  void* dry, int dry_dtype, int64_t B, int64_t P, int64_t C, hipStream_t stream) {
  MF_CHECK_ARG(dres && gx && gy && dpx && dpy && drx && dry && C % 4 == 0 && P > 0, "se_bwd_apply: bad args");
  hipLaunchKernelGGL(se_bwd_apply_kernel, dim3(grid_for(B * P * C / 4)), dim3(256), 0, stream, dres, dres_dtype, gx, gy, dpx, dpy, drx, drx_dtype, dry, dry_dtype,
                     B * P, P, (int)(C / 4));
  MF_LAUNCH_CHECK("se_bwd_apply");
  return MFNET_OK;
}

// =============================================================================== FRH pixel + channel attention
// pc = xf * (sigmoid(dwo) + gate[b])   (unetformer_mmsam.py:349-351: pa(x)*x + ca(x)*x); all fp32 [B*P, C]
__global__ __launch_bounds__(256) void frh_gate_fwd_kernel(const float* __restrict__ xf, const float* __restrict__ dwo, const float* __restrict__ gate,
                                                           float* __restrict__ pc, int64_t rows, int64_t P, int C4) {
  const int64_t total = rows * C4;
  GS_LOOP(i, total) {
    const int cg = (int)(i % C4);
    const int64_t b = (i / C4) / P;
    const float4 x = reinterpret_cast<const float4*>(xf)[i], d = reinterpret_cast<const float4*>(dwo)[i];
    const float4 g = reinterpret_cast<const float4*>(gate)[b * C4 + cg];
    reinterpret_cast<float4*>(pc)[i] = make_float4(x.x * (sigmoidf_(d.x) + g.x), x.y * (sigmoidf_(d.y) + g.y), x.z * (sigmoidf_(d.z) + g.z), x.w * (sigmoidf_(d.w) + g.w));
  }
}
// dxf (direct part) = dpc*(sig + gate) [+ dpool[b] if given] ; ddwo = dpc*xf*sig*(1-sig)
__global__ __launch_bounds__(256) void frh_gate_bwd_kernel(const float* __restrict__ dpc, const float* __restrict__ xf, const float* __restrict__ dwo,
                                                           const float* __restrict__ gate, const float* __restrict__ dpool, float* __restrict__ dxf,
                                                           float* __restrict__ ddwo, int64_t rows, int64_t P, int C4) {
  const int64_t total = rows * C4;
  GS_LOOP(i, total) {
    const int cg = (int)(i % C4);
    const int64_t b = (i / C4) / P;
    const float4 x = reinterpret_cast<const float4*>(xf)[i], d = reinterpret_cast<const float4*>(dwo)[i], g = reinterpret_cast<const float4*>(gate)[b * C4 + cg];
    const float4 dp = reinterpret_cast<const float4*>(dpc)[i];
    const float4 s = make_float4(sigmoidf_(d.x), sigmoidf_(d.y), sigmoidf_(d.z), sigmoidf_(d.w));
    float4 o = make_float4(dp.x * (s.x + g.x), dp.y * (s.y + g.y), dp.z * (s.z + g.z), dp.w * (s.w + g.w));
    if (dpool) o = f4add(o, reinterpret_cast<const float4*>(dpool)[b * C4 + cg]);
    reinterpret_cast<float4*>(dxf)[i] = o;
    reinterpret_cast<float4*>(ddwo)[i] = make_float4(dp.x * x.x * s.x * (1.f - s.x), dp.y * x.y * s.y * (1.f - s.y), dp.z * x.z * s.z * (1.f - s.z), dp.w * x.w * s.w * (1.f - s.w));
  }
}
extern "C" int mfnet_frh_gate_fwd(const float* xf, const float* dwo, const float* gate, float* pc, int64_t B, int64_t P, int64_t C, hipStream_t stream) {
  MF_CHECK_ARG(xf && dwo && gate && pc && C % 4 == 0 && P > 0, "frh_gate_fwd: bad args");
  hipLaunchKernelGGL(frh_gate_fwd_kernel, dim3(grid_for(B * P * C / 4)), dim3(256), 0, stream, xf, dwo, gate, pc, B * P, P, (int)(C / 4));
  MF_LAUNCH_CHECK("frh_gate_fwd");
  return MFNET_OK;
}
extern "C" int mfnet_frh_gate_bwd(const float* dpc, const float* xf, const float* dwo, const float* gate, const float* dpool, float* dxf, float* ddwo,
                                  int64_t B, int64_t P, int64_t C, hipStream_t stream) {
  MF_CHECK_ARG(dpc && xf && dwo && gate && dxf && ddwo && C % 4 == 0 && P > 0, "frh_gate_bwd: bad args");
  hipLaunchKernelGGL(frh_gate_bwd_kernel, dim3(grid_for(B * P * C / 4)), dim3(256), 0, stream, dpc, xf, dwo, gate, dpool, dxf, ddwo, B * P, P, (int)(C / 4));
  MF_LAUNCH_CHECK("frh_gate_bwd");
  return MFNET_OK;
}

// =============================================================================== small elementwise helpers
// out = a + b (+ c)   fp32|bf16 inputs, fp32|bf16 output (gradient joins of the decoder)
__global__ __launch_bounds__(256) void add3_kernel(const void* __restrict__ a, int a_dt, const void* __restrict__ b, int b_dt, const void* __restrict__ c, int c_dt,
                                                   void* __restrict__ out, int out_dt, int64_t n4) {
  GS_LOOP(i, n4) {
    float4 v = f4add(ld4(a, a_dt, i), ld4(b, b_dt, i));
    if (c) v = f4add(v, ld4(c, c_dt, i));
    st4(out, out_dt, i, v);
  }
}
extern "C" int mfnet_add3(const void* a, int a_dtype, const void* b, int b_dtype, const void* c, int c_dtype, void* out, int out_dtype, int64_t n, hipStream_t stream) {
  MF_CHECK_ARG(a && b && out && n % 4 == 0, "add3: bad args");
  if (n <= 0) return MFNET_OK;
  hipLaunchKernelGGL(add3_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, a, a_dtype, b, b_dtype, c, c_dtype, out, out_dtype, n / 4);
  MF_LAUNCH_CHECK("add3");
  return MFNET_OK;
}
