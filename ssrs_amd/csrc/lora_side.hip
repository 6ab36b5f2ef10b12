// One pass over a bf16 gradient matrix for BOTH rank-r products of a LoRA layer's backward (loralib/layers.py:149-165,
// 244-268 differentiated): with Y = dL/d(out) [G*Mg, C],
//
//   p_part[strip][m][j]     = sum_{c in strip} Y[m,c] * Vt[g][j][c]      the row products  P = Y . B   (-> d(x A^T))
//   dv_part[g][split][c][j] = sum_{m in split} Y[m,c] * U[m][j]          the column products  dB = Y^T . u
//
// The first version ran them as two kernels (skinny_nt, skinny_tn_mfma) that each streamed Y from HBM, with 4 KB in flight
// per wave. Here a wave owns a strip of 256 columns x `rpw` rows: per 32-row step it requests 16 KB (the next step's loads are
// in flight while this one multiplies), copies it into four wave-private [32][64] LDS slabs (row-major, XOR-swizzled) and reads
// each slab twice: by rows (ds_read_b128) as the B operand of P, transposed (ds_read_b64_tr_b16) as the A operand of dV.
// No barrier in the loop: LDS is never shared between waves. HBM-bound: algorithmic bytes = G*Mg*C*2.
// Partial sums: P over the ceil(C/256) strips (summed by the consumer: mfnet_lora_mix_bwd / mfnet_reduce_splits), dV over the
// row splits (summed by the gradient delivery, like every other split weight gradient).
#include "common.h"

// (bx, by, bz) of (strips, ny splits, groups): the workgroup's place in the problem's grid - blockIdx of the one-problem kernel, a
// slice of a 1-D grid in the batch kernel. lds: [wave][slab] images, 64 KB.
template <int R, bool WANT_DV, bool WANT_P>
__device__ __forceinline__ void lora_side_body(unsigned char (*lds)[4][32 * 128], const bf16_t* __restrict__ Y, int64_t ldy,
                                               const float* __restrict__ U, int64_t ldu, const float* __restrict__ Vt, int64_t ldvt,
                                               int64_t vt_group_stride, int RP, int64_t Mg, int C, int rpw, float* __restrict__ dv_part,
                                               float* __restrict__ p_part, int64_t Mtot, int bx, int by, int bz, int ny) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int g = bz;
  const int cw = bx * 256;
  const int ns = (C - cw) >= 256 ? 4 : (C - cw) / 64;                              // 64-column slabs of this strip (wave-uniform)
  Y += (int64_t)g * Mg * ldy;
  if (WANT_DV) U += (int64_t)g * Mg * ldu;
  const int64_t r_begin = ((int64_t)by * 4 + wave) * rpw;
  int64_t r_end = r_begin + rpw;
  if (r_end > Mg) r_end = Mg;

  // A operand of P: Vt[j][c], j = fr (< RP), the lane's eight columns of each 32-column block; kept for the whole sweep
  bf16x8_t vtf[4][2];
  if (WANT_P) {
    const float* vg = Vt + (int64_t)g * vt_group_stride + (int64_t)(fr < RP ? fr : 0) * ldvt;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        union { bf16x8_t v; unsigned int w[4]; } f;
        f.w[0] = f.w[1] = f.w[2] = f.w[3] = 0u;
        if (fr < RP && s < ns) {
          const float* q = vg + cw + 64 * s + 32 * cb + 8 * fq;
          const float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
          f.w[0] = pack_bf16x2(a.x, a.y); f.w[1] = pack_bf16x2(a.z, a.w); f.w[2] = pack_bf16x2(b.x, b.y); f.w[3] = pack_bf16x2(b.z, b.w);
        }
        vtf[s][cb] = f.v;
      }
  }
  f32x4_t acc[4][4];          // dV: [slab][16-column block]
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[s][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // staging: lane copies 16-B chunk (lane & 7) of rows (lane >> 3) + 8*i of every slab
  const int srow = lane >> 3, sch = lane & 7;
  const int ucol = fr < R ? fr : 0;
  u32x4_t yv[4][4];
  float uv[8];
  auto issue = [&](int64_t m0) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s >= ns) break;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int64_t m = m0 + srow + 8 * i;
        m = m < r_end ? m : r_end - 1;
        yv[s][i] = *reinterpret_cast<const u32x4_t*>(Y + m * ldy + cw + 64 * s + sch * 8);
      }
    }
    if (WANT_DV) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t m = m0 + 16 * (j >> 2) + 4 * fq + (j & 3);      // k-slot j of lane group fq <-> this row of the step
        uv[j] = (m < r_end && fr < R) ? U[m * ldu + ucol] : 0.f;      // rows past the range contribute nothing
      }
    }
  };
  if (r_begin < r_end) issue(r_begin);
  for (int64_t m0 = r_begin; m0 < r_end; m0 += 32) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s >= ns) break;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = srow + 8 * i;
        *reinterpret_cast<u32x4_t*>(lds[wave][s] + row * 128 + ((sch ^ (row & 7)) << 4)) = yv[s][i];
      }
    }
    union { bf16x8_t v; unsigned int w[4]; } uh, ul;
    if (WANT_DV) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uh.w[i] = pack_bf16x2(uv[2 * i], uv[2 * i + 1]);
        ul.w[i] = pack_bf16x2(uv[2 * i] - bf16_lo(uh.w[i]), uv[2 * i + 1] - bf16_hi(uh.w[i]));
      }
    }
    if (m0 + 32 < r_end) issue(m0 + 32);
    f32x4_t accp[2];
    accp[0] = accp[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int trow = 4 * fq + (fr >> 2);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s >= ns) break;
      const unsigned char* img = lds[wave][s];
      if (WANT_DV) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const unsigned char* a = img + trow * 128 + (((2 * t + ((fr >> 1) & 1)) ^ (trow & 7)) << 4) + ((fr & 1) << 3);
          union { bf16x8_t v; s16x4_t h[2]; } yf;
          yf.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)a);
          yf.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(a + 16 * 128));
          acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf.v, uh.v, acc[s][t], 0, 0, 0);
          acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf.v, ul.v, acc[s][t], 0, 0, 0);
        }
      }
      if (WANT_P) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            const int row = mb * 16 + fr;
            const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(img + row * 128 + (((cb * 4 + fq) ^ (row & 7)) << 4));
            accp[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vtf[s][cb], b, accp[mb], 0, 0, 0);
          }
      }
    }
    if (WANT_P && fq < 2) {          // accp[mb][r] = P[m0 + 16 mb + fr][4 fq + r] over this strip
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const int64_t m = m0 + 16 * mb + fr;
        if (m < r_end)
          *reinterpret_cast<float4*>(p_part + ((int64_t)bx * Mtot + (int64_t)g * Mg + m) * 8 + 4 * fq) =
              make_float4(accp[mb][0], accp[mb][1], accp[mb][2], accp[mb][3]);
      }
    }
  }
  if (!WANT_DV) return;
  // the four waves hold the sums of four row ranges: add them through LDS, wave w finishes slab w.
  // region of a wave (its own slabs, 16 KB): [value 0..63][lane]
  __syncthreads();
  float* mine = reinterpret_cast<float*>(lds[wave][0]);
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[((s * 4 + t) * 4 + r) * 64 + lane] = acc[s][t][r];
  __syncthreads();
  if (wave < ns && fr < R) {
    float* out = dv_part + (((int64_t)g * ny + by) * C) * R;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = ((wave * 4 + t) * 4 + r) * 64 + lane;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += reinterpret_cast<const float*>(lds[w][0])[idx];
        const int c = cw + 64 * wave + 16 * t + 4 * fq + r;      // acc[s][t][r] = out[cw + 64 s + 16 t + 4 fq + r][fr]
        out[(int64_t)c * R + fr] = v;
      }
  }
}

template <int R, bool WANT_DV, bool WANT_P>
__global__ __launch_bounds__(256, 2) void lora_side_kernel(const bf16_t* __restrict__ Y, int64_t ldy, const float* __restrict__ U, int64_t ldu,
                                                           const float* __restrict__ Vt, int64_t ldvt, int64_t vt_group_stride, int RP,
                                                           int64_t Mg, int C, int rpw, float* __restrict__ dv_part, float* __restrict__ p_part,
                                                           int64_t Mtot) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4][4][32 * 128];      // [wave][slab]: 64 KB
  lora_side_body<R, WANT_DV, WANT_P>(lds, Y, ldy, U, ldu, Vt, ldvt, vt_group_stride, RP, Mg, C, rpw, dv_part, p_part, Mtot,
                                     blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y);
}

// Several column products dV = Y^T . U (no P) in one launch: the dA = x^T . da products of a step (one per LoRA layer, 36 in the
// ViT-B step) feed nothing but the gradient delivery, so the host collects them like the weight-gradient GEMMs (mfnet_gemm_tn_batch).
#define LS_BATCH 24
struct LsProb { const bf16_t* Y; const float* U; float* dv_part; int64_t ldy, ldu, Mg; int C, rpw, gx, gy, wg_begin, pad_; };
struct LsBatch { int n; LsProb p[LS_BATCH]; };
template <int R>
__global__ __launch_bounds__(256, 2) void lora_side_batch_kernel(const LsBatch b) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4][4][32 * 128];
  const int id = blockIdx.x;
  int i = 0;
  for (int j = 1; j < b.n; ++j)
    if (id >= b.p[j].wg_begin) i = j;
  const LsProb& q = b.p[i];
  const int local = id - q.wg_begin;
  const int bx = local % q.gx, t = local / q.gx, by = t % q.gy, bz = t / q.gy;
  lora_side_body<R, true, false>(lds, q.Y, q.ldy, q.U, q.ldu, nullptr, 0, 0, 0, q.Mg, q.C, q.rpw, q.dv_part, nullptr, 0, bx, by, bz, q.gy);
}

static void lora_side_geometry(int64_t Mg, int64_t C, int64_t* strips, int64_t* splits, int* rpw) {
  *strips = ceil_div64(C, 256);
  *rpw = *strips <= 4 ? 32 : 64;
  *splits = ceil_div64(Mg, 4 * (int64_t)*rpw);
}
// row splits per group of mfnet_lora_side (the dV partial buffer holds n_groups * splits * C * R floats)
extern "C" int64_t mfnet_lora_side_splits(int64_t Mg, int64_t C) {
  if (Mg <= 0 || C <= 0) return 0;
  int64_t strips, splits; int rpw;
  lora_side_geometry(Mg, C, &strips, &splits, &rpw);
  return splits;
}

extern "C" int mfnet_lora_side(const void* Y, int64_t ldy, int64_t Mg, int64_t C, int n_groups, const float* U, int64_t ldu, int R,
                               float* dv_part, const float* Vt, int64_t ldvt, int64_t vt_group_stride, int RP, float* p_part,
                               hipStream_t stream) {
  MF_CHECK_ARG(Y && (U || Vt), "lora_side: null pointer (needs U and / or Vt)");
  MF_CHECK_ARG(C > 0 && C % 64 == 0 && ldy % 8 == 0 && ldy >= C && (uintptr_t)Y % 16 == 0, "lora_side: C=%lld must be a multiple of 64 (ldy of 8, Y 16-byte aligned)", (long long)C);
  MF_CHECK_ARG(n_groups >= 1 && n_groups <= 64 && Mg >= 0, "lora_side: bad group args");
  if (U) MF_CHECK_ARG(dv_part && (R == 4 || R == 8) && ldu >= R, "lora_side: U needs dv_part and R = 4 or 8 (R=%d)", R);
  if (Vt) MF_CHECK_ARG(p_part && RP >= 1 && RP <= 8 && ldvt >= C && ldvt % 4 == 0 && vt_group_stride % 4 == 0 && (uintptr_t)Vt % 16 == 0 && (uintptr_t)p_part % 16 == 0,
                       "lora_side: Vt needs p_part, 1 <= RP <= 8, ldvt >= C (multiples of 4, 16-byte aligned)");
  if (Mg == 0) return MFNET_OK;
  int64_t strips, splits; int rpw;
  lora_side_geometry(Mg, C, &strips, &splits, &rpw);
  MF_CHECK_ARG(splits <= 65535, "lora_side: too many row splits");
  const dim3 grid((unsigned)strips, (unsigned)splits, (unsigned)n_groups);
  const int64_t Mtot = Mg * n_groups;
#define LS_LAUNCH(RR, DV, PP) hipLaunchKernelGGL((lora_side_kernel<RR, DV, PP>), grid, dim3(256), 0, stream, (const bf16_t*)Y, ldy, U, ldu, Vt, ldvt, \
                                                 vt_group_stride, RP, Mg, (int)C, rpw, dv_part, p_part, Mtot)
  if (U && Vt) { if (R == 4) LS_LAUNCH(4, true, true); else LS_LAUNCH(8, true, true); }
  else if (U) { if (R == 4) LS_LAUNCH(4, true, false); else LS_LAUNCH(8, true, false); }
  else LS_LAUNCH(8, false, true);
#undef LS_LAUNCH
  MF_LAUNCH_CHECK("lora_side");
  return MFNET_OK;
}

// desc: 8 int64 per problem - Y, ldy, Mg (rows per group), C, n_groups, U, ldu, dv_part; all with the same R (4 or 8). Per problem
// exactly mfnet_lora_side(Y, ldy, Mg, C, n_groups, U, ldu, R, dv_part, NULL, ...).
extern "C" int mfnet_lora_side_batch(const int64_t* desc, int64_t n, int R, hipStream_t stream) {
  MF_CHECK_ARG(desc || n == 0, "lora_side_batch: null descriptor table");
  MF_CHECK_ARG(n >= 0 && n <= 4096 && (R == 4 || R == 8), "lora_side_batch: bad problem count / R = %d", R);
  for (int64_t i = 0; i < n; ++i) {
    const int64_t* d = desc + 8 * i;
    MF_CHECK_ARG(d[0] && d[5] && d[7] && d[3] > 0 && d[3] % 64 == 0 && d[1] % 8 == 0 && d[1] >= d[3] && d[0] % 16 == 0 && d[4] >= 1 && d[4] <= 64 && d[2] > 0 && d[6] >= R,
                 "lora_side_batch: bad problem %lld", (long long)i);
  }
  for (int64_t i0 = 0; i0 < n; i0 += LS_BATCH) {
    LsBatch b;
    b.n = (int)(n - i0 < LS_BATCH ? n - i0 : LS_BATCH);
    int64_t total = 0;
    for (int j = 0; j < b.n; ++j) {
      const int64_t* d = desc + 8 * (i0 + j);
      LsProb& q = b.p[j];
      int64_t strips, splits; int rpw;
      lora_side_geometry(d[2], d[3], &strips, &splits, &rpw);
      MF_CHECK_ARG(splits <= 65535, "lora_side_batch: too many row splits");
      q.Y = (const bf16_t*)d[0]; q.ldy = d[1]; q.Mg = d[2]; q.C = (int)d[3]; q.U = (const float*)d[5]; q.ldu = d[6]; q.dv_part = (float*)d[7];
      q.rpw = rpw; q.gx = (int)strips; q.gy = (int)splits; q.wg_begin = (int)total; q.pad_ = 0;
      total += strips * splits * d[4];
      MF_CHECK_ARG(total < (1ll << 30), "lora_side_batch: too many workgroups");
    }
    for (int j = b.n; j < LS_BATCH; ++j) b.p[j] = b.p[0];
    if (R == 4) hipLaunchKernelGGL(lora_side_batch_kernel<4>, dim3((unsigned)total), dim3(256), 0, stream, b);
    else hipLaunchKernelGGL(lora_side_batch_kernel<8>, dim3((unsigned)total), dim3(256), 0, stream, b);
    MF_LAUNCH_CHECK("lora_side_batch");
  }
  return MFNET_OK;
}
