// bf16 MFMA GEMM with fused epilogues for the stacked-modality token matrix:
//
//   C[m,n] = sum_k A[m,k] * W[n,k]                       (A: [M,K] bf16, W: [N,K] bf16, fp32 accumulate)
//          + bias[n]
//          + lr_scale * sum_{j<R} U[m,j] * V[g(m)][n,j]   (low-rank update: LoRA fwd / LoRA dgrad; fp32 U, V rounded to
//                                                           bf16 and accumulated by one extra MFMA k-step)
//          -> act: none | GELU (optionally also stores the pre-activation) | multiply by GELU'(aux[m,n])
//          + residual[m % res_row_mod, n]                 (fp32; may alias the fp32 output)
//          -> out (bf16 or fp32)
//
// It serves every dense contraction of the ViT blocks (lora_block.py:169-196 qkv/proj, :104-108 MLP,
// loralib/layers.py:149-165,244-268 low-rank terms), their dgrads (W passed pre-transposed), the
// patch-embed im2col GEMM (image_encoder.py:189-197) and the neck convs (:104-120).
//
// Design (gfx950): 128x128x64 tiles, 256 threads = 4 waves in 2x2, 64x64 per wave as 4x4
// mfma_f32_16x16x32_bf16 tiles; A/W tiles land in LDS by global_load_lds (16 B/lane) into a
// lane-linear image, XOR-swizzled on the SOURCE address so ds_read_b128 fragment reads are
// conflict-free; double-buffered, one barrier per K-step. The weight rows are the MFMA "A" operand
// and are read in a permuted order so every lane ends up owning 16 CONSECUTIVE output columns of one
// row: the epilogue loads bias/V/residual and stores C with 16-byte accesses. Workgroup ids are
// remapped so each XCD walks a contiguous run of tiles (shared A panel in its private L2).
#include "common.h"
#include "gemm_params.h"
#include <string.h>

#define BM 128
#define BN 128
#define BK 64



#ifdef GEMM_STAMPS
unsigned long long* g_stamps = nullptr;
int g_stagger = 0;
extern "C" void mfnet_diag_set_stamps(unsigned long long* p) { g_stamps = p; }
extern "C" void mfnet_diag_set_stagger(int ticks) { g_stagger = ticks; }
#endif
__device__ __forceinline__ int swz_x(int row) { return row & 7; }
__device__ __forceinline__ int swz_w(int row) { return (((row >> 4) & 3) << 1) | ((row >> 1) & 1); }

// ACT = GemmParams::act. The activations
// and low-rank epilogues are compiled into separate instantiations: one kernel with everything is ~100 KB of
// code, more than the 64 KB instruction cache a CU pair shares.
// CONV: implicit-GEMM 3x3 convolution (stride 1, zero padding 1) of a channels-last map: K-tile kt is channel block kt % conv_cb
// of tap kt / conv_cb, and the A rows of that tile are the map rows of the pixels shifted by the tap - fetched by the same
// global_load_lds as a plain A tile, from a row of zeros where the shifted pixel falls outside the image. No column tensor
// (the 9x copy im2col3x3 wrote and this kernel, the weight-gradient product and col2im3x3 read back) exists any more.
__device__ __attribute__((aligned(128))) unsigned char g_zero_row[128];     // (zero-initialised device memory, never written)

template <int ACT, bool LORA, bool CONV = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * (BM + BN) * BK * 2];  // 64 KiB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  MF_STAMP(0);
#ifdef GEMM_STAMPS
  if (p.stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512) {        // experiment: second resident workgroup of each CU starts late
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < p.stagger) __builtin_amdgcn_s_sleep(32);
  }
  if (p.stamps && threadIdx.x == 0) {
    p.stamps[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_ID
    p.stamps[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);   // XCC_ID
  }
#endif

  // XCD-aware bijective remap of the workgroup id (8 XCDs, round-robin dispatch)
  const int nwg = p.tiles_m * p.tiles_n;
  int wg;
  {
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int tm = wg / p.tiles_n, tn = wg - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  // modality group of this row tile (uniform per workgroup: the host splits launches that would straddle)
  const int grp = p.n_groups > 1 ? (int)(((unsigned)m0 / (unsigned)p.rows_per_group) % (unsigned)p.n_groups) : 0;
  const bf16_t* Wg = p.W + (int64_t)grp * p.w_group_stride;
  const float* biasg = p.bias ? p.bias + (int64_t)grp * p.bias_group_stride : nullptr;

  // ---- staging addresses: this wave fills rows [wave*32, wave*32+32) of both tiles, 8 rows/instr
  const int srow = lane >> 3;        // 0..7 within an 8-row group
  const int sch = lane & 7;          // destination chunk (16 B) inside the 128-B row
  const bf16_t* a_src[4];
  const bf16_t* w_src[4];
  // K % 64 == 32: the last K-tile holds one 32-wide k-step. Lanes whose SOURCE chunk would lie in the
  // missing half re-read the valid half instead (the LDS half they fill is never consumed), so no
  // access leaves the operand rows.
  const int nt = (p.K + BK - 1) / BK;
  const bool half_tail = (p.K & (BK - 1)) != 0;
  int a_tail[4], w_tail[4];
  int ay[4], ax[4];                  // CONV: image coordinates of this lane's four A rows
  const bf16_t* zsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + srow;
#ifdef GEMM_EXP_SAMETILE
    int gm = row, gn = row;
#else
    int gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;
    int gn = n0 + row; gn = gn < p.N ? gn : p.N - 1;
#endif
    const int ca = sch ^ swz_x(row), cw = sch ^ swz_w(row);
    a_src[i] = p.A + (int64_t)gm * p.lda + (ca << 3);
    w_src[i] = Wg + (int64_t)gn * p.ldw + (cw << 3);
    a_tail[i] = (half_tail && ca >= 4) ? -32 : 0;
    w_tail[i] = (half_tail && cw >= 4) ? -32 : 0;
    if (CONV) {
      const int pix = gm % (p.conv_H * p.conv_W);
      ay[i] = pix / p.conv_W; ax[i] = pix - ay[i] * p.conv_W;
      zsrc[i] = reinterpret_cast<const bf16_t*>(g_zero_row) + (ca << 3);
    }
  }
  auto stage = [&](int kt, int buf) {
    unsigned char* base = lds + buf * ((BM + BN) * BK * 2);
    const bool last = kt == nt - 1;
    int dyy = 0, dxx = 0;
    int64_t aoff = (int64_t)kt * BK;
    if (CONV) {
      const int tap = kt / p.conv_cb, t3 = tap / 3;
      dyy = t3 - 1; dxx = tap - 3 * t3 - 1;
      aoff = (int64_t)(dyy * p.conv_W + dxx) * p.lda + (kt - tap * p.conv_cb) * BK;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16_t* asrc = a_src[i] + aoff + (last ? a_tail[i] : 0);
      if (CONV && !((unsigned)(ay[i] + dyy) < (unsigned)p.conv_H && (unsigned)(ax[i] + dxx) < (unsigned)p.conv_W)) asrc = zsrc[i];
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)asrc,
                                       (lds_ptr_t)(base + (wave * 4 + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w_src[i] + kt * BK + (last ? w_tail[i] : 0)),
                                       (lds_ptr_t)(base + BM * BK * 2 + (wave * 4 + i) * 1024), 16, 0, 0);
    }
  };

  // ---- fragment read offsets (bytes, without the per-k-step chunk XOR)
  const int fr = lane & 15, fq = lane >> 4;
  int x_off[4], x_sw[4], w_off[4], w_sw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int xr = wm * 64 + i * 16 + fr;
    x_off[i] = xr * 128; x_sw[i] = swz_x(xr);
    const int wr = wn * 64 + 16 * (fr >> 2) + 4 * i + (fr & 3);
    w_off[i] = BM * BK * 2 + wr * 128; w_sw[i] = swz_w(wr);
  }

  f32x4_t acc[4][4];   // [ni][mi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- epilogue operands that do not depend on the accumulators are requested BEFORE the main loop, so
  // their latency hides behind it: bias, and the raw fp32 rows of the low-rank factors.
  // Lane owns rows m(mi) = m0+wm*64+mi*16+fr and columns nb..nb+15 of the output.
  const int nb = n0 + wn * 64 + 16 * fq;
  int mrow[4];
  bool mok[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    mrow[mi] = m0 + wm * 64 + mi * 16 + fr;
    mok[mi] = mrow[mi] < p.M;
  }
  float4 bias4[4];
  if (biasg) {
    const int nbc = nb < p.N ? nb : p.N - 16;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) bias4[ni] = *reinterpret_cast<const float4*>(biasg + nbc + 4 * ni);
  }
  // Low-rank update = one extra MFMA k-step per 8 ranks whose 32 k-slots hold the rank index three times:
  // slots [0,8) U_hi.V_hi, [8,16) U_lo.V_hi, [16,24) U_hi.V_lo (hi = bf16(x), lo = bf16(x - hi)), so the fp32 U (x lr_scale)
  // and V keep ~16 mantissa bits although the product runs on the bf16 matrix cores; lane quarter fq owns product fq of
  // the pass's ranks. R <= 8 (the default LoRA rank 4: stacked factors of 8 columns) is one pass whose factor rows are
  // requested here, ahead of the main loop; LoRA ranks up to 16 (R <= 32) take up to four, the later ones fetched in the
  // epilogue. The modality group is uniform per workgroup (the host splits launches that would straddle).
  const int R = p.R;
  float4 uraw[4][2], vraw[4][2];
  const float* vg = LORA ? p.V + (int64_t)grp * p.v_group_stride : nullptr;
  auto load_factors = [&](int lk0) {
    // unconditional loads (a masked-off second half re-reads the first): no branch, no wait between them
    const int lk1 = lk0 + 4 < R ? lk0 + 4 : lk0;
    auto load_raw = [&](float4 (&dst)[2], const float* src) {
      dst[0] = *reinterpret_cast<const float4*>(src + lk0);
      dst[1] = *reinterpret_cast<const float4*>(src + lk1);
    };
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) load_raw(uraw[mi], p.U + (int64_t)(mok[mi] ? mrow[mi] : p.M - 1) * p.ldu);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      int n = n0 + wn * 64 + 16 * (fr >> 2) + 4 * ni + (fr & 3);      // W-row order of the accumulator tiles
      n = n < p.N ? n : p.N - 1;
      load_raw(vraw[ni], vg + (int64_t)n * R);
    }
  };
  if (LORA) load_factors(0);

  int t_begin = 0, t_end = nt;
  if (p.k_splits > 1) {
    t_begin = (int)((int64_t)nt * blockIdx.y / p.k_splits);
    t_end = (int)((int64_t)nt * (blockIdx.y + 1) / p.k_splits);
  }
  if (t_begin < t_end) stage(t_begin, 0);
  for (int t = t_begin; t < t_end; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef GEMM_STAMPS
    if (t == t_begin) MF_STAMP(1);
#endif
    if (t + 1 < t_end) stage(t + 1, (t + 1 - t_begin) & 1);
    const unsigned char* base = lds + ((t - t_begin) & 1) * ((BM + BN) * BK * 2);
    const int nks = (half_tail && t == nt - 1) ? 1 : 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks >= nks) break;
      bf16x8_t xf[4], wf[4];
      const int ch = fq + 4 * ks;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xf[i] = *reinterpret_cast<const bf16x8_t*>(base + x_off[i] + ((ch ^ x_sw[i]) << 4));
        wf[i] = *reinterpret_cast<const bf16x8_t*>(base + w_off[i] + ((ch ^ w_sw[i]) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
    }
  }

  // ---- epilogue
  MF_STAMP(2);
  if (LORA) {
    const float on = fq <= 2 ? 1.0f : 0.0f;
    for (int lk0 = 0; lk0 < R; lk0 += 8) {
      if (lk0 > 0) load_factors(lk0);
      const float tail = lk0 + 4 < R ? 1.0f : 0.0f;
      // branch-free conversion: `lo` and a zero scale are lane selects
      auto conv = [&](const float4 (&raw)[2], float sc, bool lo) -> bf16x8_t {
        const float s1 = sc * tail;
        const float x[8] = {raw[0].x * sc, raw[0].y * sc, raw[0].z * sc, raw[0].w * sc,
                            raw[1].x * s1, raw[1].y * s1, raw[1].z * s1, raw[1].w * s1};
        union { bf16x8_t v; unsigned int w[4]; } r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned int h = pack_bf16x2(x[2 * i], x[2 * i + 1]);
          const unsigned int l = pack_bf16x2(x[2 * i] - bf16_lo(h), x[2 * i + 1] - bf16_hi(h));
          r.w[i] = lo ? l : h;
        }
        return r.v;
      };
      bf16x8_t vf[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) vf[ni] = conv(vraw[ni], on, fq == 2);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const bf16x8_t um = conv(uraw[mi], on * p.lr_scale, fq == 1);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[ni], um, acc[ni][mi], 0, 0, 0);
      }
    }
  }
  if (nb >= p.N) return;
  if (p.k_splits > 1) {      // split-K partial: plain stores into slice blockIdx.y of out [k_splits][M][ldc]
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      if (!mok[mi]) continue;
      float4* dst = reinterpret_cast<float4*>((float*)p.out + ((int64_t)blockIdx.y * p.M + mrow[mi]) * p.ldc + nb);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) dst[ni] = make_float4(acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]);
    }
    return;
  }
  if (p.bias) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        acc[ni][mi][0] += bias4[ni].x; acc[ni][mi][1] += bias4[ni].y; acc[ni][mi][2] += bias4[ni].z; acc[ni][mi][3] += bias4[ni].w;
      }
  }
  // Two row tiles per round: their aux / residual loads are issued together ahead of the stores (which may
  // alias `residual`), so each kind of operand costs two memory round trips per workgroup instead of four.
#pragma unroll
  for (int mh = 0; mh < 4; mh += 2) {
    int mc[2];
#pragma unroll
    for (int mj = 0; mj < 2; ++mj) mc[mj] = mok[mh + mj] ? mrow[mh + mj] : p.M - 1;
    if (p.out_pre) {   // value before activation / residual (GELU pre-activation, adapter input, ...)
#pragma unroll
      for (int mj = 0; mj < 2; ++mj) {
        const int mi = mh + mj;
        if (!mok[mi]) continue;
        u32x4_t w0, w1;
        w0.x = pack_bf16x2(acc[0][mi][0], acc[0][mi][1]); w0.y = pack_bf16x2(acc[0][mi][2], acc[0][mi][3]);
        w0.z = pack_bf16x2(acc[1][mi][0], acc[1][mi][1]); w0.w = pack_bf16x2(acc[1][mi][2], acc[1][mi][3]);
        w1.x = pack_bf16x2(acc[2][mi][0], acc[2][mi][1]); w1.y = pack_bf16x2(acc[2][mi][2], acc[2][mi][3]);
        w1.z = pack_bf16x2(acc[3][mi][0], acc[3][mi][1]); w1.w = pack_bf16x2(acc[3][mi][2], acc[3][mi][3]);
        u32x4_t* dst = reinterpret_cast<u32x4_t*>(p.out_pre + (int64_t)mrow[mi] * p.ldpre + nb);
        dst[0] = w0; dst[1] = w1;
      }
    }
    if (ACT == 1) {
#pragma unroll
      for (int mj = 0; mj < 2; ++mj)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[ni][mh + mj][r] = gelu_f(acc[ni][mh + mj][r]);
    } else if (ACT == 2 || ACT == 4) {
      u32x4_t ax[2][2];
#pragma unroll
      for (int mj = 0; mj < 2; ++mj) {
        const u32x4_t* ap = reinterpret_cast<const u32x4_t*>(p.aux + (int64_t)mc[mj] * p.ldaux + nb);
        ax[mj][0] = ap[0]; ax[mj][1] = ap[1];
      }
#pragma unroll
      for (int mj = 0; mj < 2; ++mj) {
        const int mi = mh + mj;
        const unsigned int aw[8] = {ax[mj][0].x, ax[mj][0].y, ax[mj][0].z, ax[mj][0].w, ax[mj][1].x, ax[mj][1].y, ax[mj][1].z, ax[mj][1].w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {        // aux words 2*ni, 2*ni+1 cover accumulator tile ni
          const float lo = bf16_lo(aw[i]), hi = bf16_hi(aw[i]);
          float v0 = acc[i >> 1][mi][2 * (i & 1)], v1 = acc[i >> 1][mi][2 * (i & 1) + 1];
          if (ACT == 2) { v0 *= gelu_grad_f(lo); v1 *= gelu_grad_f(hi); }
          else { v0 = (lo > 0.f && lo < 6.f) ? v0 : 0.f; v1 = (hi > 0.f && hi < 6.f) ? v1 : 0.f; }
          acc[i >> 1][mi][2 * (i & 1)] = v0; acc[i >> 1][mi][2 * (i & 1) + 1] = v1;
        }
      }
    } else if (ACT == 3) {
#pragma unroll
      for (int mj = 0; mj < 2; ++mj)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[ni][mh + mj][r] = fminf(fmaxf(acc[ni][mh + mj][r], 0.f), 6.f);
    }
    if (p.residual) {
      float4 rs[2][4];
#pragma unroll
      for (int mj = 0; mj < 2; ++mj) {
        const int rr = p.res_row_mod > 0 ? (int)((unsigned)mc[mj] % (unsigned)p.res_row_mod) : mc[mj];
        const float4* rp = reinterpret_cast<const float4*>(p.residual + (int64_t)rr * p.ldres + nb);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) rs[mj][ni] = rp[ni];
      }
#pragma unroll
      for (int mj = 0; mj < 2; ++mj)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          f32x4_t& a = acc[ni][mh + mj];
          a[0] += rs[mj][ni].x; a[1] += rs[mj][ni].y; a[2] += rs[mj][ni].z; a[3] += rs[mj][ni].w;
        }
    }
    if (ACT == 5) {
#pragma unroll
      for (int mj = 0; mj < 2; ++mj)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[ni][mh + mj][r] = fminf(fmaxf(acc[ni][mh + mj][r], 0.f), 6.f);
    }
#pragma unroll
    for (int mj = 0; mj < 2; ++mj) {
      const int mi = mh + mj;
      if (!mok[mi]) continue;
      const int64_t m = mrow[mi];
      if (p.out_dtype == MFNET_BF16) {
        u32x4_t w0, w1;
        w0.x = pack_bf16x2(acc[0][mi][0], acc[0][mi][1]); w0.y = pack_bf16x2(acc[0][mi][2], acc[0][mi][3]);
        w0.z = pack_bf16x2(acc[1][mi][0], acc[1][mi][1]); w0.w = pack_bf16x2(acc[1][mi][2], acc[1][mi][3]);
        w1.x = pack_bf16x2(acc[2][mi][0], acc[2][mi][1]); w1.y = pack_bf16x2(acc[2][mi][2], acc[2][mi][3]);
        w1.z = pack_bf16x2(acc[3][mi][0], acc[3][mi][1]); w1.w = pack_bf16x2(acc[3][mi][2], acc[3][mi][3]);
        u32x4_t* dst = reinterpret_cast<u32x4_t*>((bf16_t*)p.out + m * p.ldc + nb);
        dst[0] = w0; dst[1] = w1;
      } else {
        float4* dst = reinterpret_cast<float4*>((float*)p.out + m * p.ldc + nb);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) dst[ni] = make_float4(acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]);
      }
    }
  }
#ifdef GEMM_STAMPS
  MF_STAMP(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  MF_STAMP(4);
#endif
}

static int gemm_impl(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t w_group_stride, int64_t M, int64_t N,
                     int64_t K, const float* bias, int64_t bias_group_stride, const float* U, int64_t ldu, const float* V,
                     int64_t v_group_stride, int64_t rows_per_group, int n_groups, int R,
                     float lr_scale, int act, const void* aux, int64_t ldaux, void* out_pre,
                     int64_t ldpre, const float* residual, int64_t ldres, int64_t res_row_mod,
                     void* out, int out_dtype, int64_t ldc, int k_splits, hipStream_t stream, int tile = 0) {
  MF_CHECK_ARG(A && W && out, "gemm_bf16: null operand");
  MF_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gemm_bf16: bad shape M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  MF_CHECK_ARG(K % 32 == 0, "gemm_bf16: K=%lld must be a multiple of 32", (long long)K);
  MF_CHECK_ARG(N % 16 == 0, "gemm_bf16: N=%lld must be a multiple of 16", (long long)N);
  MF_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K, "gemm_bf16: lda/ldw must be >= K and multiples of 8");
  MF_CHECK_ARG(ldc >= N && ldc % 8 == 0, "gemm_bf16: bad ldc");
  MF_CHECK_ARG(((uintptr_t)A | (uintptr_t)W | (uintptr_t)out) % 16 == 0, "gemm_bf16: operands must be 16-byte aligned");
  MF_CHECK_ARG(out_dtype == MFNET_BF16 || out_dtype == MFNET_F32, "gemm_bf16: bad out dtype");
  MF_CHECK_ARG(act >= 0 && act <= 5, "gemm_bf16: bad act");
  MF_CHECK_ARG((act != 2 && act != 4) || (aux && ldaux >= N && ldaux % 8 == 0), "gemm_bf16: act=2/4 needs aux");
  MF_CHECK_ARG(!out_pre || (act != 2 && act != 4 && ldpre >= N && ldpre % 8 == 0), "gemm_bf16: bad out_pre");
  MF_CHECK_ARG(!residual || (ldres >= N && ldres % 4 == 0), "gemm_bf16: bad residual ld");
  if (U) MF_CHECK_ARG(V && R > 0 && R <= GEMM_MAX_R && R % 4 == 0 && ldu >= R && ldu % 4 == 0, "gemm_bf16: bad low-rank args (R=%d)", R);
  const bool grouped = U || w_group_stride != 0 || bias_group_stride != 0;
  if (grouped) MF_CHECK_ARG(n_groups >= 1 && (n_groups == 1 || rows_per_group >= 16), "gemm_bf16: bad group args (rows_per_group must be >= 16)");
  if (!grouped) n_groups = 1;
  MF_CHECK_ARG(w_group_stride % 8 == 0 && bias_group_stride % 4 == 0 && w_group_stride >= 0 && bias_group_stride >= 0, "gemm_bf16: group strides of W / bias must be multiples of 8 / 4 elements");
  MF_CHECK_ARG(k_splits == 1 || (w_group_stride == 0 && bias_group_stride == 0), "gemm_bf16: split-K takes no grouped weights");
  MF_CHECK_ARG(M < (1ll << 31) && N < (1ll << 31), "gemm_bf16: shape too large");
  MF_CHECK_ARG(k_splits >= 1 && k_splits <= 1024, "gemm_bf16: bad k_splits");
  MF_CHECK_ARG(k_splits == 1 || (out_dtype == MFNET_F32 && !bias && !U && act == 0 && !out_pre && !residual),
               "gemm_bf16: split-K supports only the plain fp32 accumulate-into-out epilogue");
  if (M == 0) return MFNET_OK;
  if (grouped && n_groups > 1 && rows_per_group % BM != 0) {
    // The kernel takes the modality group per 128-row tile. Row blocks that do not end on a tile boundary
    // are launched one by one (never the case for MFNet's 256-pixel tiles: rows_per_group = B * 256).
    MF_CHECK_ARG(res_row_mod == 0, "gemm_bf16: grouped low-rank update with rows_per_group %% 128 != 0 cannot take res_row_mod");
    const size_t osz = out_dtype == MFNET_BF16 ? 2 : 4;
    for (int64_t r0 = 0, b = 0; r0 < M; r0 += rows_per_group, ++b) {
      const int64_t rows = M - r0 < rows_per_group ? M - r0 : rows_per_group;
      const int gsel = (int)(b % n_groups);
      const int rc = gemm_impl((const char*)A + r0 * lda * 2, lda, (const char*)W + gsel * w_group_stride * 2, ldw, 0, rows, N, K,
                                     bias ? bias + gsel * bias_group_stride : nullptr, 0, U ? U + r0 * ldu : nullptr, ldu,
                                     V ? V + gsel * v_group_stride : nullptr, 0, 0, 1, R, lr_scale, act,
                                     aux ? (const char*)aux + r0 * ldaux * 2 : nullptr, ldaux,
                                     out_pre ? (char*)out_pre + r0 * ldpre * 2 : nullptr, ldpre,
                                     residual ? residual + r0 * ldres : nullptr, ldres, 0,
                                     (char*)out + r0 * ldc * osz, out_dtype, ldc, 1, stream, tile);
      if (rc) return rc;
    }
    return MFNET_OK;
  }
  GemmParams p;
  p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.bias = bias;
  p.U = U; p.ldu = ldu; p.V = V; p.v_group_stride = v_group_stride; p.rows_per_group = rows_per_group;
  p.n_groups = n_groups; p.R = R; p.lr_scale = lr_scale; p.act = act;
  p.aux = (const bf16_t*)aux; p.ldaux = ldaux; p.out_pre = (bf16_t*)out_pre; p.ldpre = ldpre;
  p.residual = residual; p.ldres = ldres; p.res_row_mod = res_row_mod;
  p.out = out; p.out_dtype = out_dtype; p.ldc = ldc;
  p.tiles_m = (int)ceil_div64(M, BM); p.tiles_n = (int)ceil_div64(N, BN);
  const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  MF_CHECK_ARG(nwg < (1ll << 31), "gemm_bf16: too many tiles");
  p.k_splits = k_splits;
  p.w_group_stride = w_group_stride; p.bias_group_stride = bias_group_stride;
  p.conv_H = p.conv_W = p.conv_cb = 0;
#ifdef GEMM_STAMPS
  p.stamps = g_stamps; p.stagger = g_stagger;
#endif
  // the encoder's big token matrices go to the 256-row kernel (gemm_big.hip) when their shape is one of its
  const int big = gemm_big_launch(p, act, U != nullptr, tile, stream);
  if (big == 1) { MF_LAUNCH_CHECK("gemm_bf16 (256-row tiles)"); return MFNET_OK; }
  MF_CHECK_ARG(tile <= 0, "gemm_bf16: tile=%d asked for, but M=%lld N=%lld K=%lld is not a shape of the 256-row kernel (M %% 256, K %% 64, N %% (32*tile))",
               tile, (long long)M, (long long)N, (long long)K);
  const dim3 grid((unsigned)nwg, (unsigned)k_splits);
#define MF_GEMM_LAUNCH(A_, L_) hipLaunchKernelGGL((gemm_bf16_kernel<A_, L_>), grid, dim3(256), 0, stream, p)
#define MF_GEMM_CASE(A_) case A_: if (U) MF_GEMM_LAUNCH(A_, true); else MF_GEMM_LAUNCH(A_, false); break;
  switch (act) { MF_GEMM_CASE(0) MF_GEMM_CASE(1) MF_GEMM_CASE(2) MF_GEMM_CASE(3) MF_GEMM_CASE(4) MF_GEMM_CASE(5) }
#undef MF_GEMM_CASE
#undef MF_GEMM_LAUNCH
  MF_LAUNCH_CHECK("gemm_bf16");
  return MFNET_OK;
}

// 3x3 / stride 1 / zero-pad 1 convolution of a channels-last bf16 map as an implicit GEMM: x [B*H*W, C] (row stride ldx), w [N, 9*C]
// with k = tap*C + c (tap = 3*ky + kx), out [B*H*W, N] bf16 or fp32. The data gradient is the same call on dy with the weights
// flipped and transposed ([C, 9*N], k = tap'*N + n, tap' = 8 - tap).
extern "C" int mfnet_conv3x3_bf16(const void* x, int64_t ldx, const void* w, int64_t ldw, int64_t B, int64_t H, int64_t W, int64_t C,
                                  int64_t N, void* out, int out_dtype, int64_t ldc, hipStream_t stream) {
  MF_CHECK_ARG(x && w && out, "conv3x3_bf16: null operand");
  MF_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 64 == 0, "conv3x3_bf16: bad geometry (channels must be a multiple of 64, got %lld)", (long long)C);
  MF_CHECK_ARG(N > 0 && N % 16 == 0, "conv3x3_bf16: N=%lld must be a multiple of 16", (long long)N);
  MF_CHECK_ARG(ldx >= C && ldx % 8 == 0 && ldw >= 9 * C && ldw % 8 == 0 && ldc >= N && ldc % 8 == 0, "conv3x3_bf16: bad leading dimensions");
  MF_CHECK_ARG(((uintptr_t)x | (uintptr_t)w | (uintptr_t)out) % 16 == 0, "conv3x3_bf16: operands must be 16-byte aligned");
  MF_CHECK_ARG(out_dtype == MFNET_BF16 || out_dtype == MFNET_F32, "conv3x3_bf16: bad out dtype");
  const int64_t M = B * H * W;
  MF_CHECK_ARG(M < (1ll << 31) && H * W < (1ll << 31) && H < (1 << 20) && W < (1 << 20), "conv3x3_bf16: map too large");
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = (const bf16_t*)x; p.lda = ldx; p.W = (const bf16_t*)w; p.ldw = ldw;
  p.M = (int)M; p.N = (int)N; p.K = (int)(9 * C);
  p.n_groups = 1; p.out = out; p.out_dtype = out_dtype; p.ldc = ldc;
  p.tiles_m = (int)ceil_div64(M, BM); p.tiles_n = (int)ceil_div64(N, BN);
  p.k_splits = 1;
  p.conv_H = (int)H; p.conv_W = (int)W; p.conv_cb = (int)(C / 64);
  const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  MF_CHECK_ARG(nwg < (1ll << 31), "conv3x3_bf16: too many tiles");
  hipLaunchKernelGGL((gemm_bf16_kernel<0, false, true>), dim3((unsigned)nwg, 1), dim3(256), 0, stream, p);
  MF_LAUNCH_CHECK("conv3x3_bf16");
  return MFNET_OK;
}

extern "C" int mfnet_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int64_t N,
                               int64_t K, const float* bias, const float* U, int64_t ldu, const float* V,
                               int64_t v_group_stride, int64_t rows_per_group, int n_groups, int R,
                               float lr_scale, int act, const void* aux, int64_t ldaux, void* out_pre,
                               int64_t ldpre, const float* residual, int64_t ldres, int64_t res_row_mod,
                               void* out, int out_dtype, int64_t ldc, int k_splits, hipStream_t stream) {
  return gemm_impl(A, lda, W, ldw, 0, M, N, K, bias, 0, U, ldu, V, v_group_stride, rows_per_group, n_groups, R, lr_scale, act, aux, ldaux,
                   out_pre, ldpre, residual, ldres, res_row_mod, out, out_dtype, ldc, k_splits, stream);
}

// mfnet_gemm_bf16 with the tile shape named by the caller (benchmarks, tests): tile 0 = the library's choice, -1 = the 128x128
// kernel, 3 / 4 / 6 / 8 / 9 = 256 x (32*tile) tiles.
extern "C" int mfnet_gemm_bf16_tiled(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int64_t N,
                                     int64_t K, const float* bias, const float* U, int64_t ldu, const float* V,
                                     int64_t v_group_stride, int64_t rows_per_group, int n_groups, int R,
                                     float lr_scale, int act, const void* aux, int64_t ldaux, void* out_pre,
                                     int64_t ldpre, const float* residual, int64_t ldres, int64_t res_row_mod,
                                     void* out, int out_dtype, int64_t ldc, int k_splits, int tile, hipStream_t stream) {
  return gemm_impl(A, lda, W, ldw, 0, M, N, K, bias, 0, U, ldu, V, v_group_stride, rows_per_group, n_groups, R, lr_scale, act, aux, ldaux,
                   out_pre, ldpre, residual, ldres, res_row_mod, out, out_dtype, ldc, k_splits, stream, tile);
}

// Per-modality dense layers in one launch: rows [g*rows_per_group, (g+1)*rows_per_group) (cyclically over n_groups) multiply
// W + g*w_group_stride and add bias + g*bias_group_stride - the Img_ / DSM_ and MLPx_ / MLPy_ adapters of the MMAdapter
// block, whose two half-batch GEMMs of 32-64 tiles each would leave most of the chip idle (adapter_fusionblock.py:88-108).
extern "C" int mfnet_gemm_bf16_grouped(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t w_group_stride, int64_t M,
                                       int64_t N, int64_t K, const float* bias, int64_t bias_group_stride, int64_t rows_per_group,
                                       int n_groups, int act, const void* aux, int64_t ldaux, void* out_pre, int64_t ldpre,
                                       const float* residual, int64_t ldres, void* out, int out_dtype, int64_t ldc,
                                       hipStream_t stream) {
  MF_CHECK_ARG(n_groups >= 1 && rows_per_group > 0, "gemm_bf16_grouped: bad group args");
  return gemm_impl(A, lda, W, ldw, w_group_stride, M, N, K, bias, bias_group_stride, nullptr, 0, nullptr, 0, rows_per_group, n_groups, 0,
                   1.0f, act, aux, ldaux, out_pre, ldpre, residual, ldres, 0, out, out_dtype, ldc, 1, stream);
}
