// Fused attention forward for the SAM ViT blocks: windowed (14x14, padded keys synthesised from the
// qkv bias) and global, with decomposed relative-position bias, flash-style online softmax.
// See attn_common.h for the reference semantics (lora_block.py:169-196, 248-317).
//
// Workgroup = 4 waves x 16 queries of one (image-modality, head, window); the 1-D grid (att_locate_wg) enumerates only the
// (window, 64-query tile) pairs that exist. Keys are staged in LDS in chunks of 128: K and V row-major
// [key][64] (XOR-swizzled 16-B chunks; K fragments by ds_read_b128, V^T fragments by the gfx950 transposing
// read ds_read_b64_tr_b16 - no transposed image is ever stored).
// Scores are computed "swapped" (S^T = K.Q^T: key on the accumulator rows, query on the lane), so
//   * softmax statistics are per-lane scalars (4 lane groups combined with two shuffles),
//   * the bf16 P^T accumulator registers ARE the B operand of O^T = V^T.P^T (keys permuted identically
//     in the V^T fragment read), no LDS round trip.
// The rel-pos bias T[q][KH-1-kh+qh] + T[q][Lh+KW-1-kw+qw] is NOT gathered per score (that made the kernel
// VALU-bound: ~10 of ~20 instructions per score). With the query's table rows shifted into "slots"
// T'[q][j] (j < KH: T[q][j+qh]; then T[q][Lh+(j-KH)+qw]) it is the product G.T'^T of the 0/1 indicator
// G[key][j] = (j == KH-1-kh) | (j == KH+KW-1-kw) with T' - one more MFMA k-step per 32 slots, fed by an
// indicator image staged beside K. T' is split hi + lo in bf16 (two k-steps), so the bias keeps fp32-level
// accuracy, and it is pre-divided by `scale` so one multiply serves the whole logit.
#include "attn_common.h"
#include <mutex>
#include <stdlib.h>

struct AttnFwdParams {
  AttnGeom g;
  const bf16_t* qkv;      // [G*T, 3D]
  const float* qkv_bias;  // [3D] fp32 (padded keys / values)
  const bf16_t* tbl;      // [NT, 32*ceil(hd/32)] bf16, rows >= Lh+Lw and columns >= hd are zero
  bf16_t* out;            // [G*T, D]
  float* lse;             // [G, heads, T]
};

// ROWT ("row tiles"): global attention over a grid whose rows are 64 keys wide (1024-px inputs): every 64-key tile is one
// key row, so the bias of a tile is separable in registers - a per-query scalar (the h term of that row) plus a per-lane
// 4 x 4 patch of the w term that is the same for every tile - and the indicator product (8 of the 12 MFMA k-steps per
// tile at KH + KW = 128) disappears.
template <int NSG, int HD, bool ROWT>      // NSG: 32-slot groups of the indicator product, ceil((KH + KW) / 32) rounded up to 1, 2 or 4
__global__ __launch_bounds__(256, (NSG == 1 && HD <= 64) ? 4 : ((ROWT && HD <= 64) ? 3 : 2)) void attn_fwd_kernel(const AttnFwdParams p) {
  typedef AttHd<HD> H;
  constexpr int KS = H::KS, UT = H::UT, ROWB = H::ROWB, CHS = H::CHS, CHV = H::CHV, RPI = H::RPI, NIT = ATT_KC / H::RPI;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const AttnGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int T = g.Sh * g.Sw;
  int gi, head, tile_id;
  if (!att_locate_wg(g.ntile, g.heads, g.G, &tile_id, &head, &gi)) return;
  int wi, wj, qt0;
  att_locate_tile(g, tile_id, &wi, &wj, &qt0);
  const int nq = num_queries(g, wi, wj);
  const int Nk = g.KH * g.KW;
  const int TLD = g.NT + 4;
  const int nsg = (g.KH + g.KW + 31) >> 5;
  const int regionA = att_region_a(g.NT, ROWB, ROWT);

  unsigned char* K_lds = smem;                                     // [128][64] bf16, swizzled
  unsigned char* V_lds = smem + ATT_KC * ROWB;                     // [128][hd] bf16, swizzled
  float* T_lds = (float*)smem;                                     // [64][NT+4]: only until the key loop starts
  unsigned char* G_lds = smem + regionA;                           // [nsg][128][32] bf16 indicator rows

  const int64_t row_base = (int64_t)gi * T;
  const int ld = 3 * g.D;                                          // (one image's qkv rows span < 2^31 elements)
  const bf16_t* qkv_img = p.qkv + row_base * ld;

  // ---- this lane's query (column of every accumulator tile)
  const QueryPos qp = query_pos(g, wi, wj, qt0 + wave * 16 + fr);
  const bool wave_active = qt0 + wave * 16 < nq;                   // wave-uniform: waves without a query only stage
  bf16x8_t qf[KS];
  att_load_row_frags<HD>(qkv_img + qp.tok * ld + head * HD, fq, qp.valid, qf);
  // this thread always stages 16-B chunk `ch` of a key row: the padded-key value of that chunk (the qkv bias,
  // rounded to bf16 like every other qkv entry) is kept in registers
  const int ch = tid % CHS, chl = ch < CHV ? ch : CHV - 1;         // slots >= CHV hold zeros (never loaded)
  const PaddedKV pad = att_padded_kv(p.qkv_bias, g.D, head, HD, chl);

  // All global loads of a key chunk are issued first, unconditionally: padded / out-of-range keys read row 0
  // and are replaced when the registers are written to LDS. (Prefetching a chunk ahead costs 40 VGPRs, i.e.
  // one resident workgroup per CU, and measured 20 % slower.)
  u32x4_t kv[NIT], vv[NIT];
  int kkind[NIT];
  auto issue_chunk = [&](int kc) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int kg = kc + it * RPI + tid / CHS;
      int kh = 0, kw = 0;
      const int tok = kg < Nk ? key_token(g, wi, wj, kg, &kh, &kw) : -2;
      const bf16_t* src = qkv_img + (max(tok, 0) * ld + g.D + head * HD + chl * 8);
      kv[it] = *reinterpret_cast<const u32x4_t*>(src);
      vv[it] = *reinterpret_cast<const u32x4_t*>(src + g.D);
      kkind[it] = ch >= CHV ? 2 : (tok < 0 ? (tok == -2 ? 2 : 1) : 0);   // 0 real, 1 padded, 2 beyond the window / zero slot
    }
  };
  // ---- T^T = tbl . Q^T  -> T_lds[query][table row], then this lane's slots of T' / scale as bf16 hi + lo
  bf16x8_t tqh[NSG], tql[NSG];
  float wb[4][4];                                                  // ROWT: bias / scale of this lane's keys 16 t + 4 fq + r (w term)
  float* Thq = (float*)G_lds;                                      // ROWT: [64 queries][KH] h term / scale
  if (ROWT) {
    // the table is walked 64 rows at a time through the wave's own block buffer; each block hands over the entries it holds
    const float inv = 1.0f / g.scale;
    float* Tb = T_lds + (wave * 16 + fr) * ATT_TB_LD;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) wb[t][r] = 0.f;
    for (int b0 = 0; b0 < g.NT; b0 += 64) {
      att_table_block<HD>(p.tbl, g.NT, b0, qf, Tb, fr, fq);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int idx = g.Lh + (g.KW - 1 - (16 * t + 4 * fq + r)) + qp.qw - b0;
          if (idx >= 0 && idx < 64) wb[t][r] = Tb[idx] * inv;
        }
      for (int j = fq; j < g.KH; j += 4) {
        const int idx = (g.KH - 1 - j) + qp.qh - b0;
        if (idx >= 0 && idx < 64) Thq[(wave * 16 + fr) * g.KH + j] = Tb[idx] * inv;
      }
    }
  } else {
    float* Trow = T_lds + (wave * 16 + fr) * TLD;
    att_table_product<HD>(p.tbl, g.NT, qf, Trow, fr, fq);
    att_slot_fragments<NSG>(g, Trow, qp.qh, qp.qw, fq, nsg, tqh, tql);
  }

  float m_run = -INFINITY, l_run = 0.f;
  const float cs = g.scale * 1.4426950408889634f;                  // logits are scale * acc; exp via exp2
  f32x4_t o[UT];
#pragma unroll
  for (int u = 0; u < UT; ++u) o[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // (ROWT, 32 chunks per workgroup at 3 workgroups per CU: issuing the next chunk's loads under this chunk's MFMAs measured
  // no gain - 296 against 298 us - once the table block buffer let three workgroups share a CU)
  for (int kc = 0; kc < Nk; kc += ATT_KC) {
    issue_chunk(kc);
    __syncthreads();   // previous chunk fully consumed (first pass: every wave has read its T rows)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int kl = it * RPI + tid / CHS;
      // three exclusive stores under the execution mask instead of 16 selects per row: a window without padded keys (most of the
      // work) takes the first branch as a whole
      u32x4_t* kd = reinterpret_cast<u32x4_t*>(K_lds + kl * ROWB + ((ch ^ kswz(kl)) << 4));
      u32x4_t* vd = reinterpret_cast<u32x4_t*>(V_lds + kl * ROWB + ((ch ^ kswz(kl)) << 4));
      if (kkind[it] == 0) { *kd = kv[it]; *vd = vv[it]; }
      else if (kkind[it] == 1) { *kd = pad.k; *vd = pad.v; }
      else { *kd = (u32x4_t){0u, 0u, 0u, 0u}; *vd = (u32x4_t){0u, 0u, 0u, 0u}; }
    }
    if (!ROWT) att_stage_indicator(g, G_lds, kc, Nk, nsg, tid);
    __syncthreads();
    const int kvalid = min(ATT_KC, Nk - kc);
    if (wave_active) {
      for (int kt = 0; kt * 64 < kvalid; ++kt) {
        // ---- S^T tiles (16 keys x 16 queries each): K.Q^T + G.T'^T (ROWT: the bias is the accumulator's initial value)
        f32x4_t s[4];
        const float hb = ROWT ? Thq[(wave * 16 + fr) * g.KH + ((kc + kt * 64) >> 6)] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s[t] = ROWT ? (f32x4_t){wb[t][0] + hb, wb[t][1] + hb, wb[t][2] + hb, wb[t][3] + hb} : (f32x4_t){0.f, 0.f, 0.f, 0.f};
          const int kr = kt * 64 + 16 * t + fr;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(K_lds + kr * ROWB + (((fq + 4 * ks) ^ kswz(kr)) << 4));
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[t], 0, 0, 0);
          }
#pragma unroll
          for (int c = 0; c < NSG; ++c) {
            if (!ROWT && c < nsg) {
              const bf16x8_t gf = *reinterpret_cast<const bf16x8_t*>(G_lds + c * (ATT_KC * ATT_GROW) + kr * ATT_GROW + ((fq ^ gswz(kr)) << 4));
              s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, tqh[c], s[t], 0, 0, 0);
              s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, tql[c], s[t], 0, 0, 0);
            }
          }
        }
        if (kt * 64 + 64 > kvalid) {       // only the window's last tile holds keys beyond Nk
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (kt * 64 + 16 * t + 4 * fq + r >= kvalid) s[t][r] = -INFINITY;
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t) tmax = fmaxf(tmax, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cs);    // m_run = -inf on the first tile -> 0
        const float mc = m_new * cs;
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(s[t][r], cs, -mc));
            s[t][r] = e;
            psum += e;
          }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int u = 0; u < UT; ++u) { o[u][0] *= alpha; o[u][1] *= alpha; o[u][2] *= alpha; o[u][3] *= alpha; }
        // ---- O^T += V^T . P^T ; element j of k-step s2 <-> key 32*s2 + 16*(j>>2) + 4*fq + (j&3). The V^T
        // fragment comes from two transposing reads of 4 keys x 16 head-dims each (lane 4q+p of a 16-lane
        // group addresses key q, head-dims 4p..4p+3 of the block; lane i receives head-dim i of the 4 keys).
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          union { bf16x8_t v; unsigned int w[4]; } pf;
          pf.w[0] = pack_bf16x2(s[2 * s2][0], s[2 * s2][1]);
          pf.w[1] = pack_bf16x2(s[2 * s2][2], s[2 * s2][3]);
          pf.w[2] = pack_bf16x2(s[2 * s2 + 1][0], s[2 * s2 + 1][1]);
          pf.w[3] = pack_bf16x2(s[2 * s2 + 1][2], s[2 * s2 + 1][3]);
          const int vrow = kt * 64 + 32 * s2 + 4 * fq + (fr >> 2);
#pragma unroll
          for (int u = 0; u < UT; ++u) {
            const unsigned char* va = V_lds + vrow * ROWB + (((2 * u + ((fr >> 1) & 1)) ^ kswz(vrow)) << 4) + ((fr & 1) << 3);
            union { bf16x8_t v; s16x4_t h[2]; } vf;
            vf.h[0] = att_tr_read(va);
            vf.h[1] = att_tr_read(va + 16 * ROWB);
            o[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf.v, o[u], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- finish: combine the 4 lane groups' partial sums, normalise, store
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  if (qp.valid) {
    const float inv = 1.f / l_run;
    bf16_t* op = p.out + (row_base + qp.tok) * g.D + head * HD + 4 * fq;
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      u32x2_t w;
      w.x = pack_bf16x2(o[u][0] * inv, o[u][1] * inv);
      w.y = pack_bf16x2(o[u][2] * inv, o[u][3] * inv);
      *reinterpret_cast<u32x2_t*>(op + 16 * u) = w;
    }
    if (fq == 0 && p.lse) p.lse[((int64_t)gi * g.heads + head) * T + qp.tok] = m_run * g.scale + __logf(l_run);
  }
}

// =============================================================================================== pair-resident forward
// Small token grids (T <= 256 tokens, key rows of at most 16 keys: the 256-px tiles of the training step, 14 x 14 windows and the
// 16 x 16 global blocks). The tiled kernel above gives every 64-query tile a workgroup of its own that stages the window's K / V
// again (7 tiles per image-head in a windowed block, three of them with 4 - 28 queries) and waits at two barriers per 128-key
// chunk: its waves issue instructions in 28 % of their cycles (profiles/r03_b_attn_pmc.json). Here ONE workgroup owns an
// (image-modality, head) pair: K and V of all its tokens are staged once, by token, plus one row for the padded key (the qkv bias),
// and after that single barrier each wave walks 16-query tiles on its own - a window's keys are found through the token map, so
// windows share the one image.
//   Key order: a window's keys are laid out 16 per key ROW (kw < KW real, the rest masked), so a 16-key accumulator tile is one key
// row kh: the decomposed rel-pos bias of lane (query, fq) is hb[kh] + wb[kw = 4 fq + r] - per-query table entries that start the
// accumulator in fp32. No indicator image and none of its MFMAs (half of the score MFMAs of the tiled kernel), no per-chunk
// staging, no barrier in the key loop.
#define ATP_WAVES 6
#define ATP_THREADS (64 * ATP_WAVES)
#define ATP_ROWS 257                    // 256 tokens + the padded key
#define ATP_TB_LD 36
__host__ __device__ inline int atp_tiles16(const AttnGeom& g) {
  int n = 0;
  for (int wi = 0; wi < g.nWh; ++wi)
    for (int wj = 0; wj < g.nWw; ++wj) {
      const int rh = g.ws > 0 ? (g.ws < g.Sh - wi * g.ws ? g.ws : g.Sh - wi * g.ws) : g.Sh;
      const int rw = g.ws > 0 ? (g.ws < g.Sw - wj * g.ws ? g.ws : g.Sw - wj * g.ws) : g.Sw;
      n += (rh * rw + 15) / 16;
    }
  return n;
}
__device__ __forceinline__ void atp_locate_tile(const AttnGeom& g, int t, int* wi, int* wj, int* q0) {
  int i = 0, j = 0;
  for (int w = 0; w + 1 < g.nWh * g.nWw; ++w) {
    const int nt = (num_queries(g, i, j) + 15) >> 4;
    if (t < nt) break;
    t -= nt;
    if (++j == g.nWw) { j = 0; ++i; }
  }
  *wi = i; *wj = j; *q0 = t * 16;
}

template <int HD>
__global__ __launch_bounds__(ATP_THREADS, 3) void attn_fwd_pair_kernel(const AttnFwdParams p) {
  typedef AttHd<HD> H;
  static_assert(HD == 64, "pair-resident attention: head dim 64");
  constexpr int KS = H::KS, UT = H::UT, ROWB = H::ROWB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const AttnGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int T = g.Sh * g.Sw;
  const int gi = blockIdx.x / g.heads, head = blockIdx.x - gi * g.heads;
  unsigned char* K_lds = smem;
  unsigned char* V_lds = smem + ATP_ROWS * ROWB;
  float* Tb = (float*)(smem + 2 * ATP_ROWS * ROWB) + (wave * 16 + fr) * ATP_TB_LD;      // this lane's query row of the wave's table block
  const int64_t row_base = (int64_t)gi * T;
  const int ld = 3 * g.D;
  const bf16_t* qkv_img = p.qkv + row_base * ld;

  // ---- stage K / V of every token once (16-B chunks, XOR-swizzled by row), and the padded key's row
  {
    constexpr int NLD = (256 * 8 + ATP_THREADS - 1) / ATP_THREADS;
    u32x4_t kr[NLD], vr[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * ATP_THREADS, row = min(idx >> 3, T - 1), ch = idx & 7;
      const bf16_t* src = qkv_img + (row * ld + g.D + head * HD + ch * 8);
      kr[i] = *reinterpret_cast<const u32x4_t*>(src);
      vr[i] = *reinterpret_cast<const u32x4_t*>(src + g.D);
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * ATP_THREADS, row = idx >> 3, ch = idx & 7;
      if (row < T) {
        *reinterpret_cast<u32x4_t*>(K_lds + row * ROWB + ((ch ^ kswz(row)) << 4)) = kr[i];
        *reinterpret_cast<u32x4_t*>(V_lds + row * ROWB + ((ch ^ kswz(row)) << 4)) = vr[i];
      }
    }
    if (tid < 8) {
      const PaddedKV pad = att_padded_kv(p.qkv_bias, g.D, head, HD, tid);
      *reinterpret_cast<u32x4_t*>(K_lds + 256 * ROWB + ((tid ^ kswz(256)) << 4)) = pad.k;
      *reinterpret_cast<u32x4_t*>(V_lds + 256 * ROWB + ((tid ^ kswz(256)) << 4)) = pad.v;
    }
  }
  __syncthreads();

  const float cs = g.scale * 1.4426950408889634f, inv = 1.0f / g.scale;
  const int ntile = atp_tiles16(g);
  const int nkt = (g.KH + 3) >> 2;
  for (int tile = wave; tile < ntile; tile += ATP_WAVES) {
    int wi, wj, q0;
    atp_locate_tile(g, tile, &wi, &wj, &q0);
    const QueryPos qp = query_pos(g, wi, wj, q0 + fr);
    bf16x8_t qf[KS];
    att_load_row_frags<HD>(qkv_img + qp.tok * ld + head * HD, fq, qp.valid, qf);
    const int oh = g.ws > 0 ? wi * g.ws : 0, ow = g.ws > 0 ? wj * g.ws : 0;
    // LDS row of key (kh, kw) of this window: its token, the padded key's row, or (beyond the key grid: masked) any row
    auto key_row = [&](int kh, int kw) -> int {
      const int y = oh + kh, x = ow + kw;
      return (kh < g.KH && kw < g.KW && y < g.Sh && x < g.Sw) ? y * g.Sw + x : 256;
    };
    // ---- T^T = tbl . Q^T, 32 table rows at a time through the wave's block buffer; each block hands over the entries this query
    // uses: hb[kh] = T[q][(KH-1-kh) + qh] / scale, wb[r] = T[q][Lh + (KW-1-kw) + qw] / scale for kw = 4 fq + r (-inf: no such key)
    float hb[16], wb[4];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      // pass 0: table rows [0, 32) hold every h row a query uses ((KH-1-kh) + qh <= 30); pass 1: rows [Lh, Lh + 32) every w row
      const int b0 = pass ? g.Lh : 0;
      bf16x8_t a[2][KS];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16_t* tp = p.tbl + (int64_t)min(b0 + 16 * i + fr, g.NT - 1) * H::HDP + 8 * fq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[i][ks] = *reinterpret_cast<const bf16x8_t*>(tp + 32 * ks);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], qf[ks], acc, 0, 0, 0);
        *reinterpret_cast<float4*>(Tb + 16 * i + 4 * fq) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      }
      if (pass == 0) {
#pragma unroll
        for (int kh = 0; kh < 16; ++kh) {
          const float v = Tb[max(g.KH - 1 - kh, 0) + qp.qh] * inv;
          hb[kh] = kh < g.KH ? v : -INFINITY;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kw = 4 * fq + r;
          const float v = Tb[max(g.KW - 1 - kw, 0) + qp.qw] * inv;
          wb[r] = kw < g.KW ? v : -INFINITY;
        }
      }
    }

    float m_run = -INFINITY, l_run = 0.f;
    f32x4_t o[UT];
#pragma unroll
    for (int u = 0; u < UT; ++u) o[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      if (kt < nkt) {
        // ---- S^T tiles: key row kh = 4 kt + t on the accumulator rows (kw = 4 fq + r), the query on the lane
        f32x4_t s[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float hv = hb[4 * kt + t];
          s[t] = (f32x4_t){wb[0] + hv, wb[1] + hv, wb[2] + hv, wb[3] + hv};
          const int kr = key_row(4 * kt + t, fr);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(K_lds + kr * ROWB + (((fq + 4 * ks) ^ kswz(kr)) << 4));
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[t], 0, 0, 0);
          }
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t) tmax = fmaxf(tmax, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cs);    // m_run = -inf on the first tile -> 0
        const float mc = m_new * cs;
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(s[t][r], cs, -mc));
            s[t][r] = e;
            psum += e;
          }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int u = 0; u < UT; ++u) { o[u][0] *= alpha; o[u][1] *= alpha; o[u][2] *= alpha; o[u][3] *= alpha; }
        // ---- O^T += V^T . P^T (k-slot (fq, j) of k-step s2 <-> key row 4 kt + 2 s2 + (j >> 2), key column 4 fq + (j & 3)); V^T by
        // transposing reads of 4 keys x 16 head dims (lane 4q+p of a 16-lane group addresses key column 4 fq + q)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          union { bf16x8_t v; unsigned int w[4]; } pf;
          pf.w[0] = pack_bf16x2(s[2 * s2][0], s[2 * s2][1]);
          pf.w[1] = pack_bf16x2(s[2 * s2][2], s[2 * s2][3]);
          pf.w[2] = pack_bf16x2(s[2 * s2 + 1][0], s[2 * s2 + 1][1]);
          pf.w[3] = pack_bf16x2(s[2 * s2 + 1][2], s[2 * s2 + 1][3]);
          const int v0 = key_row(4 * kt + 2 * s2, 4 * fq + (fr >> 2)), v1 = key_row(4 * kt + 2 * s2 + 1, 4 * fq + (fr >> 2));
#pragma unroll
          for (int u = 0; u < UT; ++u) {
            const int chn = 2 * u + ((fr >> 1) & 1), sub = (fr & 1) << 3;
            union { bf16x8_t v; s16x4_t h[2]; } vf;
            vf.h[0] = att_tr_read(V_lds + v0 * ROWB + ((chn ^ kswz(v0)) << 4) + sub);
            vf.h[1] = att_tr_read(V_lds + v1 * ROWB + ((chn ^ kswz(v1)) << 4) + sub);
            o[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf.v, o[u], 0, 0, 0);
          }
        }
      }
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (qp.valid) {
      const float il = 1.f / l_run;
      bf16_t* op = p.out + (row_base + qp.tok) * g.D + head * HD + 4 * fq;
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        u32x2_t w;
        w.x = pack_bf16x2(o[u][0] * il, o[u][1] * il);
        w.y = pack_bf16x2(o[u][2] * il, o[u][3] * il);
        *reinterpret_cast<u32x2_t*>(op + 16 * u) = w;
      }
      if (fq == 0 && p.lse) p.lse[((int64_t)gi * g.heads + head) * T + qp.tok] = m_run * g.scale + __logf(l_run);
    }
  }
}
// the pair-resident kernels apply: head dim 64, at most 256 tokens, key rows of at most 16 keys (MFNET_ATTN_PAIR=0: off, for A / B runs)
bool attn_pair_resident(const AttnGeom& g, int hd) {
  static const bool on = !(getenv("MFNET_ATTN_PAIR") && getenv("MFNET_ATTN_PAIR")[0] == '0');
  // windowed blocks only: global attention over 16 x 16 tokens measured 32 us against 30 for the tiled kernel (4 full tiles per
  // pair, nothing to merge), the 14 x 14 windows of the same grid 40 against 45
  return on && hd == 64 && g.ws > 0 && g.Sh * g.Sw <= 256 && g.KH <= 16 && g.KW <= 16;
}

int attn_fill_geom(AttnGeom* g, int64_t G, int64_t Sh, int64_t Sw, int64_t heads, int64_t hd, int64_t ws,
                   int64_t Lh, int64_t Lw, int64_t NT, float scale, const char* who) {
  MF_CHECK_ARG(hd == 64 || hd == 80, "%s: head dim %lld unsupported (64: ViT-B / ViT-L, 80: ViT-H)", who, (long long)hd);
  MF_CHECK_ARG(G > 0 && Sh > 0 && Sw > 0 && heads > 0 && ws >= 0, "%s: bad geometry", who);
  g->G = (int)G; g->Sh = (int)Sh; g->Sw = (int)Sw; g->heads = (int)heads; g->D = (int)(heads * hd);
  g->ws = (int)ws;
  if (ws > 0) {
    g->nWh = (int)ceil_div64(Sh, ws); g->nWw = (int)ceil_div64(Sw, ws); g->KH = g->KW = (int)ws;
  } else {
    g->nWh = g->nWw = 1; g->KH = (int)Sh; g->KW = (int)Sw;
  }
  MF_CHECK_ARG(Lh == 2 * g->KH - 1 && Lw == 2 * g->KW - 1, "%s: rel-pos tables must have 2*k-1 rows (got %lld,%lld for key grid %dx%d)", who, (long long)Lh, (long long)Lw, g->KH, g->KW);
  MF_CHECK_ARG(NT % 32 == 0 && NT >= Lh + Lw && NT <= 256, "%s: NT=%lld must be a multiple of 32 in [Lh+Lw, 256]", who, (long long)NT);
  g->Lh = (int)Lh; g->NT = (int)NT; g->scale = scale;
  return MFNET_OK;
}

bool attn_row_tiles(const AttnGeom& g);
static size_t attn_fwd_lds(const AttnGeom& g, int rowb) {
  if (attn_row_tiles(g)) return (size_t)att_region_a(g.NT, rowb, true) + (size_t)ATT_QT * g.KH * 4;      // K / V + Thq
  return (size_t)att_region_a(g.NT, rowb) + (size_t)((g.KH + g.KW + 31) / 32) * ATT_KC * ATT_GROW;
}

// global attention whose key rows are exactly one 64-key tile wide: the separable-bias (ROWT) kernels apply
bool attn_row_tiles(const AttnGeom& g) {
  return g.ws == 0 && g.KW == 64 && g.KH <= 64;
}

// number of (window, 64-query tile) pairs of one image = grid.x of the query-tiled kernels
int attn_query_tiles(const AttnGeom& g) {
  int n = 0;
  for (int wi = 0; wi < g.nWh; ++wi)
    for (int wj = 0; wj < g.nWw; ++wj) {
      const int rh = g.ws > 0 ? (g.ws < g.Sh - wi * g.ws ? g.ws : g.Sh - wi * g.ws) : g.Sh;
      const int rw = g.ws > 0 ? (g.ws < g.Sw - wj * g.ws ? g.ws : g.Sw - wj * g.ws) : g.Sw;
      n += (rh * rw + ATT_QT - 1) / ATT_QT;
    }
  return n;
}

extern "C" int mfnet_attn_fwd(const void* qkv, const float* qkv_bias, const void* tbl, void* out, float* lse,
                              int64_t G, int64_t Sh, int64_t Sw, int64_t heads, int64_t hd, int64_t ws,
                              int64_t Lh, int64_t Lw, int64_t NT, float scale, hipStream_t stream) {
  MF_CHECK_ARG(qkv && qkv_bias && tbl && out, "attn_fwd: null pointer");
  AttnFwdParams p;
  int rc = attn_fill_geom(&p.g, G, Sh, Sw, heads, hd, ws, Lh, Lw, NT, scale, "attn_fwd");
  if (rc) return rc;
  p.qkv = (const bf16_t*)qkv; p.qkv_bias = qkv_bias; p.tbl = (const bf16_t*)tbl; p.out = (bf16_t*)out; p.lse = lse;
  const int rowb = hd <= 64 ? AttHd<64>::ROWB : AttHd<80>::ROWB;
  const size_t lds = attn_fwd_lds(p.g, rowb);
  // kernels that ask for more than 64 KB of dynamic LDS need the attribute once per device; the only state kept is the
  // per-device "done" flag (guarded: entry points may be called from any thread)
  {
    static std::mutex mu;
    static bool done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { mfnet_set_error("attn_fwd: hipGetDevice failed"); return MFNET_ERR_HIP; }
    std::lock_guard<std::mutex> lk(mu);
    if (!done[dev]) {
      {
        hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_pair_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { mfnet_set_error("attn_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MFNET_ERR_HIP; }
      }
      const void* fns[8] = {(const void*)attn_fwd_kernel<1, 64, false>, (const void*)attn_fwd_kernel<2, 64, false>, (const void*)attn_fwd_kernel<4, 64, false>,
                            (const void*)attn_fwd_kernel<1, 80, false>, (const void*)attn_fwd_kernel<2, 80, false>, (const void*)attn_fwd_kernel<4, 80, false>,
                            (const void*)attn_fwd_kernel<4, 64, true>, (const void*)attn_fwd_kernel<4, 80, true>};
      for (int i = 0; i < 8; ++i) {
        hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { mfnet_set_error("attn_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MFNET_ERR_HIP; }
      }
      done[dev] = true;
    }
  }
  MF_CHECK_ARG(p.g.KH + p.g.KW <= 128, "attn_fwd: key grid %dx%d too large (KH + KW <= 128)", p.g.KH, p.g.KW);
  if (attn_pair_resident(p.g, (int)hd)) {
    const size_t plds = 2 * ATP_ROWS * AttHd<64>::ROWB + ATP_WAVES * 16 * ATP_TB_LD * 4;
    hipLaunchKernelGGL((attn_fwd_pair_kernel<64>), dim3((unsigned)(heads * G)), dim3(ATP_THREADS), plds, stream, p);
    MF_LAUNCH_CHECK("attn_fwd(pair)");
    return MFNET_OK;
  }
  p.g.ntile = attn_query_tiles(p.g);
  const int64_t nwg = att_grid(p.g.ntile, heads * G);
  MF_CHECK_ARG(nwg < (1ll << 31), "attn_fwd: grid too large");
  dim3 grid((unsigned)nwg);
  MF_CHECK_ARG(lds <= 160 * 1024, "attn_fwd: LDS demand too large");
  const int nsg = (p.g.KH + p.g.KW + 31) / 32;
#define ATT_FWD(HDV)                                                                                   \
  do {                                                                                                 \
    if (attn_row_tiles(p.g)) hipLaunchKernelGGL((attn_fwd_kernel<4, HDV, true>), grid, dim3(256), lds, stream, p);   \
    else if (nsg <= 1) hipLaunchKernelGGL((attn_fwd_kernel<1, HDV, false>), grid, dim3(256), lds, stream, p);        \
    else if (nsg == 2) hipLaunchKernelGGL((attn_fwd_kernel<2, HDV, false>), grid, dim3(256), lds, stream, p);        \
    else hipLaunchKernelGGL((attn_fwd_kernel<4, HDV, false>), grid, dim3(256), lds, stream, p);                      \
  } while (0)
  if (hd == 64) ATT_FWD(64); else ATT_FWD(80);
#undef ATT_FWD
  MF_LAUNCH_CHECK("attn_fwd");
  return MFNET_OK;
}
