// Geometry shared by the attention forward / backward kernels.
//
// Reference semantics (medsam/models/ImageEncoder/vit/lora_block.py):
//  * window blocks (:70-79,199-245): the (Sh,Sw) token grid is zero-padded AFTER norm1 to a multiple of
//    ws and cut into ws x ws windows. A padded token's qkv row is exactly the qkv bias (its input is 0,
//    so the LoRA update vanishes too); padded KEYS still enter the softmax with their own rel-pos
//    bias, padded QUERIES are discarded by window_unpartition. We therefore run only the real queries
//    and synthesise padded keys from the bias vector - no 3.06x token inflation.
//  * scores (:181-193,281-317): softmax(scale * q.k + q.Rh[qh,kh] + q.Rw[qw,kw]) with the rel-pos term
//    taken from the UNSCALED q. With tbl = [rel_h ; rel_w] (already resized by get_rel_pos's linear
//    interpolation, :260-271) the bias of (query, key) is T[q][qh-kh+KH-1] + T[q][Lh+qw-kw+KW-1],
//    T = q @ tbl^T : one extra skinny MFMA product per query tile instead of two einsums.
#pragma once
#include "common.h"

// Head dim: 64 (ViT-B / ViT-L) or 80 (ViT-H: 1280 / 16). The kernels are templated on it through AttHd: 80 runs
// three 32-wide MFMA k-steps over a zero-padded 96 and five 16-wide output tiles; LDS rows become 256 B (16 chunk
// slots, 10 with data, 2 zero, 4 unused) so that the XOR swizzle - which only touches the low three chunk bits - still
// stays inside the row.
template <int HD> struct AttHd {
  static_assert(HD % 16 == 0 && HD >= 32 && HD <= 128, "head dim must be a multiple of 16 in [32, 128]");
  static constexpr int KS = (HD + 31) / 32;          // MFMA k-steps over the head dim
  static constexpr int UT = HD / 16;                 // 16-wide head-dim tiles of O / dQ / dK / dV
  static constexpr int ROWB = HD <= 64 ? 128 : 256;  // bytes per LDS row of a [row][hd] bf16 image
  static constexpr int CHS = ROWB / 16;              // 16-B chunk slots per row
  static constexpr int CHV = HD / 8;                 // chunks that hold data; slots CHV .. 4*KS-1 are zero
  static constexpr int RPI = 256 / CHS;              // rows one staging pass of the 256 threads covers
  static constexpr int HDP = 32 * KS;                // row length of the (zero padded) rel-pos table
};
#define ATT_KC 128         // keys resident in LDS per chunk (small enough for 2-3 workgroups per CU)
#define ATT_QT 64          // queries per workgroup (4 waves x 16)
#define ATT_VT_LD (ATT_KC + 8)   // bf16 elements per row of the transposed [hd][key] images

struct AttnGeom {
  int ntile;                 // (window, 64-row tile) pairs per image = attn_query_tiles
  int G, Sh, Sw, heads, D;   // G = image-modalities, token grid Sh x Sw, D = heads*64
  int ws;                    // 0 = global attention
  int nWh, nWw;              // windows per axis (1,1 for global)
  int KH, KW;                // key grid per window (ws,ws) or (Sh,Sw)
  int Lh;                    // rows of the h table inside tbl; w table starts at row Lh
  int NT;                    // padded table rows (multiple of 32, <= 256)
  float scale;
};

struct QueryPos { int valid, tok, qh, qw; };

// k / d for 0 <= k < 2^20, 1 <= d <= 128 with one reciprocal and one multiply instead of the ~25-instruction integer
// division: (k + 0.5) / d is at least 0.5 / d away from an integer while the float product is off by < k * 2.4e-7.
__device__ __forceinline__ int att_div(int k, int d) {
  return (int)(((float)k + 0.5f) * __builtin_amdgcn_rcpf((float)d));
}

// query index qi inside (window wi,wj) -> token + in-window coordinates
__device__ __forceinline__ QueryPos query_pos(const AttnGeom& g, int wi, int wj, int qi) {
  QueryPos r;
  int rh = g.Sh, rw = g.Sw, oh = 0, ow = 0;
  if (g.ws > 0) {
    oh = wi * g.ws; ow = wj * g.ws;
    rh = min(g.ws, g.Sh - oh); rw = min(g.ws, g.Sw - ow);
  }
  r.valid = qi < rh * rw;
  const int q = r.valid ? qi : 0;
  r.qh = att_div(q, rw); r.qw = q - r.qh * rw;
  r.tok = (oh + r.qh) * g.Sw + (ow + r.qw);
  return r;
}
__device__ __forceinline__ int num_queries(const AttnGeom& g, int wi, int wj) {
  if (g.ws == 0) return g.Sh * g.Sw;
  return min(g.ws, g.Sh - wi * g.ws) * min(g.ws, g.Sw - wj * g.ws);
}
// key index k (row-major over KH x KW) -> token or -1 for a padded position
__device__ __forceinline__ int key_token(const AttnGeom& g, int wi, int wj, int k, int* kh, int* kw) {
  *kh = att_div(k, g.KW); *kw = k - *kh * g.KW;
  const int y = (g.ws > 0 ? wi * g.ws : 0) + *kh, x = (g.ws > 0 ? wj * g.ws : 0) + *kw;
  return (y < g.Sh && x < g.Sw) ? y * g.Sw + x : -1;
}

__device__ __forceinline__ int kswz(int row) { return row & 7; }   // 16-B chunk XOR for [row][64 hd] images
__device__ __forceinline__ int gswz(int row) { return (row >> 2) & 3; }   // same for the 64-B indicator rows

#define ATT_GROW 64        // bytes per key row of one 32-slot indicator image

// LDS bytes shared by the K / V images (key loop) and the T table (before it). ROWT kernels walk the table in blocks of
// 64 rows through a wave-private [16][ATT_TB_LD] buffer (att_table_block) instead of holding all NT rows of 64 queries:
// at NT = 256 the whole table is 66 KB and left ONE workgroup per CU.
#define ATT_TB_LD 68
__host__ __device__ inline int att_region_a(int NT, int rowb, bool rowt = false) {
  const int kv = 2 * ATT_KC * rowb, t = rowt ? ATT_QT * ATT_TB_LD * 4 : ATT_QT * (NT + 4) * 4;
  return kv > t ? kv : t;
}

// gfx950 transposing LDS read (ds_read_b64_tr_b16): see the forward kernel's P.V step for the lane roles
__device__ __forceinline__ s16x4_t att_tr_read(const unsigned char* lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(lds_addr));
}

// XCD-aware placement of the workgroups of a 1-D grid: the hardware deals consecutive workgroup ids round-robin to the 8
// XCDs, each with its own L2. All `nt` tiles of one (image-modality, head) pair read the same K / V (or Q / dO / bias) rows, so
// pair p goes to XCD p % 8 whole: id = 8 * s + x runs tile s % nt of pair x + 8 * (s / nt). (With (tile, head, image) as the
// grid axes a pair's tiles were spread over all XCDs: every L2 fetched every pair's rows - at 64 x 64 tokens the dK/dV kernel
// pulled 8.5 TB/s through the fabric.) Grid: att_grid(nt, pairs) workgroups; ids past the last pair of an XCD return at once.
__device__ __forceinline__ bool att_locate_wg(int nt, int heads, int G, int* tile, int* head, int* gi) {
  const int id = blockIdx.x, x = id & 7, s = id >> 3;
  const int pr = s / nt, p = x + 8 * pr;
  if (p >= heads * G) return false;
  *tile = s - pr * nt;
  *gi = p / heads; *head = p - *gi * heads;
  return true;
}
static inline int64_t att_grid(int64_t nt, int64_t pairs) { return 8 * nt * ((pairs + 7) / 8); }

// tile index -> (window, first query) over the (window, 64-query tile) pairs that exist
__device__ __forceinline__ void att_locate_tile(const AttnGeom& g, int t, int* wi, int* wj, int* qt0) {
  int i = 0, j = 0;
  for (int w = 0; w + 1 < g.nWh * g.nWw; ++w) {       // (the last window takes what is left)
    const int nt = (num_queries(g, i, j) + ATT_QT - 1) / ATT_QT;
    if (t < nt) break;
    t -= nt;
    if (++j == g.nWw) { j = 0; ++i; }
  }
  *wi = i; *wj = j; *qt0 = t * ATT_QT;
}

// ---- shared device helpers of the forward / backward kernels ------------------------------------------

// T^T = tbl . Q^T for this wave's 16 queries -> Trow[table row] (this lane's query). All table loads are
// issued before the first MFMA (rows beyond NT re-read the last tile), so the product costs one memory
// round trip instead of NT/16.
#define ATT_MAX_NT16 16
template <int HD>
__device__ __forceinline__ void att_table_product(const bf16_t* tbl, int NT, const bf16x8_t (&qf)[AttHd<HD>::KS], float* Trow, int fr, int fq) {
  constexpr int KS = AttHd<HD>::KS, HDP = AttHd<HD>::HDP;
  const int nt16 = NT >> 4;
  for (int t0 = 0; t0 < nt16; t0 += 4) {
    bf16x8_t a[4][KS];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = min(t0 + i, nt16 - 1);
      const bf16_t* tp = tbl + (int64_t)(16 * t + fr) * HDP + 8 * fq;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[i][ks] = *reinterpret_cast<const bf16x8_t*>(tp + 32 * ks);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (t0 + i < nt16) {
        f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], qf[ks], acc, 0, 0, 0);
        *reinterpret_cast<float4*>(Trow + 16 * (t0 + i) + 4 * fq) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      }
    }
  }
}

// Rows [b0, b0 + 64) of T^T = tbl . Q^T for this wave's 16 queries -> Tb[row - b0] (Tb: this lane's query row of the wave's
// [16][ATT_TB_LD] block buffer). Written and read by the same wave only (LDS serves a wave's accesses in order).
template <int HD>
__device__ __forceinline__ void att_table_block(const bf16_t* tbl, int NT, int b0, const bf16x8_t (&qf)[AttHd<HD>::KS], float* Tb, int fr, int fq) {
  constexpr int KS = AttHd<HD>::KS, HDP = AttHd<HD>::HDP;
  const int nt16 = NT >> 4, t0 = b0 >> 4;
  bf16x8_t a[4][KS];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = min(t0 + i, nt16 - 1);
    const bf16_t* tp = tbl + (int64_t)(16 * t + fr) * HDP + 8 * fq;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[i][ks] = *reinterpret_cast<const bf16x8_t*>(tp + 32 * ks);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (t0 + i < nt16) {
      f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], qf[ks], acc, 0, 0, 0);
      *reinterpret_cast<float4*>(Tb + 16 * i + 4 * fq) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  }
}

// This lane's k-step fragments of one [., hd] bf16 row (q, k, v, dO, O): k-step ks covers head dims 32 ks + 8 fq ..+7;
// fragments past the head dim (hd = 80: the upper half of the third k-step) are zero and are never loaded.
template <int HD>
__device__ __forceinline__ void att_load_row_frags(const bf16_t* row, int fq, bool valid, bf16x8_t (&f)[AttHd<HD>::KS]) {
#pragma unroll
  for (int ks = 0; ks < AttHd<HD>::KS; ++ks) {
    const bool in = 32 * ks + 8 * fq < HD;
    f[ks] = *reinterpret_cast<const bf16x8_t*>(row + (in ? 32 * ks + 8 * fq : 8 * fq));
    if (!in || !valid) f[ks] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
  }
}

// 16-B chunk `ch` of a padded key's K and V rows: the qkv bias (fp32) rounded to bf16
struct PaddedKV { u32x4_t k, v; };
__device__ __forceinline__ PaddedKV att_padded_kv(const float* qkv_bias, int D, int head, int hd, int ch) {
  const float* bk = qkv_bias + D + head * hd + ch * 8;
  const float4 a = *reinterpret_cast<const float4*>(bk), b = *reinterpret_cast<const float4*>(bk + 4);
  const float4 c = *reinterpret_cast<const float4*>(bk + D), d = *reinterpret_cast<const float4*>(bk + D + 4);
  PaddedKV r;
  r.k.x = pack_bf16x2(a.x, a.y); r.k.y = pack_bf16x2(a.z, a.w); r.k.z = pack_bf16x2(b.x, b.y); r.k.w = pack_bf16x2(b.z, b.w);
  r.v.x = pack_bf16x2(c.x, c.y); r.v.y = pack_bf16x2(c.z, c.w); r.v.z = pack_bf16x2(d.x, d.y); r.v.w = pack_bf16x2(d.z, d.w);
  return r;
}

// This lane's 8 slots of each 32-slot group of T'[q][.] / scale, split into bf16 hi + lo fragments (B operand
// of the indicator product). Slot j < KH reads T[q][j + qh]; slot KH + j' (j' < KW) reads T[q][Lh + j' + qw].
template <int NSG>
__device__ __forceinline__ void att_slot_fragments(const AttnGeom& g, const float* Trow, int qh, int qw, int fq, int nsg,
                                                   bf16x8_t (&hi)[NSG], bf16x8_t (&lo)[NSG]) {
  const float inv = 1.0f / g.scale;
#pragma unroll
  for (int c = 0; c < NSG; ++c) {
    union { bf16x8_t v; unsigned int w[4]; } h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float x[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int sl = 32 * c + 8 * fq + 2 * i + e;
        const int idx = sl < g.KH ? sl + qh : g.Lh + (sl - g.KH) + qw;
        x[e] = (c < nsg && sl < g.KH + g.KW) ? Trow[idx] * inv : 0.f;
      }
      h.w[i] = pack_bf16x2(x[0], x[1]);
      l.w[i] = pack_bf16x2(x[0] - bf16_lo(h.w[i]), x[1] - bf16_hi(h.w[i]));
    }
    hi[c] = h.v; lo[c] = l.v;
  }
}

// Indicator rows of the chunk's keys: G[key][slot] = 1 for slot == KH-1-kh and slot == KH+KW-1-kw (bf16), zero
// rows beyond Nk. Thread t fills half of key (t >> 1)'s row; images are [group][key][32 slots], 16-B chunks
// XOR-swizzled by gswz(key).
// one bf16 1.0 at slot d of a 4-slot half chunk (d outside [0, 4): nothing) - a 64-bit shift instead of a compare per slot
__device__ __forceinline__ unsigned long long att_one_at(int d) {
  return (unsigned)d < 4u ? (0x3f80ull << (16 * d)) : 0ull;
}
__device__ __forceinline__ void att_stage_indicator(const AttnGeom& g, unsigned char* G_lds, int kc, int Nk, int nsg, int tid) {
  const int key = tid >> 1, kg = kc + key;
  int jh = -1 << 20, jw = -1 << 20;
  if (kg < Nk) { const int kh = att_div(kg, g.KW), kw = kg - kh * g.KW; jh = g.KH - 1 - kh; jw = g.KH + g.KW - 1 - kw; }
  for (int cidx = (tid & 1) * 2 * nsg; cidx < ((tid & 1) + 1) * 2 * nsg; ++cidx) {
    const int c = cidx >> 2, c4 = cidx & 3, s0 = 32 * c + 8 * c4;
    const unsigned long long lo = att_one_at(jh - s0) | att_one_at(jw - s0), hi = att_one_at(jh - s0 - 4) | att_one_at(jw - s0 - 4);
    u32x4_t w;
    w.x = (unsigned)lo; w.y = (unsigned)(lo >> 32); w.z = (unsigned)hi; w.w = (unsigned)(hi >> 32);
    *reinterpret_cast<u32x4_t*>(G_lds + c * (ATT_KC * ATT_GROW) + key * ATT_GROW + ((c4 ^ gswz(key)) << 4)) = w;
  }
}
