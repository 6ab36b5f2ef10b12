// bf16 MFMA GEMM, big-tile variant for the encoder's M = 2*B*T token matrices (same contract and epilogues as gemm.hip):
//
//   C[m,n] = sum_k A[m,k] W[n,k] + bias[n] + lr_scale * U[m,:] . V[g(m)][n,:] -> act -> + residual -> out
//
// Why a second kernel: the 128x128 kernel (2 workgroups / CU, one tile of prefetch) moves 1 byte from L2 to LDS per 64 FLOP
// and keeps 64 KB in flight per CU - in the training step its main loop waits on those loads (in-kernel stamps: 42 % of the
// MFMA cycles at K = 768) and N = 768 leaves a third of the chip idle (384 tiles on 512 slots). Here ONE 8-wave workgroup
// per CU owns a 256 x 32*NF tile (NF = 3 / 4 / 6 / 8 / 9 -> N-tiles of 96 ... 288 columns, chosen on the host so that the
// tile count is a whole number of rounds of the 256 CUs: N = 768 -> 256 tiles of 256x96, N = 2304 -> 256 tiles of 256x288,
// N = 3072 -> 512 tiles of 256x192), i.e. 70 - 135 FLOP per staged byte, and the K loop runs on a ring of 2 - 3 LDS stages
// filled by global_load_lds two tiles ahead: counted s_waitcnt vmcnt(N) + a raw s_barrier per K-tile, never vmcnt(0) in the
// steady state.
//
// Layout: 8 waves = 4 (M) x 2 (N); a wave owns 64 rows x 16*NF columns as 4 x NF mfma_f32_16x16x32_bf16 tiles. LDS stage =
// [256 A rows | 32*NF W rows] x 128 B (BK = 64), lane-linear 1-KB pieces, XOR-swizzled on the SOURCE address; fragments by
// ds_read_b128. As in gemm.hip the W rows are the MFMA "A" operand and are read in a permuted order so that a lane ends up
// owning runs of consecutive output columns: the NF column fragments of a wave are cut into groups of G = 4, 2 or 1
// fragments (NF = 6 -> 4 + 2, 9 -> 4 + 4 + 1, 3 -> 2 + 1) and inside a group lane quarter fq owns columns 4G*fq .. 4G*fq+4G-1,
// so bias / aux / residual loads and the stores are 16-byte (G >= 2) or 8-byte (G = 1, bf16) accesses.
#include "gemm_params.h"
#include <type_traits>

#define BIG_BM 256
#define BIG_BK 64

template <int NF> struct BigCfg {
  static constexpr int BN = 32 * NF;
  static constexpr int ROWS = BIG_BM + BN;
  static constexpr int STAGE_BYTES = ROWS * BIG_BK * 2;
  static constexpr int STAGES = 3 * STAGE_BYTES <= 160 * 1024 ? 3 : 2;
  static constexpr int PIECES = ROWS / 8;                 // 1-KB pieces (8 rows x 128 B) per stage
  static constexpr int PPW = (PIECES + 7) / 8;            // pieces per wave (the last ones duplicate piece PIECES-1: same bytes, same place)
  static constexpr int N4 = NF / 4, REM = NF % 4;         // groups: N4 of 4 fragments, then one of 2 (REM >= 2), then one of 1 (REM odd)
  static constexpr int NG = N4 + (REM >= 2 ? 1 : 0) + (REM & 1);
};
// group gi of a wave's NF fragments: size and first fragment
template <int NF> __device__ __forceinline__ constexpr int grp_size(int gi) {
  return gi < BigCfg<NF>::N4 ? 4 : ((gi == BigCfg<NF>::N4 && BigCfg<NF>::REM >= 2) ? 2 : 1);
}
template <int NF> __device__ __forceinline__ constexpr int grp_f0(int gi) {
  return gi <= BigCfg<NF>::N4 ? 4 * gi : 4 * BigCfg<NF>::N4 + 2;
}
// chunk XOR of W row `o` (0 .. 16*NF-1 inside its wave's column range): conflict-free ds_read_b128 for the permuted row sets
template <int NF> __device__ __forceinline__ int big_swz_w(int o) {
  constexpr int c4 = 64 * BigCfg<NF>::N4;
  if (o < c4) return (((o >> 4) & 3) << 1) | ((o >> 1) & 1);
  const int o2 = o - c4;
  if (BigCfg<NF>::REM >= 2 && o2 < 32) return (((o2 >> 3) & 3) << 1) | ((o2 >> 1) & 1);
  return o & 7;
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N) - every index a constant, so per-unit register arrays
// stay registers (a runtime-indexed array goes to scratch memory)
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int NF, int ACT, bool LORA>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(const GemmParams p) {
  using C = BigCfg<NF>;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[C::STAGES * C::STAGE_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  MF_STAMP(0);

  // XCD-aware bijective remap of the workgroup id (8 XCDs, round-robin dispatch): each XCD walks a contiguous run of tiles
  const int nwg = p.tiles_m * p.tiles_n;
  int wg;
  {
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int tm = wg / p.tiles_n, tn = wg - tm * p.tiles_n;
  const int m0 = tm * BIG_BM, n0 = tn * C::BN;
  const int grp = p.n_groups > 1 ? (int)(((unsigned)m0 / (unsigned)p.rows_per_group) % (unsigned)p.n_groups) : 0;
  const bf16_t* Wg = p.W + (int64_t)grp * p.w_group_stride;
  const float* biasg = p.bias ? p.bias + (int64_t)grp * p.bias_group_stride : nullptr;

  // ---- staging: piece j = wave + 8*i covers stage rows [8j, 8j+8); lane -> row 8j + (lane>>3), destination chunk lane&7
  const int srow = lane >> 3, sch = lane & 7;
  const bf16_t* src[C::PPW];
  int dst_off[C::PPW];
#pragma unroll
  for (int i = 0; i < C::PPW; ++i) {
    int j = wave + 8 * i;
    j = j < C::PIECES ? j : C::PIECES - 1;
    const int row = 8 * j + srow;
    if (row < BIG_BM) {
      src[i] = p.A + (int64_t)(m0 + row) * p.lda + ((sch ^ (row & 7)) << 3);
    } else {
      const int jw = row - BIG_BM;                      // W row of the tile
      const int o = jw % (16 * NF);
      src[i] = Wg + (int64_t)(n0 + jw) * p.ldw + ((sch ^ big_swz_w<NF>(o)) << 3);
    }
    dst_off[i] = __builtin_amdgcn_readfirstlane(j * 1024);
  }
  auto stage = [&](int kt, int buf) {
#ifdef GEMM_NOLOAD      // diagnostic: only the first tiles are fetched - the main loop without its global traffic
    if (kt >= C::STAGES) return;
#endif
    unsigned char* base = lds + buf * C::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < C::PPW; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[i] + kt * BIG_BK), (lds_ptr_t)(base + dst_off[i]), 16, 0, 0);
  };

  // ---- fragment read offsets (bytes inside a stage, without the per-k-step chunk XOR)
  const int fr = lane & 15, fq = lane >> 4;
  int x_off[4], x_sw[4], w_off[NF], w_sw[NF];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int xr = wm * 64 + i * 16 + fr;
    x_off[i] = xr * 128; x_sw[i] = xr & 7;
  }
#pragma unroll
  for (int gi = 0; gi < C::NG; ++gi) {
    const int G = grp_size<NF>(gi), f0 = grp_f0<NF>(gi);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i >= G) break;
      const int o = 16 * f0 + 4 * G * (fr >> 2) + 4 * i + (fr & 3);
      w_off[f0 + i] = (BIG_BM + wn * 16 * NF + o) * 128;
      w_sw[f0 + i] = big_swz_w<NF>(o);
    }
  }

  f32x4_t acc[NF][4];   // [fragment][mi]
#pragma unroll
  for (int a = 0; a < NF; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- main loop over K-tiles: ring of STAGES buffers filled by global_load_lds ahead of the MFMAs
  const int nt = p.K / BIG_BK;
  if constexpr (NF <= 6) {
    // Fragment registers are double-buffered at k-step granularity ACROSS the barrier: while the MFMAs of one k-step run, the
    // ds_read_b128 of the next one are in flight - the second k-step of the tile, or (after the barrier that publishes tile
    // t + 1) the first k-step of the next tile. (With all reads of a tile issued behind its barrier and waited for together,
    // the 8 waves read LDS and then issue MFMAs in phase: a k-tile took ~1.8x its MFMA time.)
    // Ring: tiles 0 .. STAGES-1 requested up front; at the barrier inside iteration t every wave holds all of tile t in
    // registers, so tile t + STAGES goes into tile t's buffer.
#pragma unroll
    for (int s = 0; s < C::STAGES; ++s)
      if (s < nt) stage(s, s);
    bf16x8_t xa[4], wa[NF], xb[4], wb[NF];
    auto rd = [&](const unsigned char* base, int ks, bf16x8_t (&x)[4], bf16x8_t (&w)[NF]) {
      const int ch = fq + 4 * ks;
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = *reinterpret_cast<const bf16x8_t*>(base + x_off[i] + ((ch ^ x_sw[i]) << 4));
#pragma unroll
      for (int f = 0; f < NF; ++f) w[f] = *reinterpret_cast<const bf16x8_t*>(base + w_off[f] + ((ch ^ w_sw[f]) << 4));
    };
    auto mm = [&](const bf16x8_t (&x)[4], const bf16x8_t (&w)[NF], int f0, int f1) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (f < f0 || f >= f1) continue;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[f][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[f], x[mi], acc[f][mi], 0, 0, 0);
      }
    };
    // the 4 + NF reads of the next k-step are woven between the remaining 4 * (NF - 1) MFMAs of this one (PER MFMAs, then a read):
    // issued in one burst, the 8 waves' reads queue at the LDS together and the MFMAs behind them in program order wait
    auto weave = [&]() {
      constexpr int NR = 4 + NF, NM = 4 * (NF - 1), PER = NM / NR > 0 ? NM / NR : 1;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NM - PER * NR > 0 ? NM - PER * NR : 0, 0);
    };
    // the same with the PPW global_load_lds of the ring's next tile dealt out one per read slot: issued in one burst behind the
    // barrier, the 8 waves' ~50 requests fill the CU's vector-memory queue together and every wave's MFMAs wait in program order
    // behind its own third to sixth request
    auto weave_v = [&]() {
      constexpr int NR = 4 + NF, NM = 4 * (NF - 1), PER = NM / NR > 0 ? NM / NR : 1;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (i < C::PPW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NM - PER * NR > 0 ? NM - PER * NR : 0, 0);
    };
    // tile 0 landed: at most the younger tiles of this wave outstanding
    if (nt >= C::STAGES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::PPW * (C::STAGES - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    MF_STAMP(1);
    rd(lds, 0, xa, wa);
    // (in each half the first MFMAs go ahead of the next reads: hipcc's wait for fragments whose reads crossed a loop edge or
    // an asm statement is lgkmcnt(0) - it must not cover the reads just issued; the last tile is peeled so that no branch
    // joins inside the body)
    auto body = [&](auto with_stage, int t) {
      const unsigned char* base = lds + (t % C::STAGES) * C::STAGE_BYTES;
      mm(xa, wa, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      rd(base, 1, xb, wb);
      mm(xa, wa, 1, NF);
      weave();
      __builtin_amdgcn_sched_barrier(0);
      // tile t + 1 landed (tiles t + 2 .. t + STAGES - 1 may still be in flight) and this wave's reads of tile t returned
#ifdef GEMM_NOWAIT      // diagnostic (wrong results): tiles are fetched but never waited for
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
      if (t + C::STAGES - 1 < nt && C::STAGES > 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::PPW * (C::STAGES - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      mm(xb, wb, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (decltype(with_stage)::value) {
        // fragment reads and ring requests alternate in program order: the compiler keeps LDS reads and LDS-DMA writes in the
        // order it finds them (it cannot tell the two stage buffers apart)
        const unsigned char* nb = lds + ((t + 1) % C::STAGES) * C::STAGE_BYTES;
        unsigned char* sb = lds + (t % C::STAGES) * C::STAGE_BYTES;
        const int kt = t + C::STAGES;
#pragma unroll
        for (int i = 0; i < 4 + NF || i < C::PPW; ++i) {
          if (i < 4) xa[i] = *reinterpret_cast<const bf16x8_t*>(nb + x_off[i] + ((fq ^ x_sw[i]) << 4));
          else if (i < 4 + NF) wa[i - 4] = *reinterpret_cast<const bf16x8_t*>(nb + w_off[i - 4] + ((fq ^ w_sw[i - 4]) << 4));
          if (i < C::PPW) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[i] + kt * BIG_BK), (lds_ptr_t)(sb + dst_off[i]), 16, 0, 0);
        }
      } else {
        rd(lds + ((t + 1) % C::STAGES) * C::STAGE_BYTES, 0, xa, wa);
      }
      mm(xb, wb, 1, NF);
      if constexpr (decltype(with_stage)::value) weave_v(); else weave();
      __builtin_amdgcn_sched_barrier(0);
    };
    {
      int t = 0;
      for (; t + C::STAGES < nt; ++t) body(std::true_type{}, t);       // the ring's next tile exists
      for (; t + 1 < nt; ++t) body(std::false_type{}, t);
    }
    mm(xa, wa, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    rd(lds + ((nt - 1) % C::STAGES) * C::STAGE_BYTES, 1, xb, wb);
    mm(xa, wa, 1, NF);
    weave();
    __builtin_amdgcn_sched_barrier(0);
    mm(xb, wb, 0, NF);
  } else {
#pragma unroll
    for (int s = 0; s < C::STAGES - 1; ++s)
      if (s < nt) stage(s, s);
    auto body = [&](auto with_stage, int t) {
      // tile t has landed once at most the (STAGES-2) younger tiles of this wave are outstanding (vmcnt counts in issue order)
      if (C::STAGES == 3 && t + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::PPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // every wave's pieces of tile t are in LDS; every wave is done reading tile t-1's buffer
      asm volatile("" ::: "memory");
#ifdef GEMM_STAMPS
      if (t == 0) MF_STAMP(1);
#endif
      const unsigned char* base = lds + (t % C::STAGES) * C::STAGE_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t xf[4], wf[NF];
        const int ch = fq + 4 * ks;
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const bf16x8_t*>(base + x_off[i] + ((ch ^ x_sw[i]) << 4));
#pragma unroll
        for (int f = 0; f < NF; ++f) wf[f] = *reinterpret_cast<const bf16x8_t*>(base + w_off[f] + ((ch ^ w_sw[f]) << 4));
        // the ring's next tile is requested piece by piece between the MFMAs of the first k-step (one burst of PPW requests
        // per wave behind the barrier fills the CU's vector-memory queue and holds the MFMAs behind it in program order)
        if (decltype(with_stage)::value && ks == 0) stage(t + C::STAGES - 1, (t + C::STAGES - 1) % C::STAGES);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
            acc[f][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[f], xf[mi], acc[f][mi], 0, 0, 0);
        if (decltype(with_stage)::value && ks == 0) {
          constexpr int PERV = (4 * NF) / (C::PPW + 1) > 0 ? (4 * NF) / (C::PPW + 1) : 1;
#pragma unroll
          for (int i = 0; i < C::PPW; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, PERV, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 4 * NF - PERV * C::PPW, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    {
      int t = 0;
      for (; t + C::STAGES - 1 < nt; ++t) body(std::true_type{}, t);
      for (; t < nt; ++t) body(std::false_type{}, t);
    }
  }

  MF_STAMP(2);
  // ---- epilogue. Lane owns rows m(mi) = m0 + wm*64 + mi*16 + fr; in group gi columns cb(gi) .. cb(gi) + 4G - 1.
  int mrow[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) mrow[mi] = m0 + wm * 64 + mi * 16 + fr;
  const int ncol0 = n0 + wn * 16 * NF;

  if (LORA) {
    // Low-rank update = one extra MFMA k-step per 8 ranks whose 32 k-slots hold the rank index three times: slots [0,8) U_hi.V_hi,
    // [8,16) U_lo.V_hi, [16,24) U_hi.V_lo (hi = bf16(x), lo = bf16(x - hi)), as in gemm.hip: lane quarter fq owns product fq of
    // the pass's ranks [8*pass, 8*pass + 8). R = 4 / 8 (the default LoRA rank 4: stacked factors of 8 columns) take one pass, LoRA
    // ranks up to 16 (R <= 32) up to four. The raw fp32 factor rows are fetched here, one column group at a time (registers:
    // NF = 9 holds 144 accumulators).
    const int R = p.R;
    const float* vg = p.V + (int64_t)grp * p.v_group_stride;
    const float on = fq <= 2 ? 1.0f : 0.0f;
    for (int lk0 = 0; lk0 < R; lk0 += 8) {
      const int lk1 = lk0 + 4 < R ? lk0 + 4 : lk0;
      const float tail = lk0 + 4 < R ? 1.0f : 0.0f;
      auto load_raw = [&](float4 (&dst)[2], const float* s) {
        dst[0] = *reinterpret_cast<const float4*>(s + lk0);
        dst[1] = *reinterpret_cast<const float4*>(s + lk1);
      };
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
      bf16x8_t um[4];
      {
        float4 uraw[4][2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) load_raw(uraw[mi], p.U + (int64_t)mrow[mi] * p.ldu);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) um[mi] = conv(uraw[mi], on * p.lr_scale, fq == 1);
      }
#pragma unroll
      for (int gi = 0; gi < C::NG; ++gi) {
        const int G = grp_size<NF>(gi), f0 = grp_f0<NF>(gi);
        float4 vraw[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i >= G) break;
          const int n = ncol0 + 16 * f0 + 4 * G * (fr >> 2) + 4 * i + (fr & 3);      // W-row order of the accumulator tiles
          load_raw(vraw[i], vg + (int64_t)n * R);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i >= G) break;
          const bf16x8_t vf = conv(vraw[i], on, fq == 2);
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
            acc[f0 + i][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, um[mi], acc[f0 + i][mi], 0, 0, 0);
        }
      }
    }
  }

  // Store loop. A unit = two row tiles (mh, mh + 1) of one column group. vmcnt retires loads and stores IN ORDER, so a load
  // requested after a store waits for that store's acknowledgement: with the aux / residual / bias loads of every unit issued
  // behind the previous unit's stores, a tile's epilogue was a chain of 4 - 6 dependent memory round trips under the chip-wide
  // store burst (lin2: 7.9 us for 98 KB in + 98 KB out per tile; the load-free plain epilogue: 1.9). Now every bias is
  // requested before the first store and unit u + 1's operands before unit u's stores: one exposed round trip per tile.
  float bv[C::NG][16];
  if (biasg) {
#pragma unroll
    for (int gi = 0; gi < C::NG; ++gi) {
      const int G = grp_size<NF>(gi), f0 = grp_f0<NF>(gi);
      const int cb = ncol0 + 16 * f0 + 4 * G * fq;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q >= G) break;
        const float4 b4 = *reinterpret_cast<const float4*>(biasg + cb + 4 * q);
        bv[gi][4 * q] = b4.x; bv[gi][4 * q + 1] = b4.y; bv[gi][4 * q + 2] = b4.z; bv[gi][4 * q + 3] = b4.w;
      }
    }
  }
  struct EpiOperands { unsigned int aw[2][8]; float4 rs[2][4]; };
  // (an instantiation that reads aux holds those registers instead: its residual - no layer of the model has both - is
  // fetched unit by unit)
  constexpr bool RES_AHEAD = !(ACT == 2 || ACT == 4);
  auto load_res = [&](auto uc, EpiOperands& b) __attribute__((always_inline)) {
    constexpr int u = decltype(uc)::value;
    constexpr int gi = u >> 1, mh = (u & 1) * 2;
    constexpr int G = grp_size<NF>(gi), f0 = grp_f0<NF>(gi);
    const int cb = ncol0 + 16 * f0 + 4 * G * fq;
#pragma unroll
    for (int mj = 0; mj < 2; ++mj) {
      const int m = mrow[mh + mj];
      const int rr = p.res_row_mod > 0 ? (int)((unsigned)m % (unsigned)p.res_row_mod) : m;
      const float4* rp = reinterpret_cast<const float4*>(p.residual + (int64_t)rr * p.ldres + cb);
#pragma unroll
      for (int q = 0; q < 4; ++q) { if (q >= G) break; b.rs[mj][q] = rp[q]; }
    }
  };
  auto load_unit = [&](auto uc, EpiOperands& b) __attribute__((always_inline)) {
    constexpr int u = decltype(uc)::value;
    constexpr int gi = u >> 1, mh = (u & 1) * 2;
    constexpr int G = grp_size<NF>(gi), f0 = grp_f0<NF>(gi);
    constexpr int VL = 4 * G;
    const int cb = ncol0 + 16 * f0 + VL * fq;            // first column of this lane in the group
    if (ACT == 2 || ACT == 4) {
#pragma unroll
      for (int mj = 0; mj < 2; ++mj) {
        const bf16_t* ap = p.aux + (int64_t)mrow[mh + mj] * p.ldaux + cb;
        if (G == 1) {
          const u32x2_t a2 = *reinterpret_cast<const u32x2_t*>(ap);
          b.aw[mj][0] = a2.x; b.aw[mj][1] = a2.y;
        } else {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (8 * h >= VL) break;
            const u32x4_t a4 = reinterpret_cast<const u32x4_t*>(ap)[h];
            b.aw[mj][4 * h] = a4.x; b.aw[mj][4 * h + 1] = a4.y; b.aw[mj][4 * h + 2] = a4.z; b.aw[mj][4 * h + 3] = a4.w;
          }
        }
      }
    }
    if (RES_AHEAD && p.residual) load_res(uc, b);
  };
  auto finish_unit = [&](auto uc, const EpiOperands& b) __attribute__((always_inline)) {
    constexpr int u = decltype(uc)::value;
    constexpr int gi = u >> 1, mh = (u & 1) * 2;
    constexpr int G = grp_size<NF>(gi), f0 = grp_f0<NF>(gi);
    constexpr int VL = 4 * G;
    const int cb = ncol0 + 16 * f0 + VL * fq;
    float v[2][16];
#pragma unroll
    for (int mj = 0; mj < 2; ++mj)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (e >= VL) break;
        v[mj][e] = acc[f0 + (e >> 2)][mh + mj][e & 3] + (biasg ? bv[gi][e] : 0.f);
      }
    auto store_bf16 = [&](bf16_t* dst, const float (&x)[16]) {
      if (G == 1) {
        u32x2_t w; w.x = pack_bf16x2(x[0], x[1]); w.y = pack_bf16x2(x[2], x[3]);
        *reinterpret_cast<u32x2_t*>(dst) = w;
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (8 * h >= VL) break;
          u32x4_t w;
          w.x = pack_bf16x2(x[8 * h], x[8 * h + 1]); w.y = pack_bf16x2(x[8 * h + 2], x[8 * h + 3]);
          w.z = pack_bf16x2(x[8 * h + 4], x[8 * h + 5]); w.w = pack_bf16x2(x[8 * h + 6], x[8 * h + 7]);
          reinterpret_cast<u32x4_t*>(dst)[h] = w;
        }
      }
    };
    if (p.out_pre) {
#pragma unroll
      for (int mj = 0; mj < 2; ++mj) store_bf16(p.out_pre + (int64_t)mrow[mh + mj] * p.ldpre + cb, v[mj]);
    }
    if (ACT == 1) {
#pragma unroll
      for (int mj = 0; mj < 2; ++mj)
#pragma unroll
        for (int e = 0; e < 16; ++e) { if (e >= VL) break; v[mj][e] = gelu_f(v[mj][e]); }
    } else if (ACT == 2 || ACT == 4) {
#pragma unroll
      for (int mj = 0; mj < 2; ++mj)
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2) {
          if (2 * e2 >= VL) break;
          const float lo = bf16_lo(b.aw[mj][e2]), hi = bf16_hi(b.aw[mj][e2]);
          if (ACT == 2) { v[mj][2 * e2] *= gelu_grad_f(lo); v[mj][2 * e2 + 1] *= gelu_grad_f(hi); }
          else {
            v[mj][2 * e2] = (lo > 0.f && lo < 6.f) ? v[mj][2 * e2] : 0.f;
            v[mj][2 * e2 + 1] = (hi > 0.f && hi < 6.f) ? v[mj][2 * e2 + 1] : 0.f;
          }
        }
    } else if (ACT == 3) {
#pragma unroll
      for (int mj = 0; mj < 2; ++mj)
#pragma unroll
        for (int e = 0; e < 16; ++e) { if (e >= VL) break; v[mj][e] = fminf(fmaxf(v[mj][e], 0.f), 6.f); }
    }
    if (p.residual) {
      if constexpr (!RES_AHEAD) {
        EpiOperands late;
        load_res(uc, late);
#pragma unroll
        for (int mj = 0; mj < 2; ++mj)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (q >= G) break;
            v[mj][4 * q] += late.rs[mj][q].x; v[mj][4 * q + 1] += late.rs[mj][q].y; v[mj][4 * q + 2] += late.rs[mj][q].z; v[mj][4 * q + 3] += late.rs[mj][q].w;
          }
      } else
#pragma unroll
      for (int mj = 0; mj < 2; ++mj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (q >= G) break;
          v[mj][4 * q] += b.rs[mj][q].x; v[mj][4 * q + 1] += b.rs[mj][q].y; v[mj][4 * q + 2] += b.rs[mj][q].z; v[mj][4 * q + 3] += b.rs[mj][q].w;
        }
    }
    if (ACT == 5) {
#pragma unroll
      for (int mj = 0; mj < 2; ++mj)
#pragma unroll
        for (int e = 0; e < 16; ++e) { if (e >= VL) break; v[mj][e] = fminf(fmaxf(v[mj][e], 0.f), 6.f); }
    }
#pragma unroll
    for (int mj = 0; mj < 2; ++mj) {
      const int64_t m = mrow[mh + mj];
      if (p.out_dtype == MFNET_BF16) {
        store_bf16((bf16_t*)p.out + m * p.ldc + cb, v[mj]);
      } else {
        float4* dst = reinterpret_cast<float4*>((float*)p.out + m * p.ldc + cb);
#pragma unroll
        for (int q = 0; q < 4; ++q) { if (q >= G) break; dst[q] = make_float4(v[mj][4 * q], v[mj][4 * q + 1], v[mj][4 * q + 2], v[mj][4 * q + 3]); }
      }
    }
  };
  constexpr int NU = 2 * C::NG;
  if constexpr (NF <= 6) {
    EpiOperands eo_even, eo_odd;
    load_unit(std::integral_constant<int, 0>{}, eo_even);
    static_for<0, NU>([&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      if constexpr (u + 1 < NU) load_unit(std::integral_constant<int, u + 1>{}, (u & 1) ? eo_even : eo_odd);
      finish_unit(uc, (u & 1) ? eo_odd : eo_even);
    });
  } else {
    // 128 - 144 accumulators leave no room for a second unit's operands: unit by unit (the encoder's NF = 9 shape, the qkv
    // projection, has a bias only - requested above)
    static_for<0, NU>([&](auto uc) __attribute__((always_inline)) {
      EpiOperands eo;
      load_unit(uc, eo);
      finish_unit(uc, eo);
    });
  }
#ifdef GEMM_STAMPS
  MF_STAMP(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  MF_STAMP(4);
#endif
}

template <int NF>
static void big_launch_nf(const GemmParams& p, int act, bool lora, dim3 grid, hipStream_t stream) {
#define MF_BIG_LAUNCH(A_, L_) hipLaunchKernelGGL((gemm_big_kernel<NF, A_, L_>), grid, dim3(512), 0, stream, p)
#define MF_BIG_CASE(A_) case A_: if (lora) MF_BIG_LAUNCH(A_, true); else MF_BIG_LAUNCH(A_, false); break;
  switch (act) { MF_BIG_CASE(0) MF_BIG_CASE(1) MF_BIG_CASE(2) MF_BIG_CASE(3) MF_BIG_CASE(4) MF_BIG_CASE(5) }
#undef MF_BIG_CASE
#undef MF_BIG_LAUNCH
}

// Host-side choice of the N-tile: rounds of the 256 CUs x columns per tile (+ a fixed per-round cost in column units), ties
// to the wider tile. Returns 0 when the shape is left to the 128x128 kernel.
static int big_choose_nf(int64_t M, int64_t N) {
  static const int cand[] = {9, 8, 6, 4, 3};
  int best = 0;
  int64_t best_cost = 0;
  const int64_t tm = M / BIG_BM;
  for (int nf : cand) {
    if (N % (32 * nf)) continue;
    const int64_t tiles = tm * (N / (32 * nf));
    const int64_t rounds = (tiles + 255) / 256;
    if (tiles < 192) continue;                     // not enough tiles for 256 CUs: the small-tile kernel spreads better
    const int64_t cost = rounds * (32 * nf + 48);
    if (!best || cost < best_cost) { best = nf; best_cost = cost; }
  }
  return best;
}

int gemm_big_launch(const GemmParams& p0, int act, bool lora, int tile, hipStream_t stream) {
  if (tile < 0) return 0;
  const bool shape_ok = p0.M % BIG_BM == 0 && p0.K % BIG_BK == 0 && p0.K >= BIG_BK && p0.k_splits == 1 &&
                        (p0.n_groups <= 1 || p0.rows_per_group % BIG_BM == 0) && p0.N % 32 == 0 &&
                        (!p0.bias || ((uintptr_t)p0.bias % 16 == 0 && p0.bias_group_stride % 4 == 0)) && (!p0.residual || (uintptr_t)p0.residual % 16 == 0) &&
                        (!p0.aux || ((uintptr_t)p0.aux % 16 == 0)) && (!p0.out_pre || ((uintptr_t)p0.out_pre % 16 == 0)) &&
                        (!lora || ((uintptr_t)p0.U % 16 == 0 && (uintptr_t)p0.V % 16 == 0 && p0.v_group_stride % 4 == 0 && p0.ldu % 4 == 0));
  if (!shape_ok) return 0;
  int nf = tile;
  if (nf == 0) nf = big_choose_nf(p0.M, p0.N);
  if (!(nf == 3 || nf == 4 || nf == 6 || nf == 8 || nf == 9) || p0.N % (32 * nf)) return 0;
  GemmParams p = p0;
  p.tiles_m = p.M / BIG_BM; p.tiles_n = p.N / (32 * nf);
  const int64_t nwg = (int64_t)p.tiles_m * p.tiles_n;
  if (nwg >= (1ll << 31)) return 0;
  const dim3 grid((unsigned)nwg);
  switch (nf) {
    case 3: big_launch_nf<3>(p, act, lora, grid, stream); break;
    case 4: big_launch_nf<4>(p, act, lora, grid, stream); break;
    case 6: big_launch_nf<6>(p, act, lora, grid, stream); break;
    case 8: big_launch_nf<8>(p, act, lora, grid, stream); break;
    default: big_launch_nf<9>(p, act, lora, grid, stream); break;
  }
  return 1;
}
