// "Ping-pong" bf16 MFMA GEMM for gfx950: 256 x (128 NB) x 64 tile, eight waves in two groups that alternate, on every
// SIMD, between an MFMA segment and a load segment.
//
// Why a second pipeline: in gemm.hip's one-barrier-per-K-step loop all eight waves of a workgroup do the same thing at the
// same time, so the MFMA pipe, the LDS and the address unit that feeds the LDS-DMA are each ~50 % busy and their times add
// (profiles/archive/r01f_*).  Here the waves with wr = 0 (one per SIMD) and the waves with wr = 1 (their SIMD partners) run the same
// program one barrier apart: while one group issues its 8 MFMAs of a phase (256 pipe cycles), the other group issues the LDS
// fragment reads of ITS next phase and its share of the operand DMA.  The matrix pipe of a SIMD is handed from one wave to
// the other at every barrier and never waits for a load.  (CDNA "8-phase" schedule; hardware facts in
// /opt/skills/guides/MI355X_MICROARCH.md "Two waves per SIMD".)
//
// Wave (wr, wc) of the 2 x 4 grid owns rows wr*128 .. +127 and columns wc*(32 NB) .. of the tile: 4 x NB MFMA 32x32x16
// blocks, walked per 64-deep K-tile as quadrant phases  (A-sub a: 64 rows) x (B-sub b: 32 columns) = 8 MFMAs each:
//     NB = 2:  (0,0) (0,1) (1,1) (1,0)      NB = 1:  (0,0) (1,0)
// so every operand sub-tile is read from LDS exactly ONCE per K-tile (24 / 20 ds_read_b128 per 32 / 16 MFMAs) and kept in
// registers across the phases that reuse it.
//
// LDS: two K-tile buffers, each cut into 16-KiB "half-tiles" = the [128][64] image of one A-sub (both wave rows) or one
// B-sub (all four wave columns).  One half-tile is requested per phase, one K-tile ahead (buffer_load_dwordx4 ... lds, two
// 1-KiB pieces per wave), so two to three half-tiles are always in flight behind a counted vmcnt; a half-tile is waited for
// one phase before the phase that reads it and restaged four phases after its last read (both margins include the
// one-barrier lag of the second group).
//
// Forms, epilogues and the C ABI are those of gemm.hip (cocodr_gemm with impl 13 / 14).
#include <stdlib.h>

#include "common.h"
#include "gemm_tile.h"
#include "score_filter.h"

namespace cocodr_gemm_pp {
using namespace cocodr_gemm_v2;

constexpr int BM = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;  // 16 KiB: [128 rows][64 k] (or [64 k][128 columns]) bf16
constexpr int NTHREADS = 512;

template <int NB>
struct Shape {
  static constexpr int BN = 128 * NB;
  static constexpr int NTYPE = 2 + NB;              // half-tiles per K-tile
  static constexpr int NPHASE = 2 * NB;             // phases per K-tile
  static constexpr int KT_BYTES = NTYPE * HALF_BYTES;
  static constexpr int RING_BYTES = 2 * KT_BYTES;
  static constexpr int CT_LD = BN + 4;              // fp32 epilogue tile leading dimension
  static constexpr int EPI_BYTES = 128 * CT_LD * 4 + 8 * BN * 4;  // one 128-row pass + one column-sum row per wave
  static constexpr int LDS_BYTES = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;
};

// ---- half-tile "types": which rows / columns of the workgroup tile a half-tile image holds
// NB = 2 (staging order A0 B0 B1 A1):  type 0 = A-sub 0, 1 = B-sub 0, 2 = B-sub 1, 3 = A-sub 1
// NB = 1 (staging order A0 B0 A1):     type 0 = A-sub 0, 1 = B-sub 0, 2 = A-sub 1
template <int NB>
__device__ __forceinline__ constexpr bool type_is_a(int ty) { return NB == 2 ? (ty == 0 || ty == 3) : (ty == 0 || ty == 2); }
template <int NB>
__device__ __forceinline__ constexpr int type_sub(int ty) { return NB == 2 ? (ty >= 2 ? 1 : 0) : (ty == 2 ? 1 : 0); }

// local row / column r (0..127) of a half-tile image -> row / column of the workgroup tile
template <int NB>
__device__ __forceinline__ int a_tile_row(int r, int sub) { return (r >> 6) * 128 + sub * 64 + (r & 63); }
template <int NB>
__device__ __forceinline__ int b_tile_col(int c, int sub) { return (c >> 5) * (32 * NB) + sub * 32 + (c & 31); }

// byte offset (from the operand's batch base) of the 16-B chunk that must land at linear chunk p of a half-tile image
template <int TR, int NB, bool IS_A>
__device__ __forceinline__ uint32_t src_off(int p, int sub, int r0, int ld) {
  if (TR == 0) {  // image [128 rows][64 k], 8 chunks per row
    const int row = p >> 3, ch = (p & 7) ^ swz_rows<BK>(row);
    const int g = r0 + (IS_A ? a_tile_row<NB>(row, sub) : b_tile_col<NB>(row, sub));
    return (uint32_t)((g * ld + ch * 8) * 2);
  } else {        // image [64 k][128 columns], 16 chunks per k row
    const int row = p >> 4, ch = (p & 15) ^ swz_cols<128>(row);
    const int lc = ch * 8;  // 8 consecutive columns never straddle a 32- or 64-column group
    const int g = r0 + (IS_A ? a_tile_row<NB>(lc, sub) : b_tile_col<NB>(lc, sub));
    return (uint32_t)((row * ld + g) * 2);
  }
}

// LDS fragment reads of one K-sub-step with an extra immediate offset (the half-tile's place in the K-tile buffer)
template <int TR, int NF, int S, int OFF, int A = 0>
__device__ __forceinline__ void pp_frags_issue(const uint32_t (&cur)[4], FragSet<TR, NF>& f) {
  if constexpr (A < NF) {
    if constexpr (TR == 0) {
      asm_ds_read_b128<OFF + A * 32 * BK * 2>(f.q[A], cur[S]);
    } else {
      constexpr int o = OFF + S * 16 * 128 * 2;
      asm_ds_read_tr16<o>(f.lo[A], cur[A]);
      asm_ds_read_tr16<o + 4 * 128 * 2>(f.hi[A], cur[A]);
    }
    pp_frags_issue<TR, NF, S, OFF, A + 1>(cur, f);
  }
}
template <int TR, int NF, int OFF>
__device__ __forceinline__ void pp_read_sub(const uint32_t (&cur)[4], FragSet<TR, NF> (&f)[4]) {
#if defined(COCODR_ABL_NO_LDSREAD)
  return;
#endif
  pp_frags_issue<TR, NF, 0, OFF>(cur, f[0]);
  pp_frags_issue<TR, NF, 1, OFF>(cur, f[1]);
  pp_frags_issue<TR, NF, 2, OFF>(cur, f[2]);
  pp_frags_issue<TR, NF, 3, OFF>(cur, f[3]);
}

// 8 MFMAs of one quadrant phase: acc[2 asub + i][bsub] += A-sub(i, ks) x B-sub(ks); operands swapped so that a lane ends
// with 4 consecutive output columns of one row (the epilogue's layout, as in gemm.hip)
// F16: the 16-bit operands are IEEE half instead of bfloat16 (same fragment layout and rate; the search's split-precision
// score GEMM, score.hip)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int TA, int TB, bool F16, class MID>
__device__ __forceinline__ void pp_mfma(const FragSet<TA, 2> (&fa)[4], const FragSet<TB, 1> (&fb)[4], f32x16& c0, f32x16& c1, MID&& mid) {
#if defined(COCODR_ABL_NO_MFMA)
  mid();
  return;
#endif
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const bf16x8 b = frag_get<TB, 1>(fb[ks], 0);
    if constexpr (F16) {
      const f16x8 bh = __builtin_bit_cast(f16x8, b);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, __builtin_bit_cast(f16x8, frag_get<TA, 2>(fa[ks], 0)), c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, __builtin_bit_cast(f16x8, frag_get<TA, 2>(fa[ks], 1)), c1, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, frag_get<TA, 2>(fa[ks], 0), c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, frag_get<TA, 2>(fa[ks], 1), c1, 0, 0, 0);
    }
    if (ks == 0) mid();
  }
}

// Measured in rounds 2-3 and NOT part of this kernel any more (sources in the history up to commit 78acbbd, results in profiles/):
// a register-direct epilogue (permlane32_swap instead of the LDS passes: -4 % in the BERT-large step, r03_gemm_pp_reg_epilogue.md),
// a persistent tile walk whose operand stream crosses tile boundaries (-4.7 %, r03_gemm_pp_persistent.md), a ten-slot operand ring
// over all 160 KiB (-4 ... -12 %, r03_gemm_pp_ring10.md), a first-round stagger (+-, r03_gemm_pp_stagger.md), DMA requests in two
// bursts per K-tile and B fragments read a phase early (VAR 6 / 7, r03_gemm_experiments.md), DMA requests in front of the
// fragment reads / no s_setprio (VAR 3 / 2, r02_gemm_pp_variants.txt), one wave per SIMD with 128 x 128 wave tiles in HIP C++
// (tools/experiments/gemm_w4.hip, r03_gemm_experiments.md section 3).
// VAR 0: four thin phases per K-tile (8 MFMAs each; the encoder).  VAR 5 (impl 18; the search's score GEMM): "fat" phases - two per K-tile of 16 MFMAs each, (A0: B0, B1) and (A1: B1, B0), i.e.
// half the barriers per MFMA.  The fragment reads of a phase are retired (lgkmcnt) in FRONT of its first barrier, so that a
// half-tile may be requested one phase after its last read even by the group that runs a barrier ahead.  Back to back in
// tools/gemm_bench.py it is 3-9 % faster than VAR 0 on every shape (profiles/archive/r02_gemm_pp_fat.txt), inside the BERT-large
// training step 0.8 % SLOWER (same box, two alternating runs each: 3 741 vs 3 771 sequences/s) - under the package power limit
// a denser loop buys a lower clock, not time - so the encoder keeps VAR 0 and the long back-to-back launches of the search
// (+2 %) take VAR 5.  F16: IEEE-half operands.
// MULTI: up to four independent batched problems of one form in ONE launch (the weight gradients of a layer range: Wqkv, Wo, W1,
// W2 differ in shape, so they cannot be batch items of one problem) - the grid is the concatenation of the problems' flat
// (item, tile) ranges, a workgroup picks its problem from the kernel argument table
// The last partial round: when the tiles of all problems leave r <= 64 over whole rounds of the 256 CUs, those r tiles are
// not launched as one more (nearly empty) round of whole tiles but cut into s = 256 / r slices of the contraction each: r s
// workgroups write fp32 partial tiles to a workspace and a small second kernel adds the s slices in a fixed order
// (deterministic, no atomics).  split_first = first workgroup id of that region (= the grid size when nothing is cut).
struct MultiArgs {
  cocodr_gemm_args p[4];
  int tile_end[4];  // running totals of the problems' workgroup counts
  int split_first, split_s;
  float* split_ws;
};
constexpr int SPLIT_TILE = BM * 256;  // floats of one partial tile
// FILTER (the search, score_filter.h): the epilogue keeps the scores >= a per-row threshold instead of storing the tile
template <int NB, int TA, int TB, bool OUT_F32, int VAR = 5, bool F16 = false, bool MULTI = false, bool FILTER = false>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_pp_kernel(const typename std::conditional<MULTI, MultiArgs, cocodr_gemm_args>::type pa,
                                                              const int flags,
                                                              const typename std::conditional<FILTER, cocodr_score_filter, int>::type flt) {
#if defined(__HIP_DEVICE_COMPILE__)
  using S = Shape<NB>;
  static_assert(VAR == 0 || VAR == 5, "VAR 0: four thin phases per K-tile, VAR 5: two fat ones");
  const int flat = flags & 1;
  constexpr int BN = S::BN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;  // waves 0-3 (one per SIMD) form group 0, waves 4-7 group 1
  int mq = 0, mid = 0;  // MULTI: problem index and the workgroup's id inside that problem's range
  int split_chunk = -1;  // >= 0: this workgroup computes one contraction slice of a tile of the last partial round
  float* split_out = nullptr;
  if constexpr (MULTI) {
    if ((int)blockIdx.x < pa.split_first) {
      mid = xcd_remap(blockIdx.x, pa.split_first);
    } else {
      const int w = (int)blockIdx.x - pa.split_first;
      mid = pa.split_first + w / pa.split_s;
      split_chunk = w % pa.split_s;
      split_out = pa.split_ws + (size_t)w * SPLIT_TILE;
    }
    while (mq < 3 && mid >= pa.tile_end[mq]) ++mq;
    if (mq > 0) mid -= pa.tile_end[mq - 1];
  }
  const cocodr_gemm_args& p = [&]() -> const cocodr_gemm_args& {
    if constexpr (MULTI) return pa.p[mq];
    else return pa;
  }();
  const int ntn = p.N / BN, ntm = (p.M + BM - 1) / BM;
  // flat (batched launches): ONE grid axis over (batch item, tile), item-major, and the XCD remap over all of it - every
  // XCD walks a contiguous run of items' tiles, so the ~32 tiles its CUs hold at a time belong to one or two items and
  // form an 8 x 4 block of one item's output: 12 operand panel streams through the XCD's L2 for 32 tiles.  With the
  // remap per item (grid.y = item) an XCD held 4-6 tiles of each of 5-6 items at once, which share nothing: the grouped
  // weight gradients fetched 3.5x their operands from the fabric (profiles/archive/r02_gemm_pmc_large_200x128.json).
  int tile, z;
  if constexpr (MULTI) {
    const int per = ntm * ntn;
    z = mid / per;
    tile = mid - z * per;
  } else if (flat) {
    const int per = ntm * ntn;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    z = id / per;
    tile = id - z * per;
  } else {
    tile = xcd_remap(blockIdx.x, (int)gridDim.x);
    z = blockIdx.y;
  }
  int tm_, tn_;
  if (TA == 0) grouped_tile(tile, ntm, ntn, 4, tm_, tn_);
  else if (flat) grouped_tile(tile, ntm, ntn, 8, tm_, tn_);
  else { tm_ = tile / ntn; tn_ = tile % ntn; }
  const int m0 = tm_ * BM, n0 = tn_ * BN;
  if constexpr (FILTER) {
    if (flt.m_dev != nullptr && m0 >= *flt.m_dev - flt.m_base) return;  // workgroup-uniform, in front of every barrier
  }
  const uint16_t* A = p.A + (size_t)z * p.strideA;
  const uint16_t* B = p.B + (size_t)z * p.strideB;
  const uint32_t a_bytes = (uint32_t)((size_t)(TA ? p.K : p.M) * p.lda * 2);
  const uint32_t b_bytes = (uint32_t)((size_t)(TB ? p.K : p.N) * p.ldb * 2);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);

  // per-lane DMA source offsets of this wave's two 1-KiB pieces (chunks (2 wid + jj) * 64 + lane) of every half-tile type
  uint32_t off[S::NTYPE][2];
#pragma unroll
  for (int ty = 0; ty < S::NTYPE; ++ty)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int pch = (wid * 2 + jj) * 64 + lane;
      off[ty][jj] = type_is_a<NB>(ty) ? src_off<TA, NB, true>(pch, type_sub<NB>(ty), m0, p.lda)
                                      : src_off<TB, NB, false>(pch, type_sub<NB>(ty), n0, p.ldb);
    }
  const uint32_t stepa = TA ? (uint32_t)(BK * p.lda * 2) : (uint32_t)(BK * 2);
  const uint32_t stepb = TB ? (uint32_t)(BK * p.ldb * 2) : (uint32_t)(BK * 2);
  int nt = (p.K + BK - 1) / BK, t0 = 0;  // K-tiles of this workgroup: all of them, or one slice [t0, t0 + nt)
  if constexpr (MULTI) {
    if (split_chunk >= 0) {
      t0 = (int)((long long)split_chunk * nt / pa.split_s);
      nt = (int)((long long)(split_chunk + 1) * nt / pa.split_s) - t0;
    }
  }

  auto stage = [&](auto tyc, int t) {  // request half-tile `ty` of K-tile t
    constexpr int ty = decltype(tyc)::value;
#if defined(COCODR_ABL_NO_DMA)
    if (t > 0) return;
#endif
    char* dst = smem + (t & 1) * S::KT_BYTES + ty * HALF_BYTES + wid * 2048;
    if constexpr (type_is_a<NB>(ty)) {
      const uint32_t sb_ = (t + t0) * stepa;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (LDS_PTR(void))dst, 16, off[ty][0] + sb_, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (LDS_PTR(void))(dst + 1024), 16, off[ty][1] + sb_, 0, 0, 0);
    } else {
      const uint32_t sb_ = (t + t0) * stepb;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (LDS_PTR(void))dst, 16, off[ty][0] + sb_, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (LDS_PTR(void))(dst + 1024), 16, off[ty][1] + sb_, 0, 0, 0);
    }
  };
  f32x16 acc[4][NB];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const uint32_t lds_base = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;
  uint32_t adA[4], adB[4];
  frag_addrs<TA, 128, BK, 2>(wr * 64, lane, adA);
  frag_addrs<TB, 128, BK, 1>(wc * 32, lane, adB);

  // ---- prologue: the half-tiles 0 .. LOOK-1 of the request sequence (K-tile 0 and the first two of K-tile 1); phase 0 needs
  // A0 and B0 of K-tile 0, everything behind them may stay in flight
  static_for<0, S::NTYPE>([&](auto tyc) { stage(tyc, 0); });
  if (nt > 1) {
    stage(std::integral_constant<int, 0>{}, 1);
    stage(std::integral_constant<int, 1>{}, 1);
    if constexpr (VAR == 5) wait_vmcnt<6>();  // the first fat phase reads A0, B0 and B1
    else wait_vmcnt<8>();
  } else {
    if constexpr (VAR == 5) wait_vmcnt<2>();
    else wait_vmcnt<4>();
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0 from here on

#if defined(COCODR_ABL_NO_LDSREAD)
  FragSet<TA, 2> fa[4] = {};
  FragSet<TB, 1> fb0[4] = {}, fb1[4] = {};
#else
  FragSet<TA, 2> fa[4];
  FragSet<TB, 1> fb0[4], fb1[4];
#endif
  uint32_t curA[4], curB[4];
#if defined(COCODR_ABL_TIMELINE)  // per-workgroup stamps (100 MHz wall clock) into C2: [start, loop entry, loop exit, end]
  unsigned long long* tl = reinterpret_cast<unsigned long long*>(p.C2) + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8;  // (flat: y = 0)
  if (tid == 0) { tl[0] = wall_clock64(); tl[1] = wall_clock64(); }
#endif

  // Request sequence s = 4 t + j in the order A0 B0 B1 A1 of every K-tile; phase p = 4 t + j requests s = p + 6 (the buffer it
  // lands in was last read at phase p - 2 or earlier) and retires s = p + 2, which is first read at phase p + 1 or later: four
  // half-tiles (64 KiB per CU) stay in flight, five to six phases (~1.5 K-tiles) between request and first use.  With only
  // two half-tiles in flight the operand stream ran at ~50 GB/s per CU - the latency of a loaded L2 times the bytes in
  // flight - and bounded the whole loop (DMA-only ablation, profiles/archive/r02_gemm_pp_ablation.txt).
  // rem = K-tiles left including this one: the request exists while its K-tile does; the wait count shrinks with the queue.
  auto request = [&](auto jc, int t, int rem) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j < 2) { if (rem >= 2) stage(std::integral_constant<int, j + 2>{}, t + 1); }
    else { if (rem >= 3) stage(std::integral_constant<int, j - 2>{}, t + 2); }
  };
  auto wait_stage = [&](auto jc, int rem) {
    constexpr int j = decltype(jc)::value;
    if (rem >= 3) wait_vmcnt<8>();
    else if (rem == 2) wait_vmcnt<(j < 2 ? 8 : (j == 2 ? 6 : 4))>();
    else wait_vmcnt<(j == 0 ? 2 : 0)>();
  };

  // One K-tile.  STEADY: at least two more K-tiles follow (rem >= 3) - every request exists and every wait is vmcnt(8), so the
  // steady-state loop carries no scalar compare / branch at all (the rem-dependent forms cost 3-6 branches per load segment,
  // a fifth of its 256-cycle budget); the last two K-tiles run the general form.
  auto ktile = [&](auto steady_c, const int t, const int rem_in) {
    constexpr bool STEADY = decltype(steady_c)::value;
    const int rem = STEADY ? 3 : rem_in;
    const uint32_t kb = lds_base + (uint32_t)((t & 1) * S::KT_BYTES);
#pragma unroll
    for (int i = 0; i < 4; ++i) { curA[i] = adA[i] + kb; curB[i] = adB[i] + kb; }
    // one phase; RD: this phase's fragment reads, TY: the half-tile type requested for K-tile t + 1
    auto phase = [&](auto tyc, auto&& reads, const FragSet<TB, 1> (&fb)[4], f32x16& c0, f32x16& c1) {
      auto none = []() {};
      reads();
      request(tyc, t, rem);  // the DMA pieces go out in the load segment, behind the fragment reads
      wait_stage(tyc, rem);
      __builtin_amdgcn_s_barrier();
      wait_lgkmcnt<0>();
      __builtin_amdgcn_s_setprio(1);
      pp_mfma<TA, TB, F16>(fa, fb, c0, c1, none);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
    };
    if constexpr (NB == 2 && VAR == 5) {
      auto none = []() {};
      // X: (A0, B0), (A0, B1).  Requests B1(t+1); afterwards A1(t) - read by Y - must have landed: behind it in the queue are
      // A0, B0 and B1 of K-tile t + 1 (when that tile exists)
      pp_read_sub<TA, 2, 0 * HALF_BYTES>(curA, fa);
      pp_read_sub<TB, 1, 1 * HALF_BYTES>(curB, fb0);
      pp_read_sub<TB, 1, 2 * HALF_BYTES>(curB, fb1);
      if (rem >= 2) { stage(std::integral_constant<int, 2>{}, t + 1); wait_vmcnt<6>(); }
      else wait_vmcnt<0>();
      wait_lgkmcnt<0>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
      pp_mfma<TA, TB, F16>(fa, fb0, acc[0][0], acc[1][0], none);
      pp_mfma<TA, TB, F16>(fa, fb1, acc[0][1], acc[1][1], none);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      // Y: (A1, B1), (A1, B0).  Requests A1(t+1), A0(t+2), B0(t+2); afterwards A0, B0, B1 of K-tile t + 1 - read by the next X -
      // must have landed: behind B1(t+1) in the queue are exactly this phase's own requests
      pp_read_sub<TA, 2, 3 * HALF_BYTES>(curA, fa);
      if (rem >= 2) stage(std::integral_constant<int, 3>{}, t + 1);
      if (rem >= 3) {
        stage(std::integral_constant<int, 0>{}, t + 2);
        stage(std::integral_constant<int, 1>{}, t + 2);
        wait_vmcnt<6>();
      } else if (rem == 2) {
        wait_vmcnt<2>();
      } else {
        wait_vmcnt<0>();
      }
      wait_lgkmcnt<0>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
      pp_mfma<TA, TB, F16>(fa, fb1, acc[2][1], acc[3][1], none);
      pp_mfma<TA, TB, F16>(fa, fb0, acc[2][0], acc[3][0], none);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
    } else if constexpr (NB == 2) {
      phase(std::integral_constant<int, 0>{}, [&]() { pp_read_sub<TA, 2, 0 * HALF_BYTES>(curA, fa); pp_read_sub<TB, 1, 1 * HALF_BYTES>(curB, fb0); },
            fb0, acc[0][0], acc[1][0]);                                                                  // (A0, B0)
      phase(std::integral_constant<int, 1>{}, [&]() { pp_read_sub<TB, 1, 2 * HALF_BYTES>(curB, fb1); }, fb1, acc[0][1], acc[1][1]);  // (A0, B1)
      phase(std::integral_constant<int, 2>{}, [&]() { pp_read_sub<TA, 2, 3 * HALF_BYTES>(curA, fa); }, fb1, acc[2][1], acc[3][1]);   // (A1, B1)
      phase(std::integral_constant<int, 3>{}, [&]() {}, fb0, acc[2][0], acc[3][0]);                      // (A1, B0): all in registers
    }
  };
  if constexpr (VAR == 0 && NB == 2) {
    int t = 0;
    for (; t < nt - 2; ++t) ktile(std::true_type{}, t, 3);
    for (; t < nt; ++t) ktile(std::false_type{}, t, nt - t);
  } else {
    for (int t = 0; t < nt; ++t) ktile(std::false_type{}, t, nt - t);
  }
#if defined(COCODR_ABL_TIMELINE)
  if (tid == 0) tl[2] = wall_clock64();
#endif
  if (wr == 0) __builtin_amdgcn_s_barrier();  // group 0 catches up with group 1's last barrier
  if constexpr (MULTI) {
    if (split_chunk >= 0) {  // a contraction slice: the raw fp32 tile goes to the workspace, gemm_pp_split_finish adds the slices
#pragma unroll
      for (int ai = 0; ai < 4; ++ai)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int row = wr * 128 + ai * 32 + (lane & 31);
            const int col = wc * 32 * NB + b * 32 + 8 * rg + 4 * (lane >> 5);
            *reinterpret_cast<float4*>(split_out + row * BN + col) =
                make_float4(acc[ai][b][rg * 4 + 0], acc[ai][b][rg * 4 + 1], acc[ai][b][rg * 4 + 2], acc[ai][b][rg * 4 + 3]);
          }
      return;
    }
  }

  if constexpr (FILTER) {
    if (flt.mode == 0) {
      // ---- filter epilogue: the same two 128-row passes through LDS; a thread scans 8 chunks of 8 consecutive columns per pass
      // into a 64-bit hit mask (branch-free), then walks ITS hits - a wave runs as many trips as its busiest lane has hits (a
      // handful at ~1.6 % density), not one branch per accumulator register.  A hit takes its slot in the (row, column tile)
      // block from an LDS counter of the row.
      static_assert(NB == 2, "filter epilogue: 256-column tiles");
      constexpr int CLD = S::CT_LD, CPRW = BN / 8, NCH = 128 * CPRW / NTHREADS;
      static_assert(NCH == 8, "one byte of the hit mask per chunk");
      float* ct = reinterpret_cast<float*>(smem);
      float* thr_s = ct + 128 * CLD;
      int* rowcnt = reinterpret_cast<int*>(thr_s + 128);
      const int tn_g = flt.tile0 + tn_;
      __syncthreads();
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {
        const int gm_t = m0 + h * 128 + tid;  // (tid < 128)
        if (tid < 128) {
          thr_s[tid] = gm_t < p.M ? flt.thr[(size_t)gm_t * flt.thr_stride] : __builtin_inff();
          rowcnt[tid] = 0;
        }
        if (wr == h) {
#pragma unroll
          for (int ai = 0; ai < 4; ++ai)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
              for (int rg = 0; rg < 4; ++rg) {
                const int row = ai * 32 + (lane & 31);
                const int col = wc * 32 * NB + b * 32 + 8 * rg + 4 * (lane >> 5);
                *reinterpret_cast<float4*>(ct + row * CLD + col) =
                    make_float4(acc[ai][b][rg * 4 + 0], acc[ai][b][rg * 4 + 1], acc[ai][b][rg * 4 + 2], acc[ai][b][rg * 4 + 3]);
              }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        unsigned long long hits = 0;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int c = tid + i * NTHREADS;
          const int row = c / CPRW, c8 = (c % CPRW) << 3;
          const float t = thr_s[row];
          const float4 c0 = *reinterpret_cast<const float4*>(ct + row * CLD + c8);
          const float4 c1 = *reinterpret_cast<const float4*>(ct + row * CLD + c8 + 4);
          const int left = flt.n_valid - (flt.col0 + n0 + c8);  // columns of this chunk that exist
          unsigned m8 = (c0.x >= t ? 1u : 0u) | (c0.y >= t ? 2u : 0u) | (c0.z >= t ? 4u : 0u) | (c0.w >= t ? 8u : 0u) |
                        (c1.x >= t ? 16u : 0u) | (c1.y >= t ? 32u : 0u) | (c1.z >= t ? 64u : 0u) | (c1.w >= t ? 128u : 0u);
          if (left < 8) m8 &= left <= 0 ? 0u : ((1u << left) - 1u);
          hits |= (unsigned long long)m8 << (8 * i);
        }
        while (hits != 0) {
          const int bit = __builtin_ctzll(hits);
          hits &= hits - 1;
          const int c = tid + (bit >> 3) * NTHREADS;
          const int row = c / CPRW, cl = ((c % CPRW) << 3) + (bit & 7);
          const float v = ct[row * CLD + cl];
          const int pos = atomicAdd(&rowcnt[row], 1);
          if (pos + 1 < flt.capt)
            flt.cand[((size_t)(m0 + h * 128 + row) * flt.ntn_total + tn_g) * flt.capt + 1 + pos] =
                make_uint2(__float_as_uint(v), (uint32_t)(flt.col0 + n0 + cl));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < 128 && gm_t < p.M) flt.cand[((size_t)gm_t * flt.ntn_total + tn_g) * flt.capt] = make_uint2((uint32_t)rowcnt[tid], 0u);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      return;
    }
  }

  // ---- epilogue (gemm.hip's, for this geometry): two 128-row passes of the fp32 tile through LDS, row-major 16-B stores
  const float* __restrict__ bias = p.bias ? p.bias + (size_t)z * p.strideBias : nullptr;
  const uint16_t* __restrict__ R_ = p.R ? p.R + (size_t)z * p.strideR : nullptr;
  constexpr int CPRW = BN / 8;                 // 8-column chunks per output row
  constexpr int RP = 128;                      // rows per pass = one wave row
  constexpr int NCH = RP * CPRW / NTHREADS;    // chunks per thread and pass (4 / 8); CPRW divides NTHREADS: fixed columns
  const bool need_r = R_ != nullptr && (p.epi == COCODR_EPI_ADD || p.epi == COCODR_EPI_DGELU);
  uint4 rcur[NCH];
  auto fetch_r = [&](int h) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * NTHREADS;
      const int gm = m0 + h * RP + c / CPRW;
      rcur[i] = make_uint4(0, 0, 0, 0);
      if (need_r && gm < p.M) rcur[i] = *reinterpret_cast<const uint4*>(R_ + (size_t)gm * p.ldr + n0 + ((c % CPRW) << 3));
    }
  };
  fetch_r(0);
  float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr) {
    const float4 b0 = *reinterpret_cast<const float4*>(bias + n0 + ((tid % CPRW) << 3));
    const float4 b1 = *reinterpret_cast<const float4*>(bias + n0 + ((tid % CPRW) << 3) + 4);
    bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
  }
  __syncthreads();
  float* ct = reinterpret_cast<float*>(smem);
  constexpr int CLD = S::CT_LD;
  const bool do_colsum = p.colsum_partial != nullptr;
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    if (wr == h) {
#pragma unroll
      for (int ai = 0; ai < 4; ++ai)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int row = ai * 32 + (lane & 31);
            const int col = wc * 32 * NB + b * 32 + 8 * rg + 4 * (lane >> 5);
            *reinterpret_cast<float4*>(ct + row * CLD + col) =
                make_float4(acc[ai][b][rg * 4 + 0], acc[ai][b][rg * 4 + 1], acc[ai][b][rg * 4 + 2], acc[ai][b][rg * 4 + 3]);
          }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * NTHREADS;
      const int row = c / CPRW, c8 = (c % CPRW) << 3;
      const int gm = m0 + h * RP + row;
      const int gn = n0 + c8;
      if (gm < p.M) {
        float v[8];
        const float4 c0 = *reinterpret_cast<const float4*>(ct + row * CLD + c8);
        const float4 c1 = *reinterpret_cast<const float4*>(ct + row * CLD + c8 + 4);
        v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w; v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
        epilogue_store8<OUT_F32, true, true>(p, z, bias, R_, gm, gn, v, rcur[i], bias8);
        if (do_colsum) {
#pragma unroll
          for (int j = 0; j < 8; ++j) csum[j] += v[j];
        }
      }
    }
    if (h == 0) {
      fetch_r(1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  if (do_colsum) {  // workgroup-uniform: lanes that differ by a multiple of CPRW hold the same 8 columns
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (CPRW <= 16) csum[j] += __shfl_xor(csum[j], 16, 64);
      csum[j] += __shfl_xor(csum[j], 32, 64);
    }
    float* cred = ct + RP * CLD;
    if (lane < CPRW) {
#pragma unroll
      for (int j = 0; j < 8; ++j) cred[wid * BN + lane * 8 + j] = csum[j];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tid < BN) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += cred[w * BN + tid];
      p.colsum_partial[(size_t)tm_ * p.N + n0 + tid] = t;
    }
  }
#if defined(COCODR_ABL_TIMELINE)
  if (tid == 0) tl[3] = wall_clock64();
#endif
#endif
}

template <int NB, int TA, int TB, int VAR = 5, bool F16 = false>
void launch_form(const cocodr_gemm_args& a, hipStream_t st) {
  using S = Shape<NB>;
  const int ntm = (a.M + BM - 1) / BM, ntn = a.N / S::BN;
  const int flat = a.batch > 1 ? 1 : 0;  // batched launches walk (item, tile) in one XCD-remapped grid axis (see the kernel)
  dim3 grid(flat ? ntm * ntn * a.batch : ntm * ntn, flat ? 1 : a.batch);
  static cocodr_lds_once attr_done;
  if (attr_done.pending()) {
    hipFuncSetAttribute((const void*)gemm_pp_kernel<NB, TA, TB, true, VAR, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)gemm_pp_kernel<NB, TA, TB, false, VAR, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.done();
  }
  if (a.out_f32)
    hipLaunchKernelGGL((gemm_pp_kernel<NB, TA, TB, true, VAR, F16>), grid, dim3(NTHREADS), S::LDS_BYTES, st, a, flat, 0);
  else
    hipLaunchKernelGGL((gemm_pp_kernel<NB, TA, TB, false, VAR, F16>), grid, dim3(NTHREADS), S::LDS_BYTES, st, a, flat, 0);
}

}  // namespace cocodr_gemm_pp

namespace cocodr_gemm_pp {
// C tile = sum of the s contraction slices (fixed order), for the r tiles of the cut last round; grid = (r, 64), 256 threads
__global__ __launch_bounds__(256) void gemm_pp_split_finish(const MultiArgs ma, int r) {
  const int lt = blockIdx.x;
  int q = 0, id = ma.split_first + lt;
  while (q < 3 && id >= ma.tile_end[q]) ++q;
  if (q > 0) id -= ma.tile_end[q - 1];
  const cocodr_gemm_args& p = ma.p[q];
  const int ntn = p.N / 256, ntm = (p.M + BM - 1) / BM, per = ntm * ntn;
  const int z = id / per, tile = id - z * per;
  int tm_, tn_;
  cocodr_gemm_v2::grouped_tile(tile, ntm, ntn, 8, tm_, tn_);  // the mapping of the flat TN form in gemm_pp_kernel
  const int e = (blockIdx.y * 256 + threadIdx.x) * 4, row = e / 256, col = e % 256;
  const float* src = ma.split_ws + (size_t)lt * ma.split_s * SPLIT_TILE + e;
  float4 acc = *reinterpret_cast<const float4*>(src);
  for (int c = 1; c < ma.split_s; ++c) {
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)c * SPLIT_TILE);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const int gm = tm_ * BM + row, gn = tn_ * 256 + col;
  if (gm < p.M) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (size_t)z * p.strideC + (size_t)gm * p.ldc + gn) = acc;
}

// The same for ONE problem of any form with its epilogue (bias / GELU + GELU' / residual (+ dropout) / x GELU', bf16 or fp32 result,
// optional per-row-panel column sums): grid = (r tiles, 8 strips of 32 columns), 256 threads = 64 row lanes x 4 chunks of 8 columns;
// a thread walks rows lane, lane + 64, ... of its chunk, so a wave reads 16 rows x 128 contiguous bytes per slice.
template <bool OUT_F32, int TA>
__global__ __launch_bounds__(256) void gemm_pp_split_finish_epi(const MultiArgs ma) {
  __shared__ float cred[4][8][64];
  const cocodr_gemm_args& p = ma.p[0];
  const int lt = blockIdx.x, id = ma.split_first + lt;
  const int ntn = p.N / 256, ntm = (p.M + BM - 1) / BM;
  int tm_, tn_;
  if (TA == 0) cocodr_gemm_v2::grouped_tile(id, ntm, ntn, 4, tm_, tn_);
  else cocodr_gemm_v2::grouped_tile(id, ntm, ntn, 8, tm_, tn_);
  const int tid = threadIdx.x, ch = tid & 3, rl = tid >> 2;
  const int col = blockIdx.y * 32 + ch * 8, gn = tn_ * 256 + col;
  const float* __restrict__ bias = p.bias;
  const uint16_t* __restrict__ R_ = p.R;
  const float* src = ma.split_ws + (size_t)lt * ma.split_s * SPLIT_TILE + col;
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int row = rl; row < BM; row += 64) {
    const int gm = tm_ * BM + row;
    if (gm >= p.M) break;
    float v[8];
    const float4 a0 = *reinterpret_cast<const float4*>(src + (size_t)row * 256);
    const float4 a1 = *reinterpret_cast<const float4*>(src + (size_t)row * 256 + 4);
    v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
    for (int c = 1; c < ma.split_s; ++c) {
      const float4 b0 = *reinterpret_cast<const float4*>(src + (size_t)c * SPLIT_TILE + (size_t)row * 256);
      const float4 b1 = *reinterpret_cast<const float4*>(src + (size_t)c * SPLIT_TILE + (size_t)row * 256 + 4);
      v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    epilogue_store8<OUT_F32>(p, 0, bias, R_, gm, gn, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) csum[j] += v[j];
  }
  if (p.colsum_partial != nullptr) {  // the tile's column sums (workgroup-uniform branch), one row per row panel as in the whole-tile epilogue
#pragma unroll
    for (int j = 0; j < 8; ++j) cred[ch][j][rl] = csum[j];
    __syncthreads();
    if (tid < 32) {
      float t = 0.f;
      for (int i = 0; i < 64; ++i) t += cred[tid >> 3][tid & 7][i];
      p.colsum_partial[(size_t)tm_ * p.N + tn_ * 256 + blockIdx.y * 32 + tid] = t;
    }
  }
}
}  // namespace cocodr_gemm_pp

// floats of workspace the cut last round of ANY merged launch may need (see MultiArgs)
size_t cocodr_gemm_pp_multi_ws_floats() { return (size_t)256 * cocodr_gemm_pp::SPLIT_TILE; }

namespace {
// compute units the cut is planned for: the device's (a multiple of 8), or 256 (MI355X) where no device can be asked -
// layout functions run on the host alone
int multi_n_cu() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
      n_cu = 256;
    if (n_cu % 8 != 0) n_cu = -1;
  }
  return n_cu;
}
// the cut of the last partial round: r tiles in s contraction slices each (s = 0: no cut)
void multi_split_plan(const cocodr_gemm_args* a, int n, int& total, int& r, int& s) {
  using namespace cocodr_gemm_pp;
  total = 0;
  for (int q = 0; q < n; ++q) total += ((a[q].M + BM - 1) / BM) * (a[q].N / Shape<2>::BN) * (a[q].batch > 0 ? a[q].batch : 1);
  const int n_cu = multi_n_cu();
  r = n_cu > 0 ? total % n_cu : 0;
  const int nt_min = (a[0].K + BK - 1) / BK;
  s = r > 0 ? n_cu / r : 0;
  if (s > nt_min) s = nt_min;
  if (!(r > 0 && total > n_cu && s >= 4 && (size_t)r * s <= 256)) s = 0;
}
}  // namespace

// floats of workspace THIS merged launch needs for its cut last round (0: it runs whole tiles only)
size_t cocodr_gemm_pp_multi_ws_floats_for(const cocodr_gemm_args* a, int n) {
  int total, r, s;
  multi_split_plan(a, n, total, r, s);
  return s ? (size_t)r * s * cocodr_gemm_pp::SPLIT_TILE : 0;
}

// n <= 4 batched TN problems with fp32 results (validated by cocodr_gemm_multi) as one launch; ws: optional workspace of
// ws_floats >= cocodr_gemm_pp_multi_ws_floats_for(a, n) floats for the cut last round (NULL / smaller: whole tiles only)
void cocodr_gemm_pp_launch_multi(const cocodr_gemm_args* a, int n, float* ws, size_t ws_floats, hipStream_t st) {
  using namespace cocodr_gemm_pp;
  MultiArgs ma;
  int total = 0;
  for (int q = 0; q < 4; ++q) {
    ma.p[q] = a[q < n ? q : n - 1];
    if (q < n) total += ((a[q].M + BM - 1) / BM) * (a[q].N / Shape<2>::BN) * (a[q].batch > 0 ? a[q].batch : 1);
    ma.tile_end[q] = total;
  }
  static const bool nosplit = getenv("COCODR_GEMM_NOSPLIT") != nullptr;  // A/B switch
  int r, s, total2;
  multi_split_plan(a, n, total2, r, s);
  const bool split = ws != nullptr && !nosplit && s > 0 && ws_floats >= (size_t)r * s * SPLIT_TILE;
  ma.split_first = split ? total - r : total;
  ma.split_s = split ? s : 1;
  ma.split_ws = ws;
  // (two fat phases per K-tile for the merged launch, gemm_pp_kernel<2, 1, 1, true, 5, false, true>: measured +0.3 % on the
  //  BERT-base step, -0.3 ... -0.5 % on the BERT-large ones; not kept)
  auto kern = gemm_pp_kernel<2, 1, 1, true, 0, false, true>;
  static cocodr_lds_once attr_done;
  if (attr_done.pending()) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.done();
  }
  hipLaunchKernelGGL(kern, dim3(split ? total - r + r * s : total), dim3(NTHREADS), Shape<2>::LDS_BYTES, st, ma, 1, 0);
  if (split) hipLaunchKernelGGL(gemm_pp_split_finish, dim3(r, SPLIT_TILE / 4 / 256), dim3(256), 0, st, ma, r);
}

// ---- ONE forward / dgrad / weight-gradient problem (batch == 1) whose tiles leave a partial last round of the compute units: the
// r tiles of that round run as r x s contraction slices (see MultiArgs) and gemm_pp_split_finish_epi adds them and applies the
// epilogue.  Keeps arbitrary row counts (packed batches: T changes with every batch) on this pipeline: 280 tiles are 256 whole
// tiles + 24 tiles x 4 slices instead of two rounds, the second one with 9 % of the CUs at work.
size_t cocodr_gemm_pp_split_ws_floats() { return (size_t)256 * cocodr_gemm_pp::SPLIT_TILE; }
// the cut this problem would get: r tiles in s slices (s = 0: none)
void cocodr_gemm_pp_split_plan(const cocodr_gemm_args& a, int& total, int& r, int& s) {
  using namespace cocodr_gemm_pp;
  total = ((a.M + BM - 1) / BM) * (a.N / Shape<2>::BN);
  const int n_cu = multi_n_cu();
  const int nkt = (a.K + BK - 1) / BK;
  r = s = 0;
  static const int off = getenv("COCODR_GEMM_NOTAIL") ? 1 : 0;  // A/B switch
  if (off || n_cu <= 0 || a.batch > 1 || a.split_ws == nullptr || a.ab_f16 || a.drop.threshold >= 65536) return;
  r = total % n_cu;
  if (total <= n_cu || r == 0 || r > n_cu / 2) { r = 0; return; }
  s = n_cu / r;
  if (s > 8) s = 8;                // (fp32 partial tiles: 256 KB each way per slice)
  if (s > nkt / 2) s = nkt / 2;    // at least two K-tiles per slice
  if (s < 2 || (size_t)r * s * SPLIT_TILE > a.split_ws_floats) { r = s = 0; }
}
template <int TA, int TB>
static void launch_split(const cocodr_gemm_args& a, int total, int r, int s, hipStream_t st) {
  using namespace cocodr_gemm_pp;
  MultiArgs ma;
  for (int q = 0; q < 4; ++q) { ma.p[q] = a; ma.tile_end[q] = total; }
  ma.split_first = total - r;
  ma.split_s = s;
  ma.split_ws = a.split_ws;
  static cocodr_lds_once attr_done;
  if (attr_done.pending()) {
    hipFuncSetAttribute((const void*)gemm_pp_kernel<2, TA, TB, true, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)gemm_pp_kernel<2, TA, TB, false, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.done();
  }
  const dim3 grid(total - r + r * s);
  if (a.out_f32) {
    hipLaunchKernelGGL((gemm_pp_kernel<2, TA, TB, true, 0, false, true>), grid, dim3(NTHREADS), Shape<2>::LDS_BYTES, st, ma, 1, 0);
    hipLaunchKernelGGL((gemm_pp_split_finish_epi<true, TA>), dim3(r, 8), dim3(256), 0, st, ma);
  } else {
    hipLaunchKernelGGL((gemm_pp_kernel<2, TA, TB, false, 0, false, true>), grid, dim3(NTHREADS), Shape<2>::LDS_BYTES, st, ma, 1, 0);
    hipLaunchKernelGGL((gemm_pp_split_finish_epi<false, TA>), dim3(r, 8), dim3(256), 0, st, ma);
  }
}
bool cocodr_gemm_pp_launch_split(const cocodr_gemm_args& a, hipStream_t st) {
  int total, r, s;
  cocodr_gemm_pp_split_plan(a, total, r, s);
  if (s == 0) return false;
  if (!a.trans_a && !a.trans_b) launch_split<0, 0>(a, total, r, s, st);
  else if (!a.trans_a && a.trans_b) launch_split<0, 1>(a, total, r, s, st);
  else launch_split<1, 1>(a, total, r, s, st);
  return true;
}

// nb = 2: four thin phases per K-tile (the encoder's default), 105: two fat phases per K-tile (impl 18), 104: IEEE-half operands
// with fat phases (the search's split-precision score GEMM); the caller has validated the arguments (cocodr_gemm)
template <int VAR>
static void launch_any(const cocodr_gemm_args& a, hipStream_t st) {
  using namespace cocodr_gemm_pp;
  if (!a.trans_a && !a.trans_b) launch_form<2, 0, 0, VAR>(a, st);
  else if (!a.trans_a && a.trans_b) launch_form<2, 0, 1, VAR>(a, st);
  else launch_form<2, 1, 1, VAR>(a, st);
}
void cocodr_gemm_pp_launch_filter(const cocodr_gemm_args& a, const cocodr_score_filter& f, hipStream_t st) {
  using namespace cocodr_gemm_pp;
  using S = Shape<2>;
  auto kern = gemm_pp_kernel<2, 0, 0, true, 5, true, false, true>;
  static cocodr_lds_once attr_done;
  if (attr_done.pending()) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.done();
  }
  const int ntm = (a.M + BM - 1) / BM, ntn = a.N / S::BN;
  hipLaunchKernelGGL(kern, dim3(ntm * ntn, 1), dim3(NTHREADS), S::LDS_BYTES, st, a, 0, f);
}
void cocodr_gemm_pp_launch(const cocodr_gemm_args& a, int nb, hipStream_t st) {
  if (nb == 104) cocodr_gemm_pp::launch_form<2, 0, 0, 5, true>(a, st);
  else if (nb == 105) launch_any<5>(a, st);
  else launch_any<0>(a, st);
}
