// "Ping-pong" bf16 MFMA GEMM for gfx950: 256 x (128 NB) x 64 tile, eight waves in two groups that alternate, on every
// SIMD, between an MFMA segment and a load segment.
//
// Why a second pipeline: in gemm.hip's one-barrier-per-K-step loop all eight waves of a workgroup do the same thing at the
// same time, so the MFMA pipe, the LDS and the address unit that feeds the LDS-DMA are each ~50 % busy and their times add
// (profiles/r01f_*).  Here the waves with wr = 0 (one per SIMD) and the waves with wr = 1 (their SIMD partners) run the same
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

// ---- register-direct epilogue (bf16 results): the accumulators never pass through LDS.
// With the operands swapped a lane holds, per 32 x 32 block, ONE output row (lane & 31) and the 4-column groups
// 8 rg + 4 (lane >> 5) .. + 3, rg = 0..3.  Bias / GELU / residual / dropout are applied there in fp32; after the bf16 pack one
// v_permlane32_swap per dword hands the upper half-wave's group rg to the lower lane and the lower half-wave's group rg + 1 to
// the upper lane, so every lane ends with 8 consecutive columns = one 16-B store (lanes 0-31: columns 16 k .. + 7,
// lanes 32-63: 16 k + 8 .. + 15 of the same row).  No barrier, no LDS pass, no second wave-row pass.
// The residual / GELU' operand is read in the same layout (8 B per lane and group), one 32-row block ahead.
// MEASURED AND NOT ADOPTED (round 3, profiles/r03_gemm_pp_reg_epilogue.md; opt-in with COCODR_PP_REGEPI=1): in the BERT-large
// training step it is 4 % SLOWER at 200 sequences and 1-2 % slower at 64 (the residual / GELU' reads in this layout expose their
// latency; 32-B row segments per store instruction); with the stores compiled out the kernels lose only 4-7 % with EITHER
// epilogue - the "5.3 us per tile" of the LDS-staged form is the memory system taking the tile's 128 KB, not the LDS passes.
__device__ __forceinline__ void swap_halves(uint32_t& a, uint32_t& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
template <int NB>
__device__ __forceinline__ void pp_reg_epilogue(const cocodr_gemm_args& p, const int z, const f32x16 (&acc)[4][NB], const int m0,
                                                const int n0, const int wr, const int wc, const int lane) {
  const int rl = lane & 31, hh = lane >> 5;
  const int colw = n0 + wc * 32 * NB + 4 * hh;  // this lane's first column (block 0, group 0)
  const float* __restrict__ bias = p.bias ? p.bias + (size_t)z * p.strideBias : nullptr;
  const uint16_t* __restrict__ R_ = p.R ? p.R + (size_t)z * p.strideR : nullptr;
  uint16_t* __restrict__ C = reinterpret_cast<uint16_t*>(p.C) + (size_t)z * p.strideC;
  uint16_t* __restrict__ C2 = p.C2 ? p.C2 + (size_t)z * p.strideC : nullptr;
  const int epi = p.epi;
  const bool need_r = R_ != nullptr && (epi == COCODR_EPI_ADD || epi == COCODR_EPI_DGELU);
  float4 bv[NB][4];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
      bv[b][rg] = bias ? *reinterpret_cast<const float4*>(bias + colw + b * 32 + 8 * rg) : make_float4(0.f, 0.f, 0.f, 0.f);
  uint2 rr[2][NB][4];
  auto fetch_r = [&](int ai, uint2 (&dst)[NB][4]) {
    const int gm = m0 + wr * 128 + ai * 32 + rl;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        dst[b][rg] = make_uint2(0, 0);
        if (need_r && gm < p.M) dst[b][rg] = *reinterpret_cast<const uint2*>(R_ + (size_t)gm * p.ldr + colw + b * 32 + 8 * rg);
      }
  };
  fetch_r(0, rr[0]);
#pragma unroll
  for (int ai = 0; ai < 4; ++ai) {
    if (ai + 1 < 4) fetch_r(ai + 1, rr[(ai + 1) & 1]);
    const int gm = m0 + wr * 128 + ai * 32 + rl;
    const bool ok = gm < p.M;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        uint32_t w[2][2], w2[2][2];  // [group 2k / 2k + 1][dword] packed bf16 pairs of the result (and of GELU')
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int rg = 2 * k + g;
          float v[4] = {acc[ai][b][rg * 4 + 0] + bv[b][rg].x, acc[ai][b][rg * 4 + 1] + bv[b][rg].y,
                        acc[ai][b][rg * 4 + 2] + bv[b][rg].z, acc[ai][b][rg * 4 + 3] + bv[b][rg].w};
          if (epi == COCODR_EPI_GELU && C2 == nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
          } else if (epi == COCODR_EPI_GELU) {
            float gp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) gelu_erf_both(v[j], v[j], gp[j]);
            w2[g][0] = pack2bf(gp[0], gp[1]);
            w2[g][1] = pack2bf(gp[2], gp[3]);
          } else if (epi == COCODR_EPI_ADD) {
            if (p.drop.threshold) drop_apply<4>(v, (uint64_t)gm * p.N + (colw + b * 32 + 8 * rg), p.drop);
            float r[4];
            unpack4(rr[ai & 1][b][rg], r);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += r[j];
          } else if (epi == COCODR_EPI_DGELU) {
            float r[4];
            unpack4(rr[ai & 1][b][rg], r);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= r[j];
          }
          w[g][0] = pack2bf(v[0], v[1]);
          w[g][1] = pack2bf(v[2], v[3]);
        }
        swap_halves(w[0][0], w[1][0]);
        swap_halves(w[0][1], w[1][1]);
        const size_t o = (size_t)gm * p.ldc + (n0 + wc * 32 * NB + b * 32 + 16 * k + 8 * hh);
#if defined(COCODR_ABL_EPI_NOSTORE)  // ablation: everything but the global stores
        asm volatile("" ::"v"(w[0][0]), "v"(w[0][1]), "v"(w[1][0]), "v"(w[1][1]), "v"(o));
        if (epi == COCODR_EPI_GELU && C2 != nullptr) asm volatile("" ::"v"(w2[0][0]), "v"(w2[0][1]), "v"(w2[1][0]), "v"(w2[1][1]));
#else
        if (ok) *reinterpret_cast<uint4*>(C + o) = make_uint4(w[0][0], w[0][1], w[1][0], w[1][1]);
        if (epi == COCODR_EPI_GELU && C2 != nullptr) {
          swap_halves(w2[0][0], w2[1][0]);
          swap_halves(w2[0][1], w2[1][1]);
          if (ok) *reinterpret_cast<uint4*>(C2 + o) = make_uint4(w2[0][0], w2[0][1], w2[1][0], w2[1][1]);
        }
#endif
      }
  }
}

// ---- epilogue of the persistent walk: four 64-row passes of the fp32 tile through the TOP 64 KiB of the LDS ([96 K, 160 K): the
// ring's buffer-1 slots B1 / A1 plus the 32 KiB behind the ring), so that the next tile's K-tile 0 (buffer 0) and the first
// half of its K-tile 1 (buffer-1 slots A0 / B0) stay in flight underneath.  64 rows x 256 floats fill the region exactly: no
// padding, the 16-B blocks of a row are XOR-swizzled with row & 15 instead (writes: 8 consecutive rows per LDS cycle hit 8
// different blocks; reads: a row's 32 lanes cover all 64 banks).  Row-major 16-B global accesses as in the two-pass epilogue.
// Every LDS access is inline assembly: hipcc orders the LDS accesses it can see behind ALL outstanding LDS-DMA (vmcnt(0)).
__device__ __forceinline__ void asm_ds_write_b128(uint32_t addr, float a, float b, float c, float d) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = {a, b, c, d};
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void pp_lds4_epilogue(const cocodr_gemm_args& p, const int z, const f32x16 (&acc)[4][2], const int m0, const int n0,
                                                 const int wr, const int wc, const int tid_in, const int lane_in, const uint32_t lds_base) {
  constexpr int NCH = 4;  // 64 rows x 32 chunks of 8 columns over 512 threads
  // opaque copies: every address below is a function of the thread id alone, i.e. invariant over the caller's tile loop - hoisted
  // out of it, the ~20 of them are spilled over the main loop and reloaded per tile; a few integer operations per tile are cheaper
  int tid = tid_in, lane = lane_in;
  asm volatile("" : "+v"(tid), "+v"(lane));
  const float* __restrict__ bias = p.bias ? p.bias + (size_t)z * p.strideBias : nullptr;
  const uint16_t* __restrict__ R_ = p.R ? p.R + (size_t)z * p.strideR : nullptr;
  const bool need_r = R_ != nullptr && (p.epi == COCODR_EPI_ADD || p.epi == COCODR_EPI_DGELU);
  const int c8 = (tid & 31) << 3;  // this thread's 8 columns (512 % 32 == 0: the same in every chunk)
  const int gn = n0 + c8;
  uint4 rcur[NCH], rnext[NCH];
  auto fetch_r = [&](int h, uint4 (&dst)[NCH]) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int gm = m0 + h * 64 + ((tid + i * NTHREADS) >> 5);
      dst[i] = make_uint4(0, 0, 0, 0);
      if (need_r && gm < p.M) dst[i] = *reinterpret_cast<const uint4*>(R_ + (size_t)gm * p.ldr + gn);
    }
  };
  fetch_r(0, rcur);
  float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr) {
    const float4 b0 = *reinterpret_cast<const float4*>(bias + gn);
    const float4 b1 = *reinterpret_cast<const float4*>(bias + gn + 4);
    bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
  }
  const uint32_t ct = lds_base + 96u * 1024u;
  const int hh = lane >> 5, rl = lane & 31;
  static_for<0, 4>([&](auto hc) {
    constexpr int h = decltype(hc)::value;
    if constexpr (h < 3) fetch_r(h + 1, rnext);
    if (wr == (h >> 1)) {
#pragma unroll
      for (int ai2 = 0; ai2 < 2; ++ai2)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            constexpr int dummy = 0; (void)dummy;
            const int row = ai2 * 32 + rl;
            const int blk = wc * 16 + b * 8 + 2 * rg + hh;  // 16-B block of columns wc 64 + b 32 + 8 rg + 4 hh
            const f32x16& a = acc[2 * (h & 1) + ai2][b];
            asm_ds_write_b128(ct + (uint32_t)(row * 1024 + ((blk ^ (row & 15)) << 4)), a[rg * 4 + 0], a[rg * 4 + 1], a[rg * 4 + 2], a[rg * 4 + 3]);
          }
    }
    wait_lgkmcnt<0>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i0 = 0; i0 < NCH; i0 += 2) {  // two chunks at a time (register budget)
      v4i lo[2], hi[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = (tid + (i0 + i) * NTHREADS) >> 5;
        const int blk = (tid & 31) << 1;
        const uint32_t rb = ct + (uint32_t)(row * 1024);
        asm_ds_read_b128<0>(lo[i], rb + (uint32_t)((blk ^ (row & 15)) << 4));
        asm_ds_read_b128<0>(hi[i], rb + (uint32_t)(((blk + 1) ^ (row & 15)) << 4));
      }
      wait_lgkmcnt<0>();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int gm = m0 + h * 64 + ((tid + (i0 + i) * NTHREADS) >> 5);
        if (gm < p.M) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] = __int_as_float(lo[i][j]); v[4 + j] = __int_as_float(hi[i][j]); }
          epilogue_store8<false, true, true>(p, z, bias, R_, gm, gn, v, rcur[i0 + i], bias8);
        }
      }
    }
    if constexpr (h < 3) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) rcur[i] = rnext[i];
      __builtin_amdgcn_s_barrier();  // the next pass overwrites what this one read (the reads have returned: lgkmcnt(0) above)
    }
  });
}

// VAR (experiment builds, -DCOCODR_PP_VARIANTS, tools/gemm_bench.py --impls 13,15,16): 0 = DMA pieces requested in the load
// segment behind the fragment reads; 2 = as 0 without s_setprio (-15 %); 3 = requested in front of the fragment reads (=).
// Requesting them inside the MFMA segment instead cost 10-14 % (profiles/r02_gemm_pp_variants.txt).
// VAR 5 (impl 18; the search's score GEMM): "fat" phases - two per K-tile of 16 MFMAs each, (A0: B0, B1) and (A1: B1, B0), i.e.
// half the barriers per MFMA.  The fragment reads of a phase are retired (lgkmcnt) in FRONT of its first barrier, so that a
// half-tile may be requested one phase after its last read even by the group that runs a barrier ahead.  Back to back in
// tools/gemm_bench.py it is 3-9 % faster than VAR 0 on every shape (profiles/r02_gemm_pp_fat.txt), inside the BERT-large
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
// PERSIST (forward / dgrad forms with bf16 results and no fused column sums; opt-in, see launch_form): a grid of at most one
// workgroup per CU walks the tiles, and the operand stream never stops at a tile boundary - the last two K-tiles of a tile
// request the first 1.5 K-tiles of the workgroup's NEXT tile into the same ring slots, in the steady-state order and with the
// steady-state waits, the register-direct epilogue (no LDS) runs with those requests in flight, and the next tile starts on
// landed operands.  What this hides is the tile's fixed cost: workgroup launch, prologue latency, and the idle loaders under
// the epilogue (profiles/r03_gemm_vs_library.md).  The epilogue's stores sit in the vmcnt queue between the prefetched pieces
// and the next tile's first requests: K-tile 0 of a continued tile allows for them in its counted waits (gfx9 retires vector
// memory loads and stores in issue order).  Needs an even number of K-tiles (the ring parity carries over).
// MEASURED AND NOT ADOPTED (round 3, profiles/r03_gemm_pp_persistent.md; experiment builds with -DCOCODR_PP_PERSIST_BUILD, then
// COCODR_PP_PERSIST=1 / 2): bit-identical on every form incl. ragged row counts; inside the BERT-large step the 1200-tile QKV
// GEMM gains 3 % (162 -> 157 us) but the 400-tile long-K forms (two tiles per workgroup on 200 CUs) lose 14-17 % and the step
// 4.7 %.  The tile's fixed cost turned out to be ~3.6 us of 32 (not the 8 assumed): the K-loop itself, 1.8 us per K-tile against
// 1.0 of MFMA time, is where the library's hand-scheduled kernel (1.5) is ahead.
// NSLOT = 10 (VAR 0, NB = 2): the half-tile ring takes the whole 160 KiB of the CU's LDS - ten 16-KiB slots, half-tile s = 4 t + j
// in slot s % 10 - instead of two K-tile buffers (eight slots).  Phase (t, j) then requests half-tile j of K-tile t + 2 (the
// slot that half-tile (t, j) - 2 left in the previous phase), six half-tiles = 96 KiB per CU stay in flight instead of four,
// and a request has two K-tiles instead of one and a half to land.  The loop's K-tile time is the latency of the operand
// stream divided by its look-ahead (profiles/r03_load_rate_probe.txt: bytes in flight x 1 / latency), not the MFMA time - that
// was the hypothesis.  MEASURED AND NOT ADOPTED (round 3, profiles/r03_gemm_pp_ring10.md; experiment builds with
// -DCOCODR_PP_RING10_BUILD, then COCODR_PP_RING=10): bit-identical on every form and K-tile count, and 4-12 % SLOWER everywhere
// (8192^3: 1263 -> 1182 TFLOP/s; the BERT-large step at 200 sequences -5 %).  Half again as many bytes in flight buy nothing:
// the K-loop is not waiting for the operand stream's latency.
template <int NB, int TA, int TB, bool OUT_F32, int VAR = 5, bool F16 = false, bool MULTI = false, int PERSIST = 0, int NSLOT = 8>  // PERSIST: 1 = LDS epilogue, 2 = register epilogue
__global__ __launch_bounds__(NTHREADS, 2) void gemm_pp_kernel(const typename std::conditional<MULTI, MultiArgs, cocodr_gemm_args>::type pa,
                                                              const int flags) {
#if defined(__HIP_DEVICE_COMPILE__)
  using S = Shape<NB>;
  const int flat = flags & 1;  // bit 1: register-direct epilogue (bf16 results without fused column sums)
  constexpr int BN = S::BN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // bits 8-15 (COCODR_PP_STAGGER, experiment): the first round's workgroups start phase x units x 512 clocks late, phase =
  // (id / 8) % 8 (neighbours on one XCD differ), so that the 256 CUs do not reach their epilogues - a burst of 33 MB of
  // stores per round - at the same moment for the rest of the launch
  if (const int stag = (flags >> 8) & 255; stag != 0 && blockIdx.x < 256 && blockIdx.y == 0) {
    const int nph = ((flags >> 16) & 31) + 1;  // phases - 1 in bits 16-20 (a power of two)
    const int units = (((int)blockIdx.x >> 3) & (nph - 1)) * stag;
    for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(8);
  }
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
  // weight gradients fetched 3.5x their operands from the fabric (profiles/r02_gemm_pmc_large_200x128.json).
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
    tile = xcd_remap(blockIdx.x, PERSIST ? ntm * ntn : (int)gridDim.x);
    z = blockIdx.y;
  }
  int tm_, tn_;
  if (TA == 0) grouped_tile(tile, ntm, ntn, 4, tm_, tn_);
  else if (flat) grouped_tile(tile, ntm, ntn, 8, tm_, tn_);
  else { tm_ = tile / ntn; tn_ = tile % ntn; }
  int m0 = tm_ * BM, n0 = tn_ * BN;  // (PERSIST: of the tile being computed; the DMA offsets below stay those of the first tile)
  // PERSIST: virtual block id v = blockIdx.x + k gridDim.x walks this workgroup's tiles (gridDim.x % 8 == 0: v stays on the
  // workgroup's XCD in xcd_remap, and the workgroups of an XCD hold consecutive tile ids at any time, as without the walk)
  [[maybe_unused]] const int total_tiles = ntm * ntn;
  [[maybe_unused]] auto tile_origin = [&](int v, int& mo, int& no) {
    const int tl_ = xcd_remap(v, total_tiles);
    int a_, b_;
    if (TA == 0) grouped_tile(tl_, ntm, ntn, 4, a_, b_);
    else { a_ = tl_ / ntn; b_ = tl_ % ntn; }
    mo = a_ * BM; no = b_ * BN;
  };
  // byte offsets (mod 2^32) of the current / the next tile's operand panels relative to the first tile's
  [[maybe_unused]] uint32_t curA_b = 0, curB_b = 0, nxtA_b = 0, nxtB_b = 0;
  const int m0_first = m0, n0_first = n0;
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

  auto stage_at = [&](auto tyc, int t, const uint32_t base_a, const uint32_t base_b) {  // request half-tile `ty` of K-tile t
    constexpr int ty = decltype(tyc)::value;
#if defined(COCODR_ABL_NO_DMA)
    if (t > 0) return;
#endif
    char* dst = smem + (NSLOT == 10 ? ((4 * t + ty) % 10) * HALF_BYTES : (t & 1) * S::KT_BYTES + ty * HALF_BYTES) + wid * 2048;
    if constexpr (type_is_a<NB>(ty)) {
      const uint32_t sb_ = base_a + (t + t0) * stepa;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (LDS_PTR(void))dst, 16, off[ty][0] + sb_, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (LDS_PTR(void))(dst + 1024), 16, off[ty][1] + sb_, 0, 0, 0);
    } else {
      const uint32_t sb_ = base_b + (t + t0) * stepb;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (LDS_PTR(void))dst, 16, off[ty][0] + sb_, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (LDS_PTR(void))(dst + 1024), 16, off[ty][1] + sb_, 0, 0, 0);
    }
  };
  auto stage = [&](auto tyc, int t) {  // ... of the tile being computed
    if constexpr (PERSIST) stage_at(tyc, t, curA_b, curB_b);
    else stage_at(tyc, t, 0u, 0u);
  };
  [[maybe_unused]] auto stage_next = [&](auto tyc, int k) { stage_at(tyc, k, nxtA_b, nxtB_b); };  // ... K-tile k of the next tile (PERSIST)

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
  if (NSLOT == 10 && nt > 1) {  // the ten-slot ring starts two whole K-tiles deep
    static_for<0, S::NTYPE>([&](auto tyc) { stage(tyc, 1); });
    wait_vmcnt<12>();
  } else if (nt > 1) {
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
  // flight - and bounded the whole loop (DMA-only ablation, profiles/r02_gemm_pp_ablation.txt).
  // rem = K-tiles left including this one: the request exists while its K-tile does; the wait count shrinks with the queue.
  // VAR 6 (experiment): the same request sequence issued in TWO bursts per K-tile instead of one half-tile per phase - nothing in
  // the phases that carry the most fragment reads ((A0, B0): 12, (A1, B1): 8), two half-tiles in the two light ones ((A0, B1): 4
  // reads requests B1 and A1 of K-tile t + 1, (A1, B0): none requests A0 and B0 of K-tile t + 2).  Same slots, same margins.
  // (PERSIST, cont = this workgroup has another tile: where the tile's own request sequence ends, the next tile's begins -
  //  its K-tile 0 lands in buffer 0 and the first half of its K-tile 1 in buffer 1, exactly the slots and margins of a K-tile
  //  nt / nt + 1 of this tile; every wait keeps its steady-state count)
  auto request = [&](auto jc, int t, int rem, [[maybe_unused]] bool cont) {
    constexpr int j = decltype(jc)::value;
    if constexpr (VAR == 6) {
      if constexpr (j == 1) { if (rem >= 2) { stage(std::integral_constant<int, 2>{}, t + 1); stage(std::integral_constant<int, 3>{}, t + 1); } }
      if constexpr (j == 3) { if (rem >= 3) { stage(std::integral_constant<int, 0>{}, t + 2); stage(std::integral_constant<int, 1>{}, t + 2); } }
    } else if constexpr (NSLOT == 10) {
      if (rem >= 3) stage(std::integral_constant<int, j>{}, t + 2);  // the same half-tile type, two K-tiles ahead
    } else if constexpr (PERSIST) {
      if constexpr (j < 2) { if (rem >= 2) stage(std::integral_constant<int, j + 2>{}, t + 1); else if (cont) stage_next(std::integral_constant<int, j + 2>{}, 0); }
      else { if (rem >= 3) stage(std::integral_constant<int, j - 2>{}, t + 2); else if (cont) stage_next(std::integral_constant<int, j - 2>{}, rem == 2 ? 0 : 1); }
    } else {
      if constexpr (j < 2) { if (rem >= 2) stage(std::integral_constant<int, j + 2>{}, t + 1); }
      else { if (rem >= 3) stage(std::integral_constant<int, j - 2>{}, t + 2); }
    }
  };
  // sext (PERSIST, K-tile 0 of a continued tile): store instructions of the previous tile's epilogue that sit in the queue
  // behind the pieces this K-tile waits for - 16 (one result) / 32 (GELU + GELU') per wave, or 0 = unknown (ragged tile: waves
  // past the last row issue none), which makes the wait drain them
  auto wait_stage = [&](auto jc, int rem, [[maybe_unused]] bool cont, [[maybe_unused]] int sext) {
    constexpr int j = decltype(jc)::value;
    if constexpr (NSLOT == 10) {
      // after this phase's request the queue may keep what lies behind half-tile p + 2 (read from phase p + 3 on, retired one
      // phase early as in the eight-slot ring): min(6, 4 rem - j - 3) half-tiles of two pieces each
      if (rem >= 3) wait_vmcnt<12>();
      else if (rem == 2) wait_vmcnt<(j == 0 ? 10 : (j == 1 ? 8 : (j == 2 ? 6 : 4)))>();
      else wait_vmcnt<(j == 0 ? 2 : 0)>();
      return;
    }
    if constexpr (PERSIST) {
      if (rem >= 3 || cont) {
        if (sext == 0) wait_vmcnt<8>();
        else if (sext == 16) wait_vmcnt<24>();
        else wait_vmcnt<40>();
        return;
      }
    }
    if constexpr (VAR == 6) {  // what the NEXT phase reads must have landed; everything requested behind it may stay in flight
      if constexpr (j == 0) { if (rem >= 2) wait_vmcnt<6>(); else wait_vmcnt<2>(); }          // B1(t); behind it A1(t), A0 B0(t+1)
      if constexpr (j == 1) { if (rem >= 2) wait_vmcnt<8>(); else wait_vmcnt<0>(); }          // A1(t); behind it A0 B0 B1 A1(t+1)
      if constexpr (j == 3) { if (rem >= 3) wait_vmcnt<8>(); else if (rem == 2) wait_vmcnt<4>(); else wait_vmcnt<0>(); }  // A0 B0(t+1)
    } else if constexpr (VAR == 7) {  // as VAR 0, but B0 of K-tile t + 1 is read one phase early (phase 3): phase 2's wait retires it
      if (rem >= 3) wait_vmcnt<(j == 2 ? 6 : 8)>();
      else if (rem == 2) wait_vmcnt<(j < 2 ? 8 : 4)>();
      else wait_vmcnt<(j == 0 ? 2 : 0)>();
    } else {
      if (rem >= 3) wait_vmcnt<8>();
      else if (rem == 2) wait_vmcnt<(j < 2 ? 8 : (j == 2 ? 6 : 4))>();
      else wait_vmcnt<(j == 0 ? 2 : 0)>();
    }
  };

  // One K-tile.  STEADY: at least two more K-tiles follow (rem >= 3) - every request exists and every wait is vmcnt(8), so the
  // steady-state loop carries no scalar compare / branch at all (the rem-dependent forms cost 3-6 branches per load segment,
  // a fifth of its 256-cycle budget); the last two K-tiles run the general form.
  // (VAR 7 passes the two B fragment buffers in alternating roles: fbx holds B0 of this K-tile, fby receives B1 and, in phase 3,
  // B0 of the next K-tile)
  auto ktile = [&](auto steady_c, const int t, const int rem_in, FragSet<TB, 1> (&fbx)[4], FragSet<TB, 1> (&fby)[4],
                   [[maybe_unused]] const bool cont = false, [[maybe_unused]] const int sext_in = 0) {
    constexpr bool STEADY = decltype(steady_c)::value;
    const int rem = STEADY ? 3 : rem_in;
    const int sext = STEADY ? 0 : sext_in;
    const uint32_t kb = lds_base + (uint32_t)((t & 1) * S::KT_BYTES);
#pragma unroll
    for (int i = 0; i < 4; ++i) { curA[i] = adA[i] + kb; curB[i] = adB[i] + kb; }
    // one phase; RD: this phase's fragment reads, TY: the half-tile type requested for K-tile t + 1
    auto phase = [&](auto tyc, auto&& reads, const FragSet<TB, 1> (&fb)[4], f32x16& c0, f32x16& c1) {
      auto req = [&]() { request(tyc, t, rem, cont); };
      auto none = []() {};
      if constexpr (VAR == 3) req();
      reads();
      if constexpr (VAR == 0 || VAR == 2 || VAR == 6 || VAR == 7) req();
      wait_stage(tyc, rem, cont, sext);
      __builtin_amdgcn_s_barrier();
      wait_lgkmcnt<0>();
      if constexpr (VAR != 2) __builtin_amdgcn_s_setprio(1);
      pp_mfma<TA, TB, F16>(fa, fb, c0, c1, none);
      if constexpr (VAR != 2) __builtin_amdgcn_s_setprio(0);
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
    } else if constexpr (NB == 2 && VAR == 7) {
      // (experiment builds) Fragment reads per phase 8 / 4 / 8 / 4 instead of 12 / 4 / 8 / 0: B0 of K-tile t + 1 is read in phase 3 of
      // K-tile t - which reads nothing otherwise - into the buffer B1 left in phase 2, so the heaviest load segment (A0 + B0 = 12
      // reads) shrinks to A0 alone.  Bit-identical; measured: NO gain where the steady state is all there is (grouped weight
      // gradient, K = 25 600: 1301 vs 1297 TFLOP/s) - the 12-read phase is not what the loop waits for - and the role swap of the
      // two B buffers costs hipcc ~65 register moves per K-tile and spills in the tail K-tiles (profiles/r03_gemm_experiments.md).
      uint32_t curBn[4];
      const uint32_t kbn = lds_base + (uint32_t)(((t + 1) & 1) * S::KT_BYTES);
#pragma unroll
      for (int i = 0; i < 4; ++i) curBn[i] = adB[i] + kbn;
      phase(std::integral_constant<int, 0>{}, [&]() { pp_read_sub<TA, 2, 0 * HALF_BYTES>(curA, fa); }, fbx, acc[0][0], acc[1][0]);    // (A0, B0)
      phase(std::integral_constant<int, 1>{}, [&]() { pp_read_sub<TB, 1, 2 * HALF_BYTES>(curB, fby); }, fby, acc[0][1], acc[1][1]);   // (A0, B1)
      phase(std::integral_constant<int, 2>{}, [&]() { pp_read_sub<TA, 2, 3 * HALF_BYTES>(curA, fa); }, fby, acc[2][1], acc[3][1]);    // (A1, B1)
      phase(std::integral_constant<int, 3>{}, [&]() { if (rem >= 2) pp_read_sub<TB, 1, 1 * HALF_BYTES>(curBn, fby); },                // (A1, B0)
            fbx, acc[2][0], acc[3][0]);
    } else if constexpr (NB == 2 && NSLOT == 10) {
      // fragment addresses per half-tile: its slot is (4 t + type) % 10, a different one every K-tile (period 5)
      auto at = [&](const uint32_t (&ad)[4], int ty, uint32_t (&cur)[4]) {
        const uint32_t b = lds_base + (uint32_t)(((4 * t + ty) % 10) * HALF_BYTES);
#pragma unroll
        for (int i = 0; i < 4; ++i) cur[i] = ad[i] + b;
      };
      phase(std::integral_constant<int, 0>{}, [&]() { at(adA, 0, curA); pp_read_sub<TA, 2, 0>(curA, fa); at(adB, 1, curB); pp_read_sub<TB, 1, 0>(curB, fb0); },
            fb0, acc[0][0], acc[1][0]);                                                                  // (A0, B0)
      phase(std::integral_constant<int, 1>{}, [&]() { at(adB, 2, curB); pp_read_sub<TB, 1, 0>(curB, fb1); }, fb1, acc[0][1], acc[1][1]);  // (A0, B1)
      phase(std::integral_constant<int, 2>{}, [&]() { at(adA, 3, curA); pp_read_sub<TA, 2, 0>(curA, fa); }, fb1, acc[2][1], acc[3][1]);   // (A1, B1)
      phase(std::integral_constant<int, 3>{}, [&]() {}, fb0, acc[2][0], acc[3][0]);                      // (A1, B0): all in registers
    } else if constexpr (NB == 2) {
      phase(std::integral_constant<int, 0>{}, [&]() { pp_read_sub<TA, 2, 0 * HALF_BYTES>(curA, fa); pp_read_sub<TB, 1, 1 * HALF_BYTES>(curB, fb0); },
            fb0, acc[0][0], acc[1][0]);                                                                  // (A0, B0)
      phase(std::integral_constant<int, 1>{}, [&]() { pp_read_sub<TB, 1, 2 * HALF_BYTES>(curB, fb1); }, fb1, acc[0][1], acc[1][1]);  // (A0, B1)
      phase(std::integral_constant<int, 2>{}, [&]() { pp_read_sub<TA, 2, 3 * HALF_BYTES>(curA, fa); }, fb1, acc[2][1], acc[3][1]);   // (A1, B1)
      phase(std::integral_constant<int, 3>{}, [&]() {}, fb0, acc[2][0], acc[3][0]);                      // (A1, B0): all in registers
    }
  };
  if constexpr (VAR == 7 && NB == 2) {
    {  // B0 of K-tile 0 (landed: the prologue waited for A0 and B0)
      uint32_t c0[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) c0[i] = adB[i] + lds_base;
      pp_read_sub<TB, 1, 1 * HALF_BYTES>(c0, fb0);
    }
    // the B fragment buffers swap roles every K-tile: pairs of K-tiles with static roles, the parity of what is left decided
    // once (a role picked at run time per K-tile puts the fragment arrays in scratch)
    const int ns = nt > 2 ? nt - 2 : 0;  // K-tiles in the branch-free steady form
    int t = 0;
    for (; t + 1 < ns; t += 2) {
      ktile(std::true_type{}, t, 3, fb0, fb1);
      ktile(std::true_type{}, t + 1, 3, fb1, fb0);
    }
    if (ns & 1) {
      ktile(std::true_type{}, t, 3, fb0, fb1);
      ++t;
      ktile(std::false_type{}, t, nt - t, fb1, fb0);
      if (t + 1 < nt) ktile(std::false_type{}, t + 1, nt - t - 1, fb0, fb1);
    } else {
      ktile(std::false_type{}, t, nt - t, fb0, fb1);
      if (t + 1 < nt) ktile(std::false_type{}, t + 1, nt - t - 1, fb1, fb0);
    }
  } else if constexpr (PERSIST) {
    static_assert(VAR == 0 && NB == 2 && !OUT_F32 && !MULTI, "the persistent walk exists for the bf16-result forward / dgrad forms");
    int vb = (int)blockIdx.x;
    int sext = -1;  // < 0: the first tile (prologue above); else the store count of the previous epilogue for K-tile 0's waits
    for (;;) {
      const int vn = vb + (int)gridDim.x;
      const bool cont = vn < total_tiles;
      int m0n = 0, n0n = 0;
      if (cont) {
        tile_origin(vn, m0n, n0n);
        nxtA_b = (uint32_t)(m0n - m0_first) * (TA ? 2u : (uint32_t)(p.lda * 2));
        nxtB_b = (uint32_t)(n0n - n0_first) * (TB ? 2u : (uint32_t)(p.ldb * 2));
      }
      int t = 0;
      if (sext >= 0) { ktile(std::false_type{}, 0, nt, fb0, fb1, cont, sext); t = 1; }
      for (; t < nt - 2; ++t) ktile(std::true_type{}, t, 3, fb0, fb1);
      for (; t < nt; ++t) ktile(std::false_type{}, t, nt - t, fb0, fb1, cont, 0);
      if (wr == 0) __builtin_amdgcn_s_barrier();  // group 0 catches up with group 1's last barrier
      if constexpr (PERSIST == 2) pp_reg_epilogue<NB>(p, z, acc, m0, n0, wr, wc, lane);  // (COCODR_PP_PERSIST=2: A/B switch)
      else pp_lds4_epilogue(p, z, acc, m0, n0, wr, wc, tid, lane, lds_base);
      if (!cont) break;
      sext = (m0 + BM <= p.M) ? ((p.epi == COCODR_EPI_GELU && p.C2 != nullptr) ? 32 : 16) : 0;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
      // A0, B0 of the next tile's K-tile 0 were retired by every wave's waits of the last K-tile: one barrier makes all of
      // them visible, the second one puts group 1 a barrier behind group 0 again
      __builtin_amdgcn_s_barrier();
      if (wr == 1) __builtin_amdgcn_s_barrier();
      vb = vn; m0 = m0n; n0 = n0n; curA_b = nxtA_b; curB_b = nxtB_b;
    }
    return;
  } else if constexpr ((VAR == 0 || VAR == 6) && NB == 2) {
    int t = 0;
    if (!(flags & 4))  // (bit 2: A/B switch COCODR_PP_NOPEEL - every K-tile in the general form)
      for (; t < nt - 2; ++t) ktile(std::true_type{}, t, 3, fb0, fb1);
    for (; t < nt; ++t) ktile(std::false_type{}, t, nt - t, fb0, fb1);
  } else {
    for (int t = 0; t < nt; ++t) ktile(std::false_type{}, t, nt - t, fb0, fb1);
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

  if constexpr (!OUT_F32) {
    if ((flags & 2) && p.colsum_partial == nullptr) {
      pp_reg_epilogue<NB>(p, z, acc, m0, n0, wr, wc, lane);
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
  static int flat_env = -1;  // COCODR_PP_FLAT=0 keeps the per-item remap (A/B switch)
  if (flat_env < 0) {
    const char* e = getenv("COCODR_PP_FLAT");
    flat_env = e ? atoi(e) : 1;
  }
  const int flat = (a.batch > 1 && flat_env) ? 1 : 0;
  static const int regepi = getenv("COCODR_PP_REGEPI") ? atoi(getenv("COCODR_PP_REGEPI")) : 0;  // A/B switch: 1 = the register-direct epilogue (measured 4 % slower in the step, see pp_reg_epilogue)
  static const int nopeel = getenv("COCODR_PP_NOPEEL") ? atoi(getenv("COCODR_PP_NOPEEL")) : 0;  // A/B switch of the branch-free steady-state loop
  // first-round stagger (see the kernel; experiment, off unless COCODR_PP_STAGGER=units): 32 phases x units x 512 clocks.  Only
  // from four rounds of tiles on and where the last round is partly empty - the late starters then take no tile of that round,
  // while with whole rounds the launch ends as much later as it started (8192 x 4096 x 1024, two whole rounds: -8 %).  In the
  // BERT-large step at 200 sequences units = 2 measured +1.4 ... +2.2 % on three boxes and -0.4 % on a fourth, nothing elsewhere
  // (profiles/r03_gemm_pp_stagger.md): not shipped as a default.
  static const int stagger_env = getenv("COCODR_PP_STAGGER") ? atoi(getenv("COCODR_PP_STAGGER")) & 255 : 0;
  static const int stagger_ph = getenv("COCODR_PP_STAGGER_PH") ? atoi(getenv("COCODR_PP_STAGGER_PH")) : 32;
  static const int stagger_all = getenv("COCODR_PP_STAGGER_ALL") != nullptr;  // A/B: also launches of whole rounds
  const long long tiles = (long long)ntm * ntn * (a.batch > 0 ? a.batch : 1);
  const int rem = (int)(tiles % 256);
  const int stagger = (stagger_all || (tiles >= 1024 && rem >= 1 && rem <= 208)) ? stagger_env : 0;
  const int flags = flat | (regepi ? 2 : 0) | (nopeel ? 4 : 0) | (stagger << 8) | (((stagger_ph - 1) & 31) << 16);
  dim3 grid(flat ? ntm * ntn * a.batch : ntm * ntn, flat ? 1 : a.batch);
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_pp_kernel<NB, TA, TB, true, VAR, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)gemm_pp_kernel<NB, TA, TB, false, VAR, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
#if defined(COCODR_PP_PERSIST_BUILD)  // experiment builds (COCODR_EXTRA_FLAGS=-DCOCODR_PP_PERSIST_BUILD): measured, not adopted - see the kernel
  if constexpr (NB == 2 && VAR == 0 && !F16) {
    // persistent walk (see the kernel): bf16 results without fused column sums, more tiles than CUs, an even number (>= 4) of
    // K-tiles.  Grid: the fewest workgroups that need no more rounds than 256 would, a multiple of 8 (1200 tiles -> 240 x 5).
    static const int persist = getenv("COCODR_PP_PERSIST") ? atoi(getenv("COCODR_PP_PERSIST")) : 0;
    const int nkt = (a.K + BK - 1) / BK;
    if (persist && !a.out_f32 && a.batch <= 1 && a.colsum_partial == nullptr && a.colsum == nullptr && tiles > 256 && nkt >= 4 && nkt % 2 == 0 &&
        a.K % BK == 0) {
      const int rounds = (int)((tiles + 255) / 256);
      int g = (int)((tiles + rounds - 1) / rounds);
      g = (g + 7) & ~7;
      if (g > 256) g = 256;
      static bool attr_p = false;
      if (!attr_p) {
        hipFuncSetAttribute((const void*)gemm_pp_kernel<NB, TA, TB, false, VAR, F16, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)gemm_pp_kernel<NB, TA, TB, false, VAR, F16, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_p = true;
      }
      if (persist == 2)
        hipLaunchKernelGGL((gemm_pp_kernel<NB, TA, TB, false, VAR, F16, false, 2>), dim3(g), dim3(NTHREADS), 160 * 1024, st, a, flags & ~(255 << 8));
      else
        hipLaunchKernelGGL((gemm_pp_kernel<NB, TA, TB, false, VAR, F16, false, 1>), dim3(g), dim3(NTHREADS), 160 * 1024, st, a, flags & ~(255 << 8));
      return;
    }
  }
#endif
#if defined(COCODR_PP_RING10_BUILD)  // experiment builds (COCODR_EXTRA_FLAGS=-DCOCODR_PP_RING10_BUILD): measured, not adopted - see the kernel
  if constexpr (NB == 2 && VAR == 0 && !F16) {
    static const int ring = getenv("COCODR_PP_RING") ? atoi(getenv("COCODR_PP_RING")) : 8;  // A/B switch: 10 = the ten-slot ring (see the kernel)
    if (ring == 10) {
      static bool attr_r = false;
      if (!attr_r) {
        hipFuncSetAttribute((const void*)gemm_pp_kernel<NB, TA, TB, true, VAR, F16, false, 0, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)gemm_pp_kernel<NB, TA, TB, false, VAR, F16, false, 0, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_r = true;
      }
      if (a.out_f32)
        hipLaunchKernelGGL((gemm_pp_kernel<NB, TA, TB, true, VAR, F16, false, 0, 10>), grid, dim3(NTHREADS), 160 * 1024, st, a, flags);
      else
        hipLaunchKernelGGL((gemm_pp_kernel<NB, TA, TB, false, VAR, F16, false, 0, 10>), grid, dim3(NTHREADS), 160 * 1024, st, a, flags);
      return;
    }
  }
#endif
  if (a.out_f32)
    hipLaunchKernelGGL((gemm_pp_kernel<NB, TA, TB, true, VAR, F16>), grid, dim3(NTHREADS), S::LDS_BYTES, st, a, flags);
  else
    hipLaunchKernelGGL((gemm_pp_kernel<NB, TA, TB, false, VAR, F16>), grid, dim3(NTHREADS), S::LDS_BYTES, st, a, flags);
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
#if defined(COCODR_PP_RING10_BUILD)
  static const int ring = getenv("COCODR_PP_RING") ? atoi(getenv("COCODR_PP_RING")) : 8;  // A/B switch: 10 = the ten-slot ring (see the kernel)
  auto kern = ring == 10 ? gemm_pp_kernel<2, 1, 1, true, 0, false, true, 0, 10> : gemm_pp_kernel<2, 1, 1, true, 0, false, true>;
#else
  constexpr int ring = 8;
  // (two fat phases per K-tile for the merged launch, gemm_pp_kernel<2, 1, 1, true, 5, false, true>: measured +0.3 % on the
  //  BERT-base step, -0.3 ... -0.5 % on the BERT-large ones; not kept)
  auto kern = gemm_pp_kernel<2, 1, 1, true, 0, false, true>;
#endif
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  static const int nopeel = getenv("COCODR_PP_NOPEEL") ? atoi(getenv("COCODR_PP_NOPEEL")) : 0;
  hipLaunchKernelGGL(kern, dim3(split ? total - r + r * s : total), dim3(NTHREADS), ring == 10 ? 160 * 1024 : Shape<2>::LDS_BYTES, st, ma, 1 | (nopeel ? 4 : 0));
  if (split) hipLaunchKernelGGL(gemm_pp_split_finish, dim3(r, SPLIT_TILE / 4 / 256), dim3(256), 0, st, ma, r);
}

// nb = 2: 256 x 256 tile (N % 256 == 0), nb = 1: 256 x 128 tile; the caller has validated the arguments (cocodr_gemm)
template <int VAR>
static void launch_any(const cocodr_gemm_args& a, hipStream_t st) {
  using namespace cocodr_gemm_pp;
  if (!a.trans_a && !a.trans_b) launch_form<2, 0, 0, VAR>(a, st);
  else if (!a.trans_a && a.trans_b) launch_form<2, 0, 1, VAR>(a, st);
  else launch_form<2, 1, 1, VAR>(a, st);
}
void cocodr_gemm_pp_launch(const cocodr_gemm_args& a, int nb, hipStream_t st) {
  static const int b0early = getenv("COCODR_PP_B0EARLY") ? atoi(getenv("COCODR_PP_B0EARLY")) : 0;  // A/B switch of VAR 7
  static const int fat = getenv("COCODR_PP_FAT") ? atoi(getenv("COCODR_PP_FAT")) : 0;  // A/B switch: 1 = fat phases everywhere,
  if (nb == 2 && (fat == 1 || (fat == 2 && !a.trans_a) || (fat == 3 && a.trans_a))) launch_any<5>(a, st);  // 2 = forward / dgrad only, 3 = wgrads only
#if defined(COCODR_PP_VARIANTS)
  else if (nb == 2 && getenv("COCODR_PP_VAR") && atoi(getenv("COCODR_PP_VAR")) == 6) launch_any<6>(a, st);  // in-step A/B of VAR 6
  else if (nb == 107 || (nb == 2 && b0early)) launch_any<7>(a, st);              // B0 fragments read one phase early (8 / 4 / 8 / 4 reads)
#endif
  else if (nb == 2) launch_any<0>(a, st);                                        // four thin phases per K-tile (the default)
  else if (nb == 104) cocodr_gemm_pp::launch_form<2, 0, 0, 5, true>(a, st);      // IEEE-half operands (the search): fat phases
  else if (nb == 105) launch_any<5>(a, st);                                      // impl 18: two fat phases per K-tile
#if defined(COCODR_PP_VARIANTS)
  else if (nb == 102) launch_any<2>(a, st);
  else if (nb == 103) launch_any<3>(a, st);
  else if (nb == 106) launch_any<6>(a, st);
#endif
  else launch_any<0>(a, st);
}
