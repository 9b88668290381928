// "One wave per SIMD" bf16 MFMA GEMM for gfx950: 256 x 256 tile, FOUR waves (2 x 2), each owning 128 x 128 of the tile with its
// 16 accumulator blocks (256 registers) in the accumulator half of the unified 512-entry register file.
//
// Why a third pipeline (round 3, profiles/r03_gemm_w4.md).  Ablating the ping-pong loop (gemm_pp.hip) at the BERT-large shapes:
// MFMAs alone 114.6 us, MFMAs + operand DMA 123.0, everything 164.7 - the LDS fragment reads, not the DMA, are what the loop
// waits for.  Eight waves of 128 x 64 read (128 + 64) rows of 128 B per 64-deep K-tile each = 192 KB per CU per K-tile; next to
// the 64 KB the DMA writes, that is more than the LDS moves in the 2048 matrix-pipe cycles of the K-tile.  Four waves of 128 x 128
// read (128 + 128) rows each = 128 KB: a third less LDS traffic per FLOP, and the only way to a 128 x 128 wave tile is one wave
// per SIMD with the accumulators in AGPRs.  Such a wave has no partner to hide its loads behind, so its own instruction stream
// interleaves them: per 16-MFMA sub-step 8 fragment reads and (in the first sub-step of a stage) the DMA requests ride in the
// shadow of the MFMAs (<= 5 issue slots per 32-cycle MFMA, MI355X_MICROARCH.md "one wave per SIMD").
//
// Stages: [256 rows][32 k] of A and of B (16 KiB each), five in the 160 KiB of LDS.  Stage t + 4 is requested at the top of
// sub-tile t (its slot was last read in sub-tile t - 1), stage t + 1 has landed when sub-tile t starts (every wave waits for its
// own pieces in front of the barrier that opens t), so the first fragments of t + 1 are read under the last MFMAs of t and the
// matrix pipe never waits for a barrier followed by an LDS round trip.  Three stages (96 KiB per CU) are in flight, ~1.5 us
// between request and first use.  64-byte rows: 16 rows per 1-KiB DMA piece, chunk swizzle (row >> 2) & 3 on the source
// address and on the read (both conflict-free, as gemm.hip's BK = 32 geometries).
//
// Forms / epilogues: NT and NN forms with the bf16 epilogues of gemm_tile.h (bias, erf-GELU + GELU', residual, x GELU',
// dropout, fused column sums are NOT here yet: the selection in gemm.hip only sends plain / bias / GELU / residual forms).
#include <stdlib.h>

#include "common.h"
#include "gemm_tile.h"

namespace cocodr_gemm_w4 {
using namespace cocodr_gemm_v2;

constexpr int BM = 256, BN = 256, BKS = 32;        // sub-tile depth
constexpr int NSTAGE = 5;
constexpr int OP_BYTES = 256 * BKS * 2;            // 16 KiB: [256 rows][32 k] bf16
constexpr int STAGE_BYTES = 2 * OP_BYTES;          // A then B
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;    // 160 KiB
constexpr int NTHREADS = 256;
constexpr int PIECES = 4;                          // 1-KiB DMA pieces per wave, operand and stage (16 rows each)

__device__ __forceinline__ int swz32(int row) { return (row >> 2) & 3; }

template <int TA, int TB, bool OUT_F32>
__global__ __launch_bounds__(NTHREADS, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w4_kernel(const cocodr_gemm_args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(TA == 0 && TB == 0, "prototype: NT form");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int ntn = p.N / BN, ntm = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int z = blockIdx.y;
  int tm_, tn_;
  grouped_tile(tile, ntm, ntn, 4, tm_, tn_);
  const int m0 = tm_ * BM, n0 = tn_ * BN;
  const uint16_t* A = p.A + (size_t)z * p.strideA;
  const uint16_t* B = p.B + (size_t)z * p.strideB;
  const uint32_t a_bytes = (uint32_t)((size_t)p.M * p.lda * 2);
  const uint32_t b_bytes = (uint32_t)((size_t)p.N * p.ldb * 2);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);

  // DMA: piece q (0..15) of an operand stage = rows 16 q .. + 15; wave w requests pieces 4 w .. 4 w + 3 of A and of B.
  // lane -> row 16 q + (lane >> 2), stored chunk lane & 3 = logical chunk (lane & 3) ^ ((lane >> 4) & 3)
  const int prow = lane >> 2, pch = (lane & 3) ^ ((lane >> 4) & 3);
  const uint32_t va = (uint32_t)(((m0 + wid * 64 + prow) * p.lda + pch * 8) * 2);
  const uint32_t vb = (uint32_t)(((n0 + wid * 64 + prow) * p.ldb + pch * 8) * 2);
  const uint32_t pa = (uint32_t)(16 * p.lda * 2), pb = (uint32_t)(16 * p.ldb * 2);  // next piece: 16 rows down
  const int nst = (p.K + BKS - 1) / BKS;  // sub-tiles (the caller guarantees K % 32 == 0)

  // one DMA piece of sub-tile t: i < 4 -> A piece i, else B piece i - 4 (the piece offset rides in the VGPR offset: the range
  // check that zero-fills rows past M does not see the scalar offset)
  auto dma_piece = [&](auto ic, int t) {
    constexpr int i = decltype(ic)::value;
    char* dst = smem + (t % NSTAGE) * STAGE_BYTES + wid * (PIECES * 1024);
    const uint32_t kb = (uint32_t)(t * BKS * 2);
    if constexpr (i < PIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (LDS_PTR(void))(dst + i * 1024), 16, va + kb + i * pa, 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (LDS_PTR(void))(dst + OP_BYTES + (i - PIECES) * 1024), 16, vb + kb + (i - PIECES) * pb, 0, 0, 0);
  };
  auto stage = [&](int t) { static_for<0, 2 * PIECES>([&](auto ic) { dma_piece(ic, t); }); };

  // fragment addresses: lane -> row (lane & 31) of its 32-row block, logical chunk 2 s + (lane >> 5); s = 1 flips bit 5
  const uint32_t lds_base = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;
  uint32_t adA[2], adB[2];
  {
    const int ra_ = wr * 128 + (lane & 31), rb_ = wc * 128 + (lane & 31), h = lane >> 5;
    adA[0] = lds_base + (uint32_t)(ra_ * 64 + ((h ^ swz32(ra_)) << 4));
    adB[0] = lds_base + (uint32_t)(OP_BYTES + rb_ * 64 + ((h ^ swz32(rb_)) << 4));
    adA[1] = adA[0] ^ 32u;
    adB[1] = adB[0] ^ 32u;
  }

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- prologue: sub-tiles 0 .. 3 requested; 0 and 1 landed before the loop (1 so that its first fragments can be read early)
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (t < nst) stage(t);
  if (nst >= 4) wait_vmcnt<16>();
  else if (nst == 3) wait_vmcnt<8>();
  else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

#if defined(COCODR_ABL_NO_LDSREAD)
  v4i fa[2][4] = {}, fb[2][4] = {};
#else
  v4i fa[2][4], fb[2][4];  // [buffer][fragment]
#endif
  // fragment read i of a sub-step: i < 4 -> A block i, else B block i - 4 (32 rows = 2048 B apart)
  auto read_one = [&](auto ic, auto bufc, uint32_t aA, uint32_t aB) {
    constexpr int i = decltype(ic)::value, buf = decltype(bufc)::value;
    if constexpr (i < 4) asm_ds_read_b128<i * 2048>(fa[buf][i], aA);
    else asm_ds_read_b128<(i - 4) * 2048>(fb[buf][i - 4], aB);
  };
  auto mfma_one = [&](auto ic, auto bufc) {
    constexpr int i = decltype(ic)::value, buf = decltype(bufc)::value, a = i >> 2, b = i & 3;
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[buf][b]), __builtin_bit_cast(bf16x8, fa[buf][a]), acc[a][b], 0, 0, 0);
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;

  static_for<0, 8>([&](auto ic) { read_one(ic, B0{}, adA[0], adB[0]); });  // fragments of (sub-tile 0, sub-step 0)
  // slot byte offsets of sub-tiles t, t + 1 and t + 4, advanced by one slot per sub-tile (no division in the loop)
  uint32_t so = 0, sn = STAGE_BYTES, s4 = 4 * STAGE_BYTES;
  auto next_slot = [](uint32_t x) { return x + STAGE_BYTES == (uint32_t)LDS_BYTES ? 0u : x + STAGE_BYTES; };
  // One sub-tile.  STEADY: sub-tile t + 4 exists (so t + 1 .. t + 3 do as well): no condition anywhere in the body.
  auto subtile = [&](auto steady_c, const int t) {
    constexpr bool STEADY = decltype(steady_c)::value;
    const uint32_t a1 = adA[1] + so, b1 = adB[1] + so, a0n = adA[0] + sn, b0n = adB[0] + sn;
    const bool more = STEADY || t + 4 < nst, next = STEADY || t + 1 < nst;
    char* dst = smem + s4 + wid * (PIECES * 1024);
    const uint32_t kb = (uint32_t)((t + 4) * BKS * 2);
    // The 32 DMA pieces of sub-tile t + 4 (8 per wave) are spread over the 32 MFMA slots of the sub-tile, ONE piece per slot and
    // CU: in slot i the wave with wid == (i & 3) requests its piece 4 substep + (i >> 2).  The four waves run in step (one
    // barrier per sub-tile), so the CU's address unit sees one 1-KiB request per 32-cycle slot instead of four at once - a
    // wave that has to queue for it issues nothing else, and with one wave per SIMD nobody else feeds the matrix pipe meanwhile.
    auto dma_slot = [&](auto ic, auto subc) {
      constexpr int i = decltype(ic)::value, q = decltype(subc)::value * 4 + (i >> 2);
#if defined(COCODR_ABL_NO_DMA)
      if (false) {
#elif defined(COCODR_ABL_W4_BURST)
      if (more && decltype(subc)::value == 0 && i >= 8) {  // (burst form: all 8 pieces of a wave in slots 8..15 of sub-step 0)
        constexpr int q2 = i - 8;
        if constexpr (q2 < PIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (LDS_PTR(void))(dst + q2 * 1024), 16, va + kb + q2 * pa, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (LDS_PTR(void))(dst + OP_BYTES + (q2 - PIECES) * 1024), 16, vb + kb + (q2 - PIECES) * pb, 0, 0, 0);
      }
      if (false) {
#else
      if (more && wid == (i & 3)) {
#endif
        if constexpr (q < PIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (LDS_PTR(void))(dst + q * 1024), 16, va + kb + q * pa, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (LDS_PTR(void))(dst + OP_BYTES + (q - PIECES) * 1024), 16, vb + kb + (q - PIECES) * pb, 0, 0, 0);
      }
    };
    // sub-step 0: 16 MFMAs on buffer 0; in their shadow the 8 fragment reads of sub-step 1
    wait_lgkmcnt<0>();
    static_for<0, 16>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      mfma_one(ic, B0{});
#if !defined(COCODR_ABL_NO_LDSREAD)
      if constexpr (i < 8) read_one(ic, B1{}, a1, b1);
#endif
      dma_slot(ic, B0{});
      __builtin_amdgcn_sched_barrier(0);
    });
    // sub-step 1: 16 MFMAs on buffer 1; in their shadow the first fragments of sub-tile t + 1 (landed before this sub-tile began)
    wait_lgkmcnt<0>();
    static_for<0, 16>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      mfma_one(ic, B1{});
#if !defined(COCODR_ABL_NO_LDSREAD)
      if constexpr (i < 8) { if (next) read_one(ic, B0{}, a0n, b0n); }
#endif
      dma_slot(ic, B1{});
      __builtin_amdgcn_sched_barrier(0);
    });
    // sub-tile t + 2 must have landed before anybody starts t + 1: my own pieces, then everybody's (barrier).  Behind it in my
    // queue: the pieces of t + 3 and t + 4 where they exist.
#if defined(COCODR_ABL_NO_DMA)
    if (true) {
      wait_vmcnt<0>();
    } else
#endif
    if constexpr (STEADY) {
      wait_vmcnt<16>();
    } else {
      const int ahead = min(nst - 1, t + 4) - (t + 2);  // requested sub-tiles behind t + 2: 2, 1 or none
      if (ahead >= 2) wait_vmcnt<16>();
      else if (ahead == 1) wait_vmcnt<8>();
      else wait_vmcnt<0>();
    }
#if !defined(COCODR_ABL_NO_BARRIER)
    __builtin_amdgcn_s_barrier();
#endif
    so = sn;
    sn = next_slot(sn);
    s4 = next_slot(s4);
  };
  int t = 0;
  for (; t < nst - 4; ++t) subtile(std::true_type{}, t);
  for (; t < nst; ++t) subtile(std::false_type{}, t);

  // ---- epilogue (prototype): register-direct stores in the accumulator layout
  const int rl = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int gm = m0 + wr * 128 + a * 32 + rl;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int gn = n0 + wc * 128 + b * 32 + 8 * rg + 4 * hh;
        float v[4] = {acc[a][b][rg * 4 + 0], acc[a][b][rg * 4 + 1], acc[a][b][rg * 4 + 2], acc[a][b][rg * 4 + 3]};
        if (p.bias) {
          const float4 bv = *reinterpret_cast<const float4*>(p.bias + (size_t)z * p.strideBias + gn);
          v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
        }
        if (gm < p.M) {
          if (OUT_F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (size_t)z * p.strideC + (size_t)gm * p.ldc + gn) = make_float4(v[0], v[1], v[2], v[3]);
          else *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.C) + (size_t)z * p.strideC + (size_t)gm * p.ldc + gn) = pack4(v);
        }
      }
  }
#endif
}

template <int TA, int TB>
void launch_form(const cocodr_gemm_args& a, hipStream_t st) {
  const int ntm = (a.M + BM - 1) / BM, ntn = a.N / BN;
  dim3 grid(ntm * ntn, a.batch);
  static cocodr_lds_once attr_done;
  if (attr_done.pending()) {
    hipFuncSetAttribute((const void*)gemm_w4_kernel<TA, TB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)gemm_w4_kernel<TA, TB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done.done();
  }
  if (a.out_f32) hipLaunchKernelGGL((gemm_w4_kernel<TA, TB, true>), grid, dim3(NTHREADS), LDS_BYTES, st, a);
  else hipLaunchKernelGGL((gemm_w4_kernel<TA, TB, false>), grid, dim3(NTHREADS), LDS_BYTES, st, a);
}
}  // namespace cocodr_gemm_w4

// the caller (cocodr_gemm) has validated the arguments; returns false when this pipeline does not take the call
bool cocodr_gemm_w4_launch(const cocodr_gemm_args& a, hipStream_t st) {
  if (a.trans_a || a.trans_b || a.N % 256 != 0 || a.K % 32 != 0) return false;
  if (a.epi != COCODR_EPI_NONE || a.colsum || a.colsum_partial || a.ab_f16) return false;
  if ((size_t)a.M * a.lda * 2 >= (1ull << 32) || (size_t)a.N * a.ldb * 2 >= (1ull << 32)) return false;
  cocodr_gemm_w4::launch_form<0, 0>(a, st);
  return true;
}
