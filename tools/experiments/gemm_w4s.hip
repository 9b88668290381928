// "One wave per SIMD, register-staged" bf16 MFMA GEMM for gfx950, 192 x 256 tile: FOUR waves (2 x 2), each owning 96 x 128 of the
// tile = 3 x 4 accumulator blocks (192 registers).  Same pipeline as gemm_w4r.hip (operands global -> registers -> LDS, "issue
// early, write late", one barrier per 32-deep sub-tile certifying the stage two ahead) with TWELVE accumulators per wave instead of
// sixteen: tools/probes/mfma_rate.hip shows that a wave streaming MFMAs into 16 accumulator blocks runs at 1.4-1.6 PFLOP/s on the
// whole chip whichever register file holds them (pinned by asm constraints: 1.52-1.62 in ArchVGPRs, 1.39-1.50 in AccVGPRs) while 12,
// 14 or 15 blocks run at 2.0-2.1 - the cliff is the 16th block, not the accumulator file (the reading of
// profiles/r04_gemm_one_wave_per_simd.md's first version).  Fragment bytes per MAC: (96 + 128) / (96 x 128) = 0.0182 against 0.0234
// for the ping-pong pipeline's 128 x 64 wave tiles (and 0.0156 for 128 x 128).
//
// Per sub-tile and wave: 24 MFMAs, 14 fragment reads, 7 LDS writes, 7 global loads (3 A + 4 B pieces of 16 rows).
// Forms / epilogues: the NT form with every bf16 / fp32 epilogue of gemm_tile.h (no fused column sums), two 96-row LDS passes.
//
// RESULT (tools/w4s_check.py, one MI355X): bit-identical to the ping-pong pipeline on every shape tried, and 7-45 % SLOWER - cube 8192:
// 1052 against 1332 TFLOP/s, XL fwd qkv 755 against 970, packed base qkv 403 against 707 - slower even than the 16-accumulator
// gemm_w4r.hip (1215 / 825).  Ablations (-DCOCODR_W4R_ABL_*): without the global loads 1352 / 884, without the fragment reads
// 1168 / 906, without the barrier 1069 / 768: no single piece is the limit, every memory instruction costs its issue slots in the
// one MFMA stream a SIMD has, and nothing else runs there meanwhile.  Not adopted; the eight-wave ping-pong shape stays.
#include <stdlib.h>

#include "common.h"
#include "gemm_tile.h"

namespace cocodr_gemm_w4s {
using namespace cocodr_gemm_v2;

constexpr int BM = 192, BN = 256, BKS = 32;
constexpr int NSLOT = 3;
constexpr int A_BYTES = BM * BKS * 2;              // 12 KiB: [192 rows][32 k] bf16
constexpr int B_BYTES = BN * BKS * 2;              // 16 KiB
constexpr int OP_BYTES = A_BYTES;                  // B follows A inside a stage
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;     // 28 KiB
constexpr int RING_BYTES = NSLOT * STAGE_BYTES;    // 96 KiB
constexpr int CT_LD = BN + 4;                      // fp32 epilogue tile leading dimension
constexpr int EPI_BYTES = 96 * CT_LD * 4 + 4 * BN * 4;
constexpr int LDS_BYTES = EPI_BYTES > RING_BYTES ? EPI_BYTES : RING_BYTES;
constexpr int NTHREADS = 256;
constexpr int DIST = 4;                            // global loads run this many sub-tiles ahead of the MFMAs
constexpr int NSET = 3;                            // register sets of staged operands (sub-tiles t + 2 .. t + 4)

template <bool OUT_F32>
__global__ __launch_bounds__(NTHREADS, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w4s_kernel(const cocodr_gemm_args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int ntn = p.N / BN, ntm = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  int tm_, tn_;
  grouped_tile(tile, ntm, ntn, 4, tm_, tn_);
  const int m0 = tm_ * BM, n0 = tn_ * BN;
  const uint32_t a_bytes = (uint32_t)((size_t)p.M * p.lda * 2);
  const uint32_t b_bytes = (uint32_t)((size_t)p.N * p.ldb * 2);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, b_bytes, 0x00020000);

  // staging: piece q of an operand stage = rows 16 q .. + 15 (1 KiB); wave w moves A pieces 3 w .. 3 w + 2 and B pieces 4 w .. 4 w + 3.
  // lane -> row 16 q + (lane >> 2), stored chunk lane & 3 = logical chunk (lane & 3) ^ ((row >> 2) & 3) (rows past M / N read 0)
  const int prow = lane >> 2, pch = (lane & 3) ^ ((lane >> 4) & 3);
  const uint32_t va = (uint32_t)(((m0 + wid * 48 + prow) * p.lda + pch * 8) * 2);
  const uint32_t vb = (uint32_t)(((n0 + wid * 64 + prow) * p.ldb + pch * 8) * 2);
  const uint32_t pa = (uint32_t)(16 * p.lda * 2), pb = (uint32_t)(16 * p.ldb * 2);
  const int nst = p.K / BKS;  // sub-tiles (the caller guarantees K % 32 == 0, K >= 32 * (DIST + 1))
  const uint32_t wstA = (uint32_t)(wid * 3072 + lane * 16);            // this lane's 16 bytes inside the wave's three A pieces
  const uint32_t wstB = (uint32_t)(A_BYTES + wid * 4096 + lane * 16);  // ... and inside its four B pieces

  v4i st[NSET][7];  // staged operand chunks: [set][A pieces 0-2 | B pieces 0-3]
  auto gload = [&](auto ic, auto setc, int t) {  // piece i of sub-tile t -> registers
    constexpr int i = decltype(ic)::value, s = decltype(setc)::value;
    const uint32_t kb = (uint32_t)(t * BKS * 2);
#if defined(COCODR_W4R_ABL_NOGLOAD)
    if (t >= 4) return;
#endif
    if constexpr (i < 3) st[s][i] = __builtin_amdgcn_raw_buffer_load_b128(ra, va, kb + i * pa, 0);
    else st[s][i] = __builtin_amdgcn_raw_buffer_load_b128(rb, vb, kb + (i - 3) * pb, 0);
  };
  auto lwrite = [&](auto ic, auto setc, uint32_t slot) {  // registers -> LDS slot byte offset `slot`
    constexpr int i = decltype(ic)::value, s = decltype(setc)::value;
    char* dst = smem + slot + (i < 3 ? wstA + i * 1024 : wstB + (i - 3) * 1024);
    *reinterpret_cast<v4i*>(dst) = st[s][i];
  };

  // fragment addresses: lane -> row (lane & 31) of its 32-row block, logical chunk 2 s + (lane >> 5); sub-step 1 flips bit 5
  uint32_t adA[2], adB[2];
  {
    const int ra_ = wr * 96 + (lane & 31), rb_ = wc * 128 + (lane & 31), h = lane >> 5;
    adA[0] = (uint32_t)(ra_ * 64 + ((h ^ ((ra_ >> 2) & 3)) << 4));
    adB[0] = (uint32_t)(OP_BYTES + rb_ * 64 + ((h ^ ((rb_ >> 2) & 3)) << 4));
    adA[1] = adA[0] ^ 32u;
    adB[1] = adB[0] ^ 32u;
  }
#if defined(COCODR_W4R_ABL_NOREAD)
  v4i fa[2][3] = {}, fb[2][4] = {};
#else
  v4i fa[2][3], fb[2][4];  // [fragment buffer][32-row block]
#endif
  auto fread = [&](auto ic, auto bufc, uint32_t aA, uint32_t aB) {
    constexpr int i = decltype(ic)::value, buf = decltype(bufc)::value;
#if defined(COCODR_W4R_ABL_NOREAD)
    return;
#endif
    if constexpr (i < 3) fa[buf][i] = *reinterpret_cast<const v4i*>(smem + aA + i * 2048);
    else fb[buf][i - 3] = *reinterpret_cast<const v4i*>(smem + aB + (i - 3) * 2048);
  };

  f32x16 acc[3][4];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  auto mfma_one = [&](auto ic, auto bufc) {
    constexpr int i = decltype(ic)::value, buf = decltype(bufc)::value, a = i >> 2, b = i & 3;
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[buf][b]), __builtin_bit_cast(bf16x8, fa[buf][a]), acc[a][b], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;

  // ---- prologue: sub-tiles 0 .. 3 requested (sets 0, 1, 2, 0: sub-tile 3 re-uses set 0 after sub-tile 0 has been written);
  // 0 and 1 written and certified, the fragments of (0, sub-step 0) read
  static_for<0, 7>([&](auto ic) { gload(ic, I0{}, 0); });
  static_for<0, 7>([&](auto ic) { gload(ic, I1{}, 1); });
  static_for<0, 7>([&](auto ic) { gload(ic, I2{}, 2); });
  static_for<0, 7>([&](auto ic) { lwrite(ic, I0{}, 0u); });
  static_for<0, 7>([&](auto ic) { gload(ic, I0{}, 3); });
  static_for<0, 7>([&](auto ic) { lwrite(ic, I1{}, (uint32_t)STAGE_BYTES); });
  __syncthreads();
  static_for<0, 7>([&](auto ic) { fread(ic, I0{}, adA[0], adB[0]); });

  // One sub-tile t.  SET2 = register set of sub-tile t + 2 (written here) - sub-tile t + 4 is loaded into set (t + 4) % 3 = (t + 1) % 3.
  // slot offsets: so = slot of t, sn = slot of t + 1, sw = slot of t + 2 (= slot of t - 1)
  uint32_t so = 0, sn = STAGE_BYTES, sw = 2 * STAGE_BYTES;
  // STEADY: sub-tile t + DIST exists (so t + 1, t + 2 do as well) - no condition (= no branch per MFMA slot) anywhere in the body
  auto subtile = [&](auto steady_c, auto set2c, auto set4c, const int t) {
    constexpr bool STEADY = decltype(steady_c)::value;
    const uint32_t a1 = adA[1] + so, b1 = adB[1] + so, a0n = adA[0] + sn, b0n = adB[0] + sn;
    const bool w2 = STEADY || t + 2 < nst, g4 = STEADY || t + DIST < nst, nx = STEADY || t + 1 < nst;
    // sub-step 0: 12 MFMAs on buffer 0; in their shadow the 7 fragment reads of sub-step 1, then sub-tile t + 2 goes to LDS (7 writes)
    static_for<0, 12>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      mfma_one(ic, I0{});
      if constexpr (i < 7) fread(ic, I1{}, a1, b1);
      if constexpr (i >= 5) { if (w2) lwrite(std::integral_constant<int, i - 5>{}, set2c, sw); }
      __builtin_amdgcn_sched_barrier(0);
    });
    // sub-step 1: 12 MFMAs on buffer 1; in their shadow the first fragments of sub-tile t + 1, then the 7 requests of sub-tile t + 4
    static_for<0, 12>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      mfma_one(ic, I1{});
      if constexpr (i < 7) { if (nx) fread(ic, I0{}, a0n, b0n); }
      if constexpr (i >= 5) { if (g4) gload(std::integral_constant<int, i - 5>{}, set4c, t + DIST); }
      __builtin_amdgcn_sched_barrier(0);
    });
#if !defined(COCODR_W4R_ABL_NOBARRIER)
    __syncthreads();  // sub-tile t + 2 is in LDS for everybody (and nobody reads slot t any more)
#endif
    const uint32_t o = so;
    so = sn; sn = sw; sw = o;
  };
  int t = 0;
  for (; t + 2 + DIST < nst; t += 3) {  // set of t + 2: (t + 2) % 3 with t % 3 == 0 -> 2, 0, 1; set of t + 4: 1, 2, 0
    subtile(std::true_type{}, I2{}, I1{}, t);
    subtile(std::true_type{}, I0{}, I2{}, t + 1);
    subtile(std::true_type{}, I1{}, I0{}, t + 2);
  }
  // the last 4 .. 6 sub-tiles (t is a multiple of 3 here): the general form, straight line
  if (t < nst) { subtile(std::false_type{}, I2{}, I1{}, t); ++t; }
  if (t < nst) { subtile(std::false_type{}, I0{}, I2{}, t); ++t; }
  if (t < nst) { subtile(std::false_type{}, I1{}, I0{}, t); ++t; }
  if (t < nst) { subtile(std::false_type{}, I2{}, I1{}, t); ++t; }
  if (t < nst) { subtile(std::false_type{}, I0{}, I2{}, t); ++t; }
  if (t < nst) { subtile(std::false_type{}, I1{}, I0{}, t); ++t; }

  // ---- epilogue: two 96-row passes of the fp32 tile through LDS, row-major 16-B stores
  const float* __restrict__ bias = p.bias;
  const uint16_t* __restrict__ R_ = p.R;
  constexpr int CPRW = BN / 8;                 // 8-column chunks per output row
  constexpr int RP = 96;
  constexpr int NCH = RP * CPRW / NTHREADS;    // 12 chunks per thread and pass, fixed columns
  const bool need_r = R_ != nullptr && (p.epi == COCODR_EPI_ADD || p.epi == COCODR_EPI_DGELU);
  float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr) {
    const float4 b0 = *reinterpret_cast<const float4*>(bias + n0 + ((tid % CPRW) << 3));
    const float4 b1 = *reinterpret_cast<const float4*>(bias + n0 + ((tid % CPRW) << 3) + 4);
    bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
  }
  float* ct = reinterpret_cast<float*>(smem);
  constexpr int CLD = CT_LD;
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    if (wr == h) {
#pragma unroll
      for (int ai = 0; ai < 3; ++ai)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int row = ai * 32 + (lane & 31);
            const int col = wc * 128 + b * 32 + 8 * rg + 4 * (lane >> 5);
            *reinterpret_cast<float4*>(ct + row * CLD + col) =
                make_float4(acc[ai][b][rg * 4 + 0], acc[ai][b][rg * 4 + 1], acc[ai][b][rg * 4 + 2], acc[ai][b][rg * 4 + 3]);
          }
    }
    __syncthreads();
#pragma unroll 4
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
        uint4 rr = make_uint4(0, 0, 0, 0);
        if (need_r) rr = *reinterpret_cast<const uint4*>(R_ + (size_t)gm * p.ldr + gn);
        epilogue_store8<OUT_F32, true, true>(p, 0, bias, R_, gm, gn, v, rr, bias8);
      }
    }
    if (h == 0) __syncthreads();
  }
#endif
}

}  // namespace cocodr_gemm_w4s

// experiment entry (tools/w4s_check.py): 0 = launched, 1 = shape not taken
extern "C" int w4s_gemm(const cocodr_gemm_args* pa, void* stream) {
  using namespace cocodr_gemm_w4s;
  const cocodr_gemm_args& a = *pa;
  if (a.trans_a || a.trans_b || a.N % 256 != 0 || a.K % 32 != 0 || a.K < 32 * 6 || a.batch > 1 || a.ab_f16 || a.colsum || a.colsum_partial) return 1;
  if ((size_t)a.M * a.lda * 2 >= (1ull << 32) || (size_t)a.N * a.ldb * 2 >= (1ull << 32)) return 1;
  const int ntm = (a.M + BM - 1) / BM, ntn = a.N / BN;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_w4s_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)gemm_w4s_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done = true;
  }
  if (a.out_f32) hipLaunchKernelGGL((gemm_w4s_kernel<true>), dim3(ntm * ntn), dim3(NTHREADS), LDS_BYTES, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((gemm_w4s_kernel<false>), dim3(ntm * ntn), dim3(NTHREADS), LDS_BYTES, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
