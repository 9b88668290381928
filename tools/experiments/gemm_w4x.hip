// (round 5: gemm_w4r.hip with v_mfma_f32_16x16x32_bf16 - 64 accumulator blocks of 16 x 16 per wave, one 32-deep step per sub-tile; see the bottom of
//  the loop comment)
// "One wave per SIMD, register-staged" bf16 MFMA GEMM for gfx950: 256 x 256 tile, FOUR waves (2 x 2), each owning 128 x 128 of the
// tile with its 16 accumulator blocks (256 registers) in the accumulator half of the unified 512-entry register file.
//
// Why (round 4).  The ping-pong pipeline (gemm_pp.hip) is LDS-bandwidth bound: eight waves of 128 x 64 read 192 KB of fragments
// per 64-deep K-tile and CU and the operand DMA writes 64 KB, 2048 LDS cycles against 2048 matrix-pipe cycles.  Four waves of
// 128 x 128 read 128 KB - a third less - but such a wave has no partner on its SIMD, so its own instruction stream has to carry
// the loads in the shadow of its MFMAs.  Round 3 tried that with LDS-DMA operand loads (tools/experiments/gemm_w4.hip) and lost:
// removing the DMA alone bought 23 % - a `buffer_load ... lds` needs M0 rewritten in front of every instruction and holds the
// wave until the address unit has taken it, and with one wave per SIMD nobody else feeds the matrix pipe meanwhile.  Here the
// operands go global -> registers -> LDS ("issue early, write late"): an ordinary `buffer_load_dwordx4` into VGPRs is fire and
// forget, the matching `ds_write_b128` follows two sub-tiles later when the data has long arrived, and because nothing LDS-bound
// is in flight from the memory side every access is visible to hipcc, which counts vmcnt / lgkmcnt itself (no hand-counted waits).
//
// Pipeline per 32-deep sub-tile t (one iteration = 32 MFMAs per wave, one memory instruction in the shadow of each):
//     sub-step 0 (16 MFMAs, fragment buffer 0):  8 fragment reads of (t, sub-step 1)  |  8 ds_write of sub-tile t + 2
//     sub-step 1 (16 MFMAs, fragment buffer 1):  8 fragment reads of (t + 1, sub-step 0)  |  8 global loads of sub-tile t + 4
//     barrier: sub-tile t + 2 is in LDS for everybody
// LDS ring: three stages of [256 rows][32 k] A + B (32 KiB each).  Sub-tile s is requested in iteration s - 4, written in s - 2
// (slot s % 3, last read in iteration s - 3: ordered by that iteration's barrier), certified by the barrier that ends s - 2, its
// first fragments are read under the last MFMAs of s - 1 and it is consumed in s - so no wave ever waits for a barrier followed
// by an LDS round trip; the barrier only absorbs the skew between the four waves.
//
// Forms / epilogues: the NT form (forward GEMMs) with every bf16 / fp32 epilogue of gemm_tile.h through the two-pass LDS epilogue
// of gemm_pp.hip.  Reached as impl 14 (cocodr_gemm_set_impl) and from the selection in gemm.hip where it measured ahead.
#include <stdlib.h>

#include "common.h"
#include "gemm_tile.h"

namespace cocodr_gemm_w4x {
using namespace cocodr_gemm_v2;

constexpr int BM = 256, BN = 256, BKS = 32;
constexpr int NSLOT = 3;
constexpr int OP_BYTES = 256 * BKS * 2;            // 16 KiB: [256 rows][32 k] bf16
constexpr int STAGE_BYTES = 2 * OP_BYTES;          // A then B
constexpr int RING_BYTES = NSLOT * STAGE_BYTES;    // 96 KiB
constexpr int CT_LD = BN + 4;                      // fp32 epilogue tile leading dimension
constexpr int EPI_BYTES = 128 * CT_LD * 4 + 4 * BN * 4;
constexpr int LDS_BYTES = EPI_BYTES > RING_BYTES ? EPI_BYTES : RING_BYTES;
constexpr int NTHREADS = 256;
constexpr int DIST = 4;                            // global loads run this many sub-tiles ahead of the MFMAs
constexpr int NSET = 3;                            // register sets of staged operands (sub-tiles t + 2 .. t + 4)

template <bool OUT_F32>
__global__ __launch_bounds__(NTHREADS, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w4x_kernel(const cocodr_gemm_args p) {
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

  // staging: piece q (0..15) of an operand stage = rows 16 q .. + 15 (1 KiB); wave w moves pieces 4 w .. 4 w + 3 of A and of B.
  // lane -> row 16 q + (lane >> 2), stored chunk lane & 3 = logical chunk (lane & 3) ^ ((row >> 2) & 3) (rows past M / N read 0)
  // 16 x 16 x 32 fragments: a ds_read_b128 covers rows r .. r + 15 x the four 16-B chunks of a 64-B row; with the stored chunk =
  // logical chunk ^ ((-(row >> 2)) & 3) every 16-lane group of the b128 access hits 16 distinct slots of a 256-B bank row
  const int prow = lane >> 2, pch = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
  const uint32_t va = (uint32_t)(((m0 + wid * 64 + prow) * p.lda + pch * 8) * 2);
  const uint32_t vb = (uint32_t)(((n0 + wid * 64 + prow) * p.ldb + pch * 8) * 2);
  const uint32_t pa = (uint32_t)(16 * p.lda * 2), pb = (uint32_t)(16 * p.ldb * 2);
  const int nst = p.K / BKS;  // sub-tiles (the caller guarantees K % 32 == 0, K >= 32 * (DIST + 1))
  const uint32_t wst = (uint32_t)(wid * 4096 + lane * 16);  // this lane's 16 bytes inside a wave's four 1-KiB pieces

  v4i st[NSET][8];  // staged operand chunks: [set][A pieces 0-3 | B pieces 0-3]
  auto gload = [&](auto ic, auto setc, int t) {  // piece i of sub-tile t -> registers
    constexpr int i = decltype(ic)::value, s = decltype(setc)::value;
    const uint32_t kb = (uint32_t)(t * BKS * 2);
#if defined(COCODR_W4R_ABL_NOGLOAD)
    if (t >= 4) return;
#endif
    if constexpr (i < 4) st[s][i] = __builtin_amdgcn_raw_buffer_load_b128(ra, va, kb + i * pa, 0);
    else st[s][i] = __builtin_amdgcn_raw_buffer_load_b128(rb, vb, kb + (i - 4) * pb, 0);
  };
  auto lwrite = [&](auto ic, auto setc, uint32_t slot) {  // registers -> LDS slot byte offset `slot`
    constexpr int i = decltype(ic)::value, s = decltype(setc)::value;
    char* dst = smem + slot + (i < 4 ? 0 : OP_BYTES) + wst + (i & 3) * 1024;
#if defined(COCODR_W4R_ABL_NOWRITE)  // (timing ablations, tools/w4r_ablate.sh: results are wrong)
    asm volatile("" ::"v"(st[s][i]), "v"(dst));
#else
    *reinterpret_cast<v4i*>(dst) = st[s][i];
#endif
  };

  // fragment addresses: lane -> row (lane & 15) of its 16-row block, logical chunk lane >> 4; block i adds 16 rows = 1024 bytes
  uint32_t adA, adB;
  {
    const int ra_ = wr * 128 + (lane & 15), rb_ = wc * 128 + (lane & 15), ch = lane >> 4;
    adA = (uint32_t)(ra_ * 64 + ((ch ^ ((0 - (ra_ >> 2)) & 3)) << 4));
    adB = (uint32_t)(OP_BYTES + rb_ * 64 + ((ch ^ ((0 - (rb_ >> 2)) & 3)) << 4));
  }
  v4i fa[4], fb[2][8];  // A: four single fragments in rotation, read TWO block rows ahead (one row = 8 MFMAs = 128 clocks is less than an LDS
                        // round trip under load: with one row of distance every row started behind an lgkmcnt stall); B: 8 fragments, double-buffered
  typedef float f32x4a __attribute__((ext_vector_type(4)));
  f32x4a acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  auto read_a = [&](auto pc, uint32_t slot, int blk) {  // (blk is a compile-time constant at every call site)
    constexpr int pp_ = decltype(pc)::value;
    fa[pp_] = *reinterpret_cast<const v4i*>(smem + adA + slot + blk * 1024);
  };
  auto read_b = [&](auto bufc, auto jc, uint32_t slot) {
    constexpr int buf = decltype(bufc)::value, j = decltype(jc)::value;
    fb[buf][j] = *reinterpret_cast<const v4i*>(smem + adB + slot + j * 1024);
  };

  // ---- prologue: sub-tiles 0 .. 3 requested (sets 0, 1, 2, 0), 0 and 1 written and certified, the fragments of sub-tile 0 read
  static_for<0, 8>([&](auto ic) { gload(ic, I0{}, 0); });
  static_for<0, 8>([&](auto ic) { gload(ic, I1{}, 1); });
  static_for<0, 8>([&](auto ic) { gload(ic, I2{}, 2); });
  static_for<0, 8>([&](auto ic) { lwrite(ic, I0{}, 0u); });
  static_for<0, 8>([&](auto ic) { gload(ic, I0{}, 3); });
  static_for<0, 8>([&](auto ic) { lwrite(ic, I1{}, (uint32_t)STAGE_BYTES); });
  __syncthreads();
  static_for<0, 8>([&](auto jc) { read_b(I0{}, jc, 0u); });
  read_a(I0{}, 0u, 0);
  read_a(I1{}, 0u, 1);

  // One sub-tile t = 64 MFMAs (block row i of A x the 8 B fragments), one memory instruction in the shadow of every second MFMA:
  //   A fragment of block row i + 1 (of sub-tile t + 1's row 0 behind the last row) right behind the first MFMA of row i;
  //   queue of 24: 8 B fragments of sub-tile t + 1 (other fragment buffer), 8 ds_write of sub-tile t + 2, 8 global loads of t + 4.
  uint32_t so = 0, sn = STAGE_BYTES, sw = 2 * STAGE_BYTES;
  auto subtile = [&](auto steady_c, auto set2c, auto set4c, auto bufc, const int t) {
    constexpr bool STEADY = decltype(steady_c)::value;
    constexpr int cur = decltype(bufc)::value, nxt = cur ^ 1;
    const bool w2 = STEADY || t + 2 < nst, g4 = STEADY || t + DIST < nst, nx = STEADY || t + 1 < nst;
    static_for<0, 8>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      static_for<0, 8>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        // inline asm with the accumulator pinned to the AGPR file: through the builtin hipcc keeps part of the 64 blocks in VGPRs and
        // shuttles them with v_accvgpr_write / read + s_nop around the MFMAs
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fb[cur][j]), "v"(fa[i & 3]));
        if constexpr (j == 0) {  // the fragment of block row i + 2 (rows 0 / 1 of sub-tile t + 1 behind rows 6 / 7)
          if constexpr (i < 6) read_a(std::integral_constant<int, (i + 2) & 3>{}, so, i + 2);
          else { if (nx) read_a(std::integral_constant<int, (i + 2) & 3>{}, sn, i - 6); }
        }
        if constexpr ((j & 1) == 1) {
          constexpr int q = i * 4 + (j >> 1);  // 0 .. 31
          if constexpr (q < 8) { if (nx) read_b(std::integral_constant<int, nxt>{}, std::integral_constant<int, q>{}, sn); }
          else if constexpr (q < 16) { if (w2) lwrite(std::integral_constant<int, q - 8>{}, set2c, sw); }
          else if constexpr (q < 24) { if (g4) gload(std::integral_constant<int, q - 16>{}, set4c, t + DIST); }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
#if !defined(COCODR_W4R_ABL_NOBARRIER)
    __syncthreads();  // sub-tile t + 2 is in LDS for everybody (and nobody reads slot t any more)
#endif
    const uint32_t o = so;
    so = sn; sn = sw; sw = o;
  };
  // register set of t + 2: (t + 2) % 3, of t + 4: (t + 1) % 3; fragment buffer t % 2: period 6
  int t = 0;
  for (; t + 5 + DIST < nst; t += 6) {
    subtile(std::true_type{}, I2{}, I1{}, I0{}, t);
    subtile(std::true_type{}, I0{}, I2{}, I1{}, t + 1);
    subtile(std::true_type{}, I1{}, I0{}, I0{}, t + 2);
    subtile(std::true_type{}, I2{}, I1{}, I1{}, t + 3);
    subtile(std::true_type{}, I0{}, I2{}, I0{}, t + 4);
    subtile(std::true_type{}, I1{}, I0{}, I1{}, t + 5);
  }
  // the last sub-tiles (t is a multiple of 6 here): the general form, straight line
  for (int rep = 0; rep < 2; ++rep) {
    if (t < nst) { subtile(std::false_type{}, I2{}, I1{}, I0{}, t); ++t; }
    if (t < nst) { subtile(std::false_type{}, I0{}, I2{}, I1{}, t); ++t; }
    if (t < nst) { subtile(std::false_type{}, I1{}, I0{}, I0{}, t); ++t; }
    if (t < nst) { subtile(std::false_type{}, I2{}, I1{}, I1{}, t); ++t; }
    if (t < nst) { subtile(std::false_type{}, I0{}, I2{}, I0{}, t); ++t; }
    if (t < nst) { subtile(std::false_type{}, I1{}, I0{}, I1{}, t); ++t; }
  }

  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (the asm MFMAs are invisible to hipcc's hazard recognizer: let the last ones retire)
  // ---- epilogue (gemm_pp.hip's): two 128-row passes of the fp32 tile through LDS, row-major 16-B stores
  const float* __restrict__ bias = p.bias;
  const uint16_t* __restrict__ R_ = p.R;
  constexpr int CPRW = BN / 8;                 // 8-column chunks per output row
  constexpr int RP = 128;
  constexpr int NCH = RP * CPRW / NTHREADS;    // 16 chunks per thread and pass, fixed columns
  const bool need_r = R_ != nullptr && (p.epi == COCODR_EPI_ADD || p.epi == COCODR_EPI_DGELU);
  float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr) {
    const float4 b0 = *reinterpret_cast<const float4*>(bias + n0 + ((tid % CPRW) << 3));
    const float4 b1 = *reinterpret_cast<const float4*>(bias + n0 + ((tid % CPRW) << 3) + 4);
    bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
  }
  float* ct = reinterpret_cast<float*>(smem);
  constexpr int CLD = CT_LD;
  const bool do_colsum = p.colsum_partial != nullptr;
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    if (wr == h) {
#pragma unroll
      for (int ai = 0; ai < 8; ++ai)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const int row = ai * 16 + (lane & 15);
          const int col = wc * 128 + b * 16 + 4 * (lane >> 4);
          *reinterpret_cast<float4*>(ct + row * CLD + col) = make_float4(acc[ai][b][0], acc[ai][b][1], acc[ai][b][2], acc[ai][b][3]);
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
        if (do_colsum) {
#pragma unroll
          for (int j = 0; j < 8; ++j) csum[j] += v[j];
        }
      }
    }
    if (h == 0) __syncthreads();
  }
  if (do_colsum) {  // workgroup-uniform: lanes that differ by a multiple of CPRW = 32 hold the same 8 columns
#pragma unroll
    for (int j = 0; j < 8; ++j) csum[j] += __shfl_xor(csum[j], 32, 64);
    float* cred = ct + RP * CLD;
    __syncthreads();
    if (lane < CPRW) {
#pragma unroll
      for (int j = 0; j < 8; ++j) cred[wid * BN + lane * 8 + j] = csum[j];
    }
    __syncthreads();
    if (tid < BN) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) s += cred[w * BN + tid];
      p.colsum_partial[(size_t)tm_ * p.N + n0 + tid] = s;
    }
  }
#endif
}

}  // namespace cocodr_gemm_w4x

extern "C" int w4s_gemm(const cocodr_gemm_args* pa, void* stream) {  // (the name tools/w4s_check.py binds; W4LIB=libw4x.so)
  using namespace cocodr_gemm_w4x;
  const cocodr_gemm_args& a = *pa;
  if (a.trans_a || a.trans_b || a.N % 256 != 0 || a.K % 32 != 0 || a.K < 32 * 12 || a.batch > 1 || a.ab_f16 || a.colsum || a.colsum_partial) return 1;
  if ((size_t)a.M * a.lda * 2 >= (1ull << 32) || (size_t)a.N * a.ldb * 2 >= (1ull << 32)) return 1;
  const int ntm = (a.M + BM - 1) / BM, ntn = a.N / BN;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_w4x_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)gemm_w4x_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done = true;
  }
  if (a.out_f32) hipLaunchKernelGGL((gemm_w4x_kernel<true>), dim3(ntm * ntn), dim3(NTHREADS), LDS_BYTES, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((gemm_w4x_kernel<false>), dim3(ntm * ntn), dim3(NTHREADS), LDS_BYTES, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
