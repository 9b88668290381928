// bf16 MFMA GEMM for gfx950 with fused epilogues - the dense contractions of the BERT encoder
// (hf: nn.Linear in BertSelfAttention / BertSelfOutput / BertIntermediate / BertOutput) and of
// their backward passes.  One kernel template covers
//     forward  Y  = X  W^T   (A [M,K], B [N,K]   : trans_a=0, trans_b=0)
//     dgrad    dX = dY W     (A [M,K], B [K,N]   : trans_a=0, trans_b=1)
//     wgrad    dW = dY^T X   (A [K,M], B [K,N]   : trans_a=1, trans_b=1), batched over layers.
//
// Tile: 128x128x64 per 256-thread workgroup (4 waves, 2x2, each wave 64x64 = 2x2 MFMA 32x32x16).
// Operand tiles are staged HBM -> registers -> LDS (double buffered; the loads of tile t+1 are in
// flight while tile t is multiplied, the LDS write lands after the MFMAs).  Operands whose
// contraction index is the slow memory axis are kept row-major in LDS and read with the gfx950
// transposing LDS read (ds_read_b64_tr_b16), so no transposed copy of weights or activations is
// ever materialised in HBM.  The accumulators go through an fp32 LDS tile so that the epilogue
// (bias, erf-GELU, residual, GELU') runs row-major with 16-byte coalesced loads/stores.
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "prof.h"
#include "gemm_tile.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB per operand tile
constexpr int CT_LD = 132;                // fp32 epilogue tile leading dim (floats)

// ---- global -> registers (4 x 16 B per thread per operand tile), zero filled out of range
template <int TR>
__device__ __forceinline__ void load_tile(const uint16_t* __restrict__ G, int ld, int r0, int rlim, int k0, int klim,
                                          int tid, uint4 (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = tid + 256 * i;
    int gr, gk;
    if (TR == 0) {  // [rows][k] : 8 chunks per 64-wide k row
      gr = r0 + (q >> 3);
      gk = k0 + ((q & 7) << 3);
      v[i] = (gr < rlim && gk < klim) ? *reinterpret_cast<const uint4*>(G + (size_t)gr * ld + gk) : make_uint4(0, 0, 0, 0);
    } else {  // stored [k][rows] : 16 chunks per 128-wide row
      gk = k0 + (q >> 4);
      gr = r0 + ((q & 15) << 3);
      v[i] = (gr < rlim && gk < klim) ? *reinterpret_cast<const uint4*>(G + (size_t)gk * ld + gr) : make_uint4(0, 0, 0, 0);
    }
  }
}

template <int TR>
__device__ __forceinline__ void store_tile(char* lds, int tid, const uint4 (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = tid + 256 * i;
    const int off = (TR == 0) ? tile64_off(q >> 3, q & 7) : tile128_off(q >> 4, q & 15);
    *reinterpret_cast<uint4*>(lds + off) = v[i];
  }
}

// ---- LDS -> MFMA 32x32x16 operand fragment for the 32 rows starting at r0, k-step s (16 wide)
// lane l supplies row r0 + (l & 31) and the 8 contraction slots 16*s + 8*(l >> 5) + 0..7.
template <int TR>
__device__ __forceinline__ bf16x8 read_frag(const char* lds, int r0, int s, int lane) {
  if (TR == 0) {
    return lds_read_b128(lds, tile64_off(r0 + (lane & 31), 2 * s + (lane >> 5)));
  } else {
    const int g = lane >> 4, c = lane & 15;
    const int col = r0 + ((g & 1) << 4) + ((c & 3) << 2);
    const int row = 16 * s + ((g >> 1) << 3) + (c >> 2);
    const int off = tile128_off(row, col >> 3) + ((col & 7) << 1);
    return join_tr(lds_read_tr16(lds, off), lds_read_tr16(lds, off + 4 * 256));
  }
}

template <int TA, int TB, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const cocodr_gemm_args p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int ntn = p.N / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  const int z = blockIdx.y;
  const uint16_t* __restrict__ A = p.A + (size_t)z * p.strideA;
  const uint16_t* __restrict__ B = p.B + (size_t)z * p.strideB;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nt = (p.K + BK - 1) / BK;
  uint4 ra[4], rb[4];
  load_tile<TA>(A, p.lda, m0, p.M, 0, p.K, tid, ra);
  load_tile<TB>(B, p.ldb, n0, p.N, 0, p.K, tid, rb);
  store_tile<TA>(smem, tid, ra);
  store_tile<TB>(smem + TILE_BYTES, tid, rb);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const char* bufA = smem + (t & 1) * 2 * TILE_BYTES;
    const char* bufB = bufA + TILE_BYTES;
    const bool more = (t + 1 < nt);
    if (more) {
      load_tile<TA>(A, p.lda, m0, p.M, (t + 1) * BK, p.K, tid, ra);
      load_tile<TB>(B, p.ldb, n0, p.N, (t + 1) * BK, p.K, tid, rb);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = read_frag<TA>(bufA, wm * 64 + a * 32, s, lane);
#pragma unroll
      for (int b = 0; b < 2; ++b) fb[b] = read_frag<TB>(bufB, wn * 64 + b * 32, s, lane);
      // operands swapped: D[i = n][j = m], so a lane ends up with 4 consecutive n of one m
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b], fa[a], acc[a][b], 0, 0, 0);
    }
    if (more) {
      char* nb = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
      store_tile<TA>(nb, tid, ra);
      store_tile<TB>(nb + TILE_BYTES, tid, rb);
    }
    __syncthreads();
  }

  // ---- epilogue through an fp32 LDS tile, two 64-row halves
  float* ct = reinterpret_cast<float*>(smem);
  const float* __restrict__ bias = p.bias ? p.bias + (size_t)z * p.strideBias : nullptr;
  const uint16_t* __restrict__ R = p.R ? p.R + (size_t)z * p.strideR : nullptr;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (wm == h) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int row = a * 32 + (lane & 31);
            const int col = wn * 64 + b * 32 + 8 * rg + 4 * (lane >> 5);
            *reinterpret_cast<float4*>(ct + row * CT_LD + col) =
                make_float4(acc[a][b][rg * 4 + 0], acc[a][b][rg * 4 + 1], acc[a][b][rg * 4 + 2], acc[a][b][rg * 4 + 3]);
          }
    }
    __syncthreads();
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int row = (tid >> 4) + 16 * pp;
      const int gm = m0 + h * 64 + row;
      const int gn = n0 + ((tid & 15) << 3);
      if (gm < p.M) {
        float v[8];
        const float4 c0 = *reinterpret_cast<const float4*>(ct + row * CT_LD + ((tid & 15) << 3));
        const float4 c1 = *reinterpret_cast<const float4*>(ct + row * CT_LD + ((tid & 15) << 3) + 4);
        v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w; v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
        if (bias) {
          const float4 b0 = *reinterpret_cast<const float4*>(bias + gn);
          const float4 b1 = *reinterpret_cast<const float4*>(bias + gn + 4);
          v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (p.epi == COCODR_EPI_GELU && p.C2 == nullptr) {  // inference: nobody needs the derivative
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = gelu_erf(v[j]);
        } else if (p.epi == COCODR_EPI_GELU) {
          float gp[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) gelu_erf_both(v[j], v[j], gp[j]);
          *reinterpret_cast<uint4*>(p.C2 + (size_t)z * p.strideC + (size_t)gm * p.ldc + gn) = pack8(gp);
        } else if (p.epi == COCODR_EPI_ADD) {
          if (p.drop.threshold) drop_apply<8>(v, (uint64_t)gm * p.N + gn, p.drop);
          float r[8];
          unpack8(*reinterpret_cast<const uint4*>(R + (size_t)gm * p.ldr + gn), r);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += r[j];
        } else if (p.epi == COCODR_EPI_DGELU) {
          float r[8];
          unpack8(*reinterpret_cast<const uint4*>(R + (size_t)gm * p.ldr + gn), r);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] *= r[j];
        }
        if (OUT_F32) {
          float* C = reinterpret_cast<float*>(p.C) + (size_t)z * p.strideC + (size_t)gm * p.ldc + gn;
          *reinterpret_cast<float4*>(C) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(C + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          uint16_t* C = reinterpret_cast<uint16_t*>(p.C) + (size_t)z * p.strideC + (size_t)gm * p.ldc + gn;
          *reinterpret_cast<uint4*>(C) = pack8(v);
        }
      }
    }
    __syncthreads();
  }
}


}  // namespace

// =====================================================================================================
// v2: direct-to-LDS pipeline.  Operand tiles go HBM/L2 -> LDS with buffer_load_dwordx4 ... lds (no VGPR
// round trip) into a 3-deep ring, two tiles in flight behind a counted s_waitcnt vmcnt(N) and ONE raw
// s_barrier per K-step.  The LDS image is lane-linear per wave instruction, so the XOR swizzle the
// fragment reads need is applied to the per-lane SOURCE address (the permutation is an involution).
// Out-of-range rows come back as zeros through the buffer descriptor's bounds check (one descriptor
// per batch item), which is what makes ragged token counts legal in the contraction dimension.
//
// Geometry <BM, BK, WTM>: workgroup tile BM x 128 x BK; a wave owns (32*WTM) x 64 outputs = WTM x 2 MFMA
// 32x32x16 tiles, so the workgroup has (BM / (32*WTM)) x 2 waves.
//   <256,64,2>: 8 waves, 144 KiB LDS, 1 workgroup / CU     <256,32,2>: 8 waves, 72 KiB, 2 workgroups / CU
//   <128,64,2>: 4 waves,  96 KiB LDS, 1 workgroup / CU     <128,32,2>: 4 waves, 48 KiB, 3 workgroups / CU
//   <256,32,4>: 4 waves of 128x64 (6 LDS fragment reads per 8 MFMAs instead of 4 per 4), 72 KiB, 2 / CU
// With >= 2 workgroups per CU one workgroup's prologue / epilogue / LDS phase overlaps the other's MFMAs.
//
// LDS fragment reads are issued from inline asm: hipcc drains every in-flight LDS-DMA
// (s_waitcnt vmcnt(0)) in front of a ds_read_b64_tr_b16 it can see, which serialises the ring for the
// dgrad / wgrad forms.  Completion is counted by hand (LDS reads return in order): the reads of K-sub-step
// s+1 are issued before the MFMAs of s and `s_waitcnt lgkmcnt(R)` (R = reads per sub-step) retires s.
// Register moves that assemble a fragment sit after the wait.
// =====================================================================================================
// (named namespace: hipFuncSetAttribute takes the kernels' addresses, which needs external linkage)
namespace cocodr_gemm_v2 {

template <int BMv, int BKv, int WTM, int WTN, int LD = 0, int WC = 2, int NS = 3>
struct Geom {
  static constexpr int BNv = 32 * WTN * WC;  // WC wave columns of 32*WTN
  static constexpr int NWAVES = (BMv / (32 * WTM)) * WC;  // MFMA waves
  static constexpr int NLOAD = LD;                       // dedicated loader waves (0: the MFMA waves issue the DMA themselves)
  static constexpr int NISSUE = LD ? LD : NWAVES;        // waves that issue DMA pieces
  static constexpr int NTHREADS = (NWAVES + NLOAD) * 64;
  static constexpr int CTHREADS = NWAVES * 64;
  static constexpr int A_BYTES = BMv * BKv * 2;
  static constexpr int B_BYTES = BNv * BKv * 2;
  static constexpr int CT_LDv = BNv + 4;  // fp32 epilogue tile leading dim
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int GA = (A_BYTES / 1024) / NISSUE;  // wave-level 1 KiB loads per stage and issuing wave
  static constexpr int GB = (B_BYTES / 1024) / NISSUE;
  static_assert(GA * NISSUE * 1024 == A_BYTES && GB * NISSUE * 1024 == B_BYTES, "operand tiles must split evenly over the issuing waves");
  static constexpr int NSTAGE = NS;  // ring depth: NS - 1 stages are in flight or landed ahead of the one being read.  Six 24-KiB
                                     // stages (BK = 32) instead of three 48-KiB ones measured 10-20 % SLOWER on every encoder shape
                                     // (profiles/archive/r01k_gemm_ring_depth.txt): a K-step costs ~0.17 us of hand-off however short it is
  static constexpr int KS = BKv / 16;  // MFMA K-sub-steps per stage
  static constexpr int WG_PER_CU = (160 * 1024) / (NSTAGE * STAGE);
  static constexpr int MIN_WAVES_PER_SIMD = (WG_PER_CU * (NWAVES + NLOAD) + 3) / 4;
  // epilogue: rows of the fp32 tile that fit the ring's LDS, rounded down to a power-of-two multiple of a wave's rows
  static constexpr int EPI_FIT = (NSTAGE * STAGE) / (CT_LDv * 4);
  static constexpr int EPI_ROWS = EPI_FIT >= BMv ? BMv : (EPI_FIT >= BMv / 2 ? BMv / 2 : BMv / 4);
  static_assert(EPI_ROWS * CT_LDv * 4 + NWAVES * BNv * 4 <= NSTAGE * STAGE, "no room for the column-sum row behind the epilogue tile");
};

template <int BMv, int BKv, int WTM, int WTN, int LD, int TA, int TB, bool OUT_F32, int WC = 2, int NS = 3>
__global__ __launch_bounds__((Geom<BMv, BKv, WTM, WTN, LD, WC, NS>::NTHREADS), (Geom<BMv, BKv, WTM, WTN, LD, WC, NS>::MIN_WAVES_PER_SIMD)) void gemm_glds_kernel(
    const cocodr_gemm_args p, const int stagger, const int flat) {
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-resource type only exists in the device pass; the host pass just needs the stub
  using G = Geom<BMv, BKv, WTM, WTN, LD, WC, NS>;
  constexpr int BNv = G::BNv;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WC, wn = wid % WC;
  const int ntn = p.N / BNv, ntm = (p.M + BMv - 1) / BMv;
  // flat (batched launches): one grid axis over (batch item, tile), item-major, XCD remap over all of it, so the tiles an
  // XCD holds at a time belong to one or two items instead of a few tiles of each of many (see gemm_pp.hip)
  int tile, z;
  if (flat) {
    const int per = ntm * ntn;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    z = id / per;
    tile = id - z * per;
  } else {
    tile = xcd_remap(blockIdx.x, gridDim.x);
    z = blockIdx.y;
  }
  int tm_, tn_;
  if (TA == 0) grouped_tile(tile, ntm, ntn, 1024 / BMv, tm_, tn_);  // measured: helps fwd/dgrad, hurts the long-K wgrad
  else if (flat) grouped_tile(tile, ntm, ntn, 8, tm_, tn_);
  else { tm_ = tile / ntn; tn_ = tile % ntn; }
  const int m0 = tm_ * BMv, n0 = tn_ * BNv;
  const uint16_t* A = p.A + (size_t)z * p.strideA;
  const uint16_t* B = p.B + (size_t)z * p.strideB;
  const uint32_t a_bytes = (uint32_t)((size_t)(TA ? p.K : p.M) * p.lda * 2);
  const uint32_t b_bytes = (uint32_t)((size_t)(TB ? p.K : p.N) * p.ldb * 2);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);

  // LD > 0: the last LD waves of the workgroup only issue the operand DMA (one per SIMD next to two MFMA waves).  A
  // buffer_load ... lds has to wait for its turn in the CU's address unit, which at this tile shape is busy for most of a
  // K-step; a wave that also carries MFMAs is stuck behind its own loads meanwhile (in-order issue), a loader wave is not.
  constexpr bool SELF_ISSUE = (LD == 0);
  const bool is_loader = LD > 0 && wid >= G::NWAVES;
  const int iw = LD > 0 ? wid - G::NWAVES : wid;  // index among the issuing waves
  // per-lane source offsets of this wave's loads (tile 0); advancing one K-step adds a constant
  uint32_t offa[G::GA], offb[G::GB];
#pragma unroll
  for (int j = 0; j < G::GA; ++j) offa[j] = glds_src_off<TA, BMv, BKv>((iw * G::GA + j) * 64 + lane, m0, p.lda);
#pragma unroll
  for (int j = 0; j < G::GB; ++j) offb[j] = glds_src_off<TB, BNv, BKv>((iw * G::GB + j) * 64 + lane, n0, p.ldb);
  const uint32_t stepa = TA ? (uint32_t)(BKv * p.lda * 2) : (uint32_t)(BKv * 2);
  const uint32_t stepb = TB ? (uint32_t)(BKv * p.ldb * 2) : (uint32_t)(BKv * 2);

  // De-phasing of the workgroups that share a CU: all first-round workgroups start together, run the same number of
  // K-steps and would reach their store-bound epilogues (and the next prologues) at the same moment, every round, with
  // the MFMA pipes idle meanwhile.  The first-round workgroup that was given the upper part of the CU's LDS sleeps for
  // `stagger` x 4096 clocks once; its successors inherit the offset, so one workgroup's epilogue / prologue runs under
  // the other's main loop from then on.
  uint32_t lds_slot_base = 0;
  if (G::WG_PER_CU >= 2 && stagger > 0 && (int)(blockIdx.y * gridDim.x + blockIdx.x) < 256 * G::WG_PER_CU) {
    lds_slot_base = __builtin_amdgcn_s_getreg(6 | (11 << 11));  // HW_REG_LDS_ALLOC bits [11:0]: LDS_BASE
    if (lds_slot_base != 0)
      for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(64);
  }
#if defined(COCODR_ABL_TIMELINE)  // per-workgroup phase stamps (100 MHz wall clock) into C2: [start, loop entry, loop exit, end, LDS base]
  unsigned long long* tl = reinterpret_cast<unsigned long long*>(p.C2) + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8;
  if (tid == 0) { tl[0] = wall_clock64(); tl[4] = __builtin_amdgcn_s_getreg(6 | (31 << 11)); }
#endif
  // one 1-KiB wave-level piece of K-step t's operand tiles (pieces 0..GA-1 belong to A, the rest to B)
  auto issue_piece = [&](auto jc, int t) {
    constexpr int j = decltype(jc)::value;
    char* st = smem + (t % G::NSTAGE) * G::STAGE;
    if constexpr (j < G::GA)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (LDS_PTR(void))(st + (iw * G::GA + j) * 1024), 16, offa[j] + t * stepa, 0, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (LDS_PTR(void))(st + G::A_BYTES + (iw * G::GB + (j - G::GA)) * 1024), 16,
                                               offb[j - G::GA] + t * stepb, 0, 0, 0);
  };
  constexpr int NP = G::GA + G::GB;

  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int a = 0; a < WTM; ++a)
#pragma unroll
    for (int b = 0; b < WTN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const uint32_t lds_base = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;
  uint32_t adA[4], adB[4];
  frag_addrs<TA, BMv, BKv, WTM>(wm * 32 * WTM, lane, adA);
  frag_addrs<TB, BNv, BKv, WTN>(wn * 32 * WTN, lane, adB);
  constexpr int R = (TA ? 2 : 1) * WTM + (TB ? 2 : 1) * WTN;  // LDS reads per K-sub-step (<= 12 < the 4-bit lgkmcnt range)

  // Schedule of one K-step (KS sub-steps of 16): the fragment reads of sub-step s+1 and a share of the NEXT-but-one
  // stage's DMA pieces are issued in front of the MFMAs of sub-step s, so a wave's DMA issue slots and LDS latency sit
  // under MFMAs already in the pipe.  The stage hand-off (own pieces landed -> barrier) is taken BEFORE the last
  // sub-step's MFMAs, and the next stage's first fragments are requested right behind it: the barrier wait and that
  // read latency are covered by the MFMAs still executing, instead of opening a bubble at every K-step boundary.
  const int nt = (p.K + BKv - 1) / BKv;
  constexpr int D = G::NSTAGE - 1;  // prefetch distance: stage t + D is requested once stage t - 1 has been released
  if (SELF_ISSUE || is_loader) {
    static_for<0, D>([&](auto dc) {
      constexpr int d = decltype(dc)::value;
      if (d < nt) static_for<0, NP>([&](auto jc) { issue_piece(jc, d); });
    });
    wait_vmcnt_stages<NP, D - 1>(min(nt, D) - 1);
  }
  __builtin_amdgcn_s_barrier();  // barrier 0: stage 0 complete
  if (is_loader) {
    // loader wave: after barrier t every MFMA wave has finished reading stage t-1 -> refill its slot with stage t+D, then
    // hand over stage t+1 (own pieces landed) at barrier t+1.  Same barrier sequence as the MFMA waves below.
    for (int t = 0; t < nt; ++t) {
      if (t + D < nt) static_for<0, NP>([&](auto jc) { issue_piece(jc, t + D); });
      if (t + 1 < nt) {
        wait_vmcnt_stages<NP, D - 1>(min(nt - 1, t + D) - (t + 1));  // stages requested beyond t + 1 may stay in flight
        __builtin_amdgcn_s_barrier();
      }
    }
  } else {
#if defined(COCODR_ABL_NO_LDSREAD)
  FragSet<TA, WTM> fa0{}, fa1{};
  FragSet<TB, WTN> fb0{}, fb1{};
#else
  FragSet<TA, WTM> fa0, fa1;
  FragSet<TB, WTN> fb0, fb1;
#endif
  uint32_t curA[4], curB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { curA[i] = adA[i] + lds_base; curB[i] = adB[i] + lds_base + G::A_BYTES; }
  frags_issue<TA, BMv, BKv, WTM, 0>(curA, fa0); frags_issue<TB, BNv, BKv, WTN, 0>(curB, fb0);
#if defined(COCODR_ABL_TIMELINE)
  if (tid == 0) tl[1] = wall_clock64();
#endif
  constexpr int SLOTS = G::KS - 1;  // DMA issue slots per K-step (in front of the MFMAs of sub-steps 0 .. KS-2)
  for (int t = 0; t < nt; ++t) {
    const bool more = SELF_ISSUE && t + D < nt;  // every wave is past barrier t: the slot of stage t-1 is free
    frags_issue<TA, BMv, BKv, WTM, 1>(curA, fa1); frags_issue<TB, BNv, BKv, WTN, 1>(curB, fb1);
#if !defined(COCODR_ABL_NO_DMA)  // ablation builds (tools/gemm_ablate.py) only; never defined in the product library
    if (more) static_for<0, NP / SLOTS>([&](auto jc) { issue_piece(jc, t + D); });
#endif
    wait_lgkmcnt<R>();
    mfma_step<TA, TB, WTM, WTN>(fa0, fb0, acc);
    if constexpr (G::KS == 4) {
      frags_issue<TA, BMv, BKv, WTM, 2>(curA, fa0); frags_issue<TB, BNv, BKv, WTN, 2>(curB, fb0);
#if !defined(COCODR_ABL_NO_DMA)
      if (more) static_for<NP / SLOTS, 2 * NP / SLOTS>([&](auto jc) { issue_piece(jc, t + D); });
#endif
      wait_lgkmcnt<R>();
      mfma_step<TA, TB, WTM, WTN>(fa1, fb1, acc);
      frags_issue<TA, BMv, BKv, WTM, 3>(curA, fa1); frags_issue<TB, BNv, BKv, WTN, 3>(curB, fb1);
#if !defined(COCODR_ABL_NO_DMA)
      if (more) static_for<2 * NP / SLOTS, NP>([&](auto jc) { issue_piece(jc, t + D); });
#endif
      wait_lgkmcnt<R>();
      mfma_step<TA, TB, WTM, WTN>(fa0, fb0, acc);
    }
    wait_lgkmcnt<0>();  // this wave is done reading stage t
    if (t + 1 < nt) {
      if (SELF_ISSUE) wait_vmcnt_stages<NP, D - 1>(min(nt - 1, t + D) - (t + 1));
      __builtin_amdgcn_s_barrier();  // barrier t+1: stage t+1 complete, stage t released
      const uint32_t sb = lds_base + (uint32_t)(((t + 1) % G::NSTAGE) * G::STAGE);
#pragma unroll
      for (int i = 0; i < 4; ++i) { curA[i] = adA[i] + sb; curB[i] = adB[i] + sb + G::A_BYTES; }
      frags_issue<TA, BMv, BKv, WTM, 0>(curA, fa0); frags_issue<TB, BNv, BKv, WTN, 0>(curB, fb0);
    }
    mfma_step<TA, TB, WTM, WTN>(fa1, fb1, acc);
  }
  }  // MFMA waves
  // residual / pre-activation chunks of the first epilogue pass: requested before the LDS transposition so their L2 / HBM
  // latency runs under it (each pass requests the next one's once its own stores are issued)
  const float* __restrict__ bias = p.bias ? p.bias + (size_t)z * p.strideBias : nullptr;
  const uint16_t* __restrict__ R_ = p.R ? p.R + (size_t)z * p.strideR : nullptr;
  constexpr int CPRW = BNv / 8;        // 8-column chunks per output row
  constexpr int RP = G::EPI_ROWS;      // rows per pass
  constexpr int NCH = RP * CPRW / G::CTHREADS;  // the copy-out is done by the MFMA waves
  constexpr bool PREFETCH_R = G::WG_PER_CU == 1;  // the 2-3 workgroups / CU geometries have no registers to spare for it
  const bool need_r = PREFETCH_R && R_ != nullptr && (p.epi == COCODR_EPI_ADD || p.epi == COCODR_EPI_DGELU) && !is_loader;
  uint4 rcur[PREFETCH_R ? NCH : 1];
  auto fetch_r = [&](int h, uint4 (&dst)[PREFETCH_R ? NCH : 1]) {
#pragma unroll
    for (int i = 0; i < (PREFETCH_R ? NCH : 0); ++i) {
      const int c = tid + i * G::CTHREADS;
      const int gm = m0 + h * RP + c / CPRW;
      dst[i] = make_uint4(0, 0, 0, 0);
      if (need_r && gm < p.M) dst[i] = *reinterpret_cast<const uint4*>(R_ + (size_t)gm * p.ldr + n0 + ((c % CPRW) << 3));
    }
  };
  fetch_r(0, rcur);
  // a thread's copy-out chunks all lie in the same 8 columns when CTHREADS % CPRW == 0: one bias fetch per tile instead of
  // two 16-B loads per chunk inside the store-bound copy-out loop
#if defined(COCODR_ABL_NO_BIAS_HOIST)
  constexpr bool COLS_FIXED = false;
#else
  constexpr bool COLS_FIXED = (G::CTHREADS % CPRW) == 0;
#endif
  float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (COLS_FIXED && bias != nullptr && !is_loader) {
    const float4 b0 = *reinterpret_cast<const float4*>(bias + n0 + ((tid % CPRW) << 3));
    const float4 b1 = *reinterpret_cast<const float4*>(bias + n0 + ((tid % CPRW) << 3) + 4);
    bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
  }
  __syncthreads();
#if defined(COCODR_ABL_TIMELINE)
  if (tid == 0) tl[2] = wall_clock64();
#endif

  // ---- epilogue through an fp32 LDS tile (as many rows per pass as the ring's LDS holds), row-major 16-B stores.
  // The barriers between the passes only order LDS traffic (lgkmcnt): waiting on vmcnt there would stall every pass
  // on the write acknowledgements of the previous one's global stores.
  float* ct = reinterpret_cast<float*>(smem);
  constexpr int CLD = G::CT_LDv;
  constexpr int WROWS = 32 * WTM;      // rows owned by one wave
  // fused column sums (bias gradient): a thread's chunks all lie in the same 8 columns (CTHREADS % CPRW == 0)
  constexpr bool CAN_COLSUM = (CPRW == 16) && (G::CTHREADS % CPRW == 0);
  const bool do_colsum = CAN_COLSUM && p.colsum_partial != nullptr;
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  static_assert(RP % WROWS == 0 && (RP * CPRW) % G::CTHREADS == 0, "epilogue pass geometry");
#pragma unroll 1
  for (int h = 0; h < BMv / RP; ++h) {
    if (!is_loader && (wm * WROWS) / RP == h) {
      const int rbase = wm * WROWS - h * RP;
#pragma unroll
      for (int ai = 0; ai < WTM; ++ai)
#pragma unroll
        for (int b = 0; b < WTN; ++b)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int row = rbase + ai * 32 + (lane & 31);
            const int col = wn * 32 * WTN + b * 32 + 8 * rg + 4 * (lane >> 5);
            *reinterpret_cast<float4*>(ct + row * CLD + col) =
                make_float4(acc[ai][b][rg * 4 + 0], acc[ai][b][rg * 4 + 1], acc[ai][b][rg * 4 + 2], acc[ai][b][rg * 4 + 3]);
          }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if defined(COCODR_ABL_TIMELINE)
    if (tid == 0 && h == 0) tl[5] = wall_clock64();  // own LDS writes done
#endif
    __builtin_amdgcn_s_barrier();
#if defined(COCODR_ABL_TIMELINE)
    if (tid == 0 && h == 0) tl[6] = wall_clock64();  // everybody's LDS writes done
#endif
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * G::CTHREADS;
      const int row = c / CPRW, c8 = (c % CPRW) << 3;
      const int gm = m0 + h * RP + row;
      const int gn = n0 + c8;
      if (!is_loader && gm < p.M) {
        float v[8];
        const float4 c0 = *reinterpret_cast<const float4*>(ct + row * CLD + c8);
        const float4 c1 = *reinterpret_cast<const float4*>(ct + row * CLD + c8 + 4);
        v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w; v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
        epilogue_store8<OUT_F32, PREFETCH_R, COLS_FIXED>(p, z, bias, R_, gm, gn, v, rcur[PREFETCH_R ? i : 0], bias8);
        if (do_colsum) {
#pragma unroll
          for (int j = 0; j < 8; ++j) csum[j] += v[j];
        }
      }
    }
    if (h + 1 < BMv / RP) {
      fetch_r(h + 1, rcur);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  if constexpr (CAN_COLSUM) {
    if (do_colsum) {  // workgroup-uniform
      // lanes l, l^16, l^32, l^48 hold the same 8 columns (4 rows per wave instruction); then one row of 128 partial
      // sums per MFMA wave goes through the LDS behind the fp32 tile and 128 threads add them in wave order
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        csum[j] += __shfl_xor(csum[j], 16, 64);
        csum[j] += __shfl_xor(csum[j], 32, 64);
      }
      float* cred = ct + RP * CLD;
      if (!is_loader && lane < 16) {
#pragma unroll
        for (int j = 0; j < 8; ++j) cred[wid * BNv + lane * 8 + j] = csum[j];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tid < BNv) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < G::NWAVES; ++w) t += cred[w * BNv + tid];
        p.colsum_partial[(size_t)tm_ * p.N + n0 + tid] = t;
      }
    }
  }
#if defined(COCODR_ABL_TIMELINE)
  if (tid == 0) tl[3] = wall_clock64();
#endif
#endif
}

template <int BMv, int BKv, int WTM, int WTN, int LD, int TA, int TB, int WC = 2, int NS = 3>
void launch_glds(const cocodr_gemm_args& a, hipStream_t st) {
  using G = Geom<BMv, BKv, WTM, WTN, LD, WC, NS>;
  const int ntm = (a.M + BMv - 1) / BMv, ntn = a.N / G::BNv;
  static int flat_env = -1;  // COCODR_PP_FLAT=0 keeps the per-item remap (A/B switch, shared with gemm_pp.hip)
  if (flat_env < 0) {
    const char* e = getenv("COCODR_PP_FLAT");
    flat_env = e ? atoi(e) : 1;
  }
  const int flat = (a.batch > 1 && flat_env) ? 1 : 0;
  dim3 grid(flat ? ntm * ntn * a.batch : ntm * ntn, flat ? 1 : a.batch);
  const size_t lds = (size_t)G::NSTAGE * G::STAGE;
  static cocodr_lds_once attr_done;
  if (attr_done.pending()) {
    hipFuncSetAttribute((const void*)gemm_glds_kernel<BMv, BKv, WTM, WTN, LD, TA, TB, true, WC, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)gemm_glds_kernel<BMv, BKv, WTM, WTN, LD, TA, TB, false, WC, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.done();
  }
  static int stagger = -1;  // x 4096 clocks; COCODR_GEMM_STAGGER overrides (0 disables)
  if (stagger < 0) {
    const char* e = getenv("COCODR_GEMM_STAGGER");
    stagger = e ? atoi(e) : 0;
  }
  const int sg = (int)(grid.x * grid.y) > 256 * G::WG_PER_CU / 2 ? stagger : 0;
  if (a.out_f32)
    hipLaunchKernelGGL((gemm_glds_kernel<BMv, BKv, WTM, WTN, LD, TA, TB, true, WC, NS>), grid, dim3(G::NTHREADS), lds, st, a, sg, flat);
  else
    hipLaunchKernelGGL((gemm_glds_kernel<BMv, BKv, WTM, WTN, LD, TA, TB, false, WC, NS>), grid, dim3(G::NTHREADS), lds, st, a, sg, flat);
}

template <int BMv, int BKv, int WTM, int WTN = 2, int LD = 0, int WC = 2, int NS = 3>
void launch_glds_any(const cocodr_gemm_args& a, hipStream_t st) {
  if (!a.trans_a && !a.trans_b) launch_glds<BMv, BKv, WTM, WTN, LD, 0, 0, WC, NS>(a, st);
  else if (!a.trans_a && a.trans_b) launch_glds<BMv, BKv, WTM, WTN, LD, 0, 1, WC, NS>(a, st);
  else launch_glds<BMv, BKv, WTM, WTN, LD, 1, 1, WC, NS>(a, st);
}

}  // namespace cocodr_gemm_v2
using cocodr_gemm_v2::launch_glds_any;
void cocodr_gemm_pp_launch(const cocodr_gemm_args& a, int nb, hipStream_t st);  // gemm_pp.hip: the ping-pong pipeline
bool cocodr_gemm_pp_launch_split(const cocodr_gemm_args& a, hipStream_t st);    // ... with its last partial round cut into contraction slices
void cocodr_gemm_pp_split_plan(const cocodr_gemm_args& a, int& total, int& r, int& s);
size_t cocodr_gemm_pp_split_ws_floats();
void cocodr_gemm_pp_launch_multi(const cocodr_gemm_args* a, int n, float* ws, size_t ws_floats, hipStream_t st);
size_t cocodr_gemm_pp_multi_ws_floats();
size_t cocodr_gemm_pp_multi_ws_floats_for(const cocodr_gemm_args* a, int n);
bool cocodr_gemm_a4_ok(const cocodr_gemm_args& a);                              // gemm_a4.hip: one wave per SIMD, hand-scheduled K loop (NT form)
void cocodr_gemm_a4_launch(const cocodr_gemm_args& a, hipStream_t st);
bool cocodr_gemm_a4_walk_ok(const cocodr_gemm_args& a);                         // ... as a persistent walk with a register epilogue
void cocodr_gemm_a4_walk_launch(const cocodr_gemm_args& a, hipStream_t st);
void cocodr_gemm_a4_walk_launch_multi(const cocodr_gemm_args* a, int n, hipStream_t st);   // 2..4 weight-gradient problems in one walk

namespace {

int g_gemm_impl = -1;  // 0 = auto, 1 = register-staged v1, direct-to-LDS <BM,BK,WTM>: 2 = <128,64,2>, 3 = <256,64,2>, 4 = <128,32,2>, 5 = <256,32,2>, 6 = <256,32,4>, 7 = <256,64,4>, 8 = 128x192 tile <128,64,2> with 64x96 wave tiles, 9 = <256,64,2> + 4 loader waves, 10 = <256,64,4> + 4 loader waves, 11 = 256x256 tile, 12 = 256x96 tile: 4x3 waves of 64x32 + 4 loader waves
int gemm_impl_override() {
  if (g_gemm_impl < 0) {
    const char* e = getenv("COCODR_GEMM_IMPL");
    g_gemm_impl = e ? atoi(e) : 0;
  }
  return g_gemm_impl;
}

template <int TA, int TB>
void launch(const cocodr_gemm_args& a, dim3 grid, hipStream_t st) {
  if (a.out_f32)
    hipLaunchKernelGGL((gemm_kernel<TA, TB, true>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((gemm_kernel<TA, TB, false>), grid, dim3(256), 0, st, a);
}

}  // namespace

extern "C" int cocodr_gemm_set_impl(int impl) {
  CK_ARG((impl >= 0 && impl <= 15) || impl == 18, "gemm_set_impl: impl must be in [0,15] or 18 (ping-pong with two fat phases per K-tile)");
  g_gemm_impl = impl;
  return COCODR_OK;
}

namespace {
// the window in which cutting the last partial round of 256 x 256 tiles into contraction slices was measured to win (see select_impl):
// r <= 64 tiles in s >= 4 slices at K >= 2048.  An automatically selected launch is cut ONLY there (a plan with fewer slices or a
// longer tail exists - cocodr_gemm_pp_split_plan is general, the parity tests force it - but loses to whole tiles)
bool split_tail_wins(const cocodr_gemm_args& a) {
  if (a.split_ws == nullptr || a.batch > 1 || a.K < 2048) return false;
  int total, r, sl;
  cocodr_gemm_pp_split_plan(a, total, r, sl);
  return sl >= 4 && r <= 64;
}
// which pipeline a call runs on (see g_gemm_impl); shape-only, so callers can ask before they launch
int select_impl(const cocodr_gemm_args& a) {
  // the direct-to-LDS pipeline needs whole 64-wide K-steps for every operand whose contraction index is the
  // fast axis (ragged K is only zero-filled along rows), and 32-bit byte offsets per batch item
  const bool k_ok = (a.K % BK == 0) || (a.trans_a && a.trans_b);
  const bool small = (size_t)(a.trans_a ? a.K : a.M) * a.lda * 2 < (1ull << 32) && (size_t)(a.trans_b ? a.K : a.N) * a.ldb * 2 < (1ull << 32);
  const int batch = a.batch > 0 ? a.batch : 1;
  int impl = gemm_impl_override();
  if (impl == 0) {
    // auto (measured on MI355X, tools/gemm_bench.py): with >= 1.5 tiles per CU the BK=32 geometry wins because a
    // second resident workgroup hides prologue/epilogue; with fewer tiles (and for the mid-sized grouped wgrad) the
    // deeper BK=64 ring with four loader waves wins - also for the long-K forward / dgrad forms (K >= 2048) at any tile
    // count, where the per-tile prologue / epilogue is amortised anyway; below half a wave of 256-row tiles fall back
    // to 128-row tiles to occupy more CUs.
    const long long tiles256 = (long long)((a.M + 255) / 256) * (a.N / BN) * batch;
    const long long tiles128 = (long long)((a.M + 127) / 128) * (a.N / BN) * batch;
    // 256x96 tiles (no fused column sums there): when they need fewer rounds of the 256 CUs even at 3/4 of a tile's work
    // each - the N = 768 GEMMs of 8192 tokens become exactly one tile per CU instead of 192 tiles, QKV 3 rounds of
    // smaller tiles instead of 2.25 (measured 2-8 % on those shapes; the long-K wgrads lose 10-60 % and stay out)
    const long long tiles96 = a.N % 96 == 0 ? (long long)((a.M + 255) / 256) * (a.N / 96) * batch : 0;
    static const bool no96 = getenv("COCODR_GEMM_NO96") != nullptr;  // A/B switch of this rule
    // (only while there are few rounds to quantise: from ~3 rounds of 256-row tiles on, the larger tiles win again - packed
    // batches of 512 sequences, 45 312 x 768 x 3072: 814 vs 946-970 TFLOP/s)
    const bool fewer_rounds = !no96 && tiles96 > 0 && tiles256 <= 768 && ((tiles96 + 255) / 256) * 3 < ((tiles256 + 255) / 256) * 4;
    // ping-pong pipeline (gemm_pp.hip, 256x256 tiles): wins once its tiles fill the 256 CUs for nearly whole rounds -
    // 2-12 % on the forward / dgrad forms from 400 tiles (BERT-large FFN1 at 8192 tokens, everything at 25600 tokens),
    // 3-7 % on the grouped weight gradients from ~1500 tiles; with 1.5 rounds or less it loses to the 256x128 tiles
    // (profiles/archive/r02_gemm_pp_vs_glds.txt)
    const long long tilespp = a.N % 256 == 0 ? (long long)((a.M + 255) / 256) * (a.N / 256) * batch : 0;
    const long long roundspp = (tilespp + 255) / 256;
    static const bool nopp = getenv("COCODR_GEMM_NOPP") != nullptr;  // A/B switch of this rule
    // (with two fat phases per K-tile the grouped weight gradients gain from ~400 tiles as well: profiles/archive/r02_gemm_pp_fat.txt)
    bool pp_fills = !nopp && tilespp >= 400 && tilespp * 100 >= roundspp * 256 * 78;
    // with a split workspace the last partial round can run as contraction slices (gemm_pp.hip launch_split).  Measured over
    // packed row counts 4 416 ... 26 016 (profiles/r04_gemm_tail_sweep.txt): it beats every other pipeline only where the
    // tail is short and the contraction long - r <= 64 tiles in s >= 4 slices at K >= 2048 (17 888 rows x 1024 x 4096: 145 us
    // against 163 for the 256 x 128 tiles and 180 for whole 256 x 256 tiles; the same shape at K = 1024: 64 against 55) - and
    // loses wherever the tail is long (368 tiles: 200 against 180) because two slices of a tile cost a ramp, a 256 KB fp32
    // partial tile each way and the finishing pass.  So: exactly that window.
    if (!nopp && !pp_fills && tilespp > 256 && split_tail_wins(a)) pp_fills = true;
    // one round of 256 x 256 tiles that fills >= 2/3 of the CUs: the ping-pong pipeline beats the 256 x 96 / 256 x 128 tiles by
    // 8-11 % on the forward (NT) form there (packed BERT-base batches: 5 024-6 304 rows x 2304 x 768 = 180-225 tiles, 28-31 us
    // against 31-34; profiles/r04_gemm_impl_sweep_base.txt); the dgrad (NN) form is level at BERT-base sizes and 3-9 % ahead
    // at BERT-large ones (15 040 rows x 1024 x 4096 = 236 tiles: 110 us against 120; profiles/r04_gemm_impl_sweep_unaligned.txt)
    const bool pp_one_round = !nopp && !a.trans_a && batch == 1 && tilespp >= 176 && tilespp <= 256;
    // hand-scheduled one-wave-per-SIMD kernel as a persistent walk (gemm_a4.hip; the forward NT form): ahead of every other pipeline
    // when its 256 x 256 tiles fill one round of the CUs to >= 5/8 or several rounds to >= 80 % (profiles/r06_gemm_a4.md: 32 768 x
    // 3072 x 1024 165 us against 207; 4 776 x 2304 x 768 25.2 against 28.6), behind the smaller tiles in between (276-384 tiles)
    static const bool noa4 = getenv("COCODR_GEMM_NOA4") != nullptr;  // A/B switch of this rule
    bool a4_wins = false;
    if (!noa4 && cocodr_gemm_a4_walk_ok(a)) {   // (every form: NT, NN with the [K, N] operand and TN with both operands through transposing LDS reads)
      const long long t = tilespp;
      if (t <= 256) a4_wins = t >= (a.epi == COCODR_EPI_GELU ? 224 : 160);
      else {
        const long long rounds = (t + 255) / 256;
        a4_wins = t * 100 >= rounds * 256 * 80;
      }
    }
    if (!(k_ok && small)) impl = 1;
    else if (a4_wins) impl = 15;
    else if (pp_fills || pp_one_round) impl = 13;
    else if (fewer_rounds && !a.trans_a && tiles256 >= 128 && !a.colsum && !a.colsum_partial) impl = 12;
    else if (tiles256 >= 384 && !(a.trans_a && tiles256 < 800) && !(!a.trans_a && a.K >= 2048)) impl = 5;
    else if (tiles256 >= 128) impl = 9;
    else impl = tiles128 >= 512 ? 4 : 2;
  }
  if (impl != 1 && !(k_ok && small)) impl = 1;
  if (impl == 8 && a.N % 192 != 0) impl = 3;  // the 128x192 tile needs N % 192 == 0
  if (impl == 11 && a.N % 256 != 0) impl = 5;  // the 256x256 tile needs N % 256 == 0
  if (impl == 12 && a.N % 96 != 0) impl = 9;   // the 256x96 tile needs N % 96 == 0
  if (impl >= 13 && a.N % 256 != 0) impl = 9;  // the ping-pong pipeline's 256x256 tile needs N % 256 == 0
  if (impl == 15 && !cocodr_gemm_a4_walk_ok(a)) impl = 14;
  if (impl == 14 && !cocodr_gemm_a4_ok(a)) impl = 13;  // the hand-scheduled kernel: NT form, K % 128 == 0, K >= 256
  return impl;
}
// row panels of the fused column sums for that pipeline (0: not fused there)
int colsum_rows(int impl, int M) {
  if (impl == 1 || impl == 8 || impl == 11 || impl == 12) return 0;
  const int bm = (impl == 2 || impl == 4) ? 128 : 256;
  return (M + bm - 1) / bm;
}
}  // namespace

extern "C" int cocodr_gemm_colsum_rows(const cocodr_gemm_args* args) {
  if (!args || args->M <= 0 || args->N <= 0 || args->K <= 0 || args->N % BN != 0 || args->batch > 1 || args->out_f32) return 0;
  cocodr_gemm_args q = *args;
  static float dummy;
  if (!q.colsum_partial) q.colsum_partial = &dummy;  // "which pipeline would this call run on with column sums requested"
  return colsum_rows(select_impl(q), q.M);
}

extern "C" int cocodr_gemm(const cocodr_gemm_args* args, cocodr_stream_t stream) {
  CK_ARG(args != nullptr, "gemm: null args");
  cocodr_gemm_args a = *args;
  if (a.batch <= 0) a.batch = 1;
  CK_ARG(a.A && a.B && a.C, "gemm: null operand");
  CK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "gemm: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
  CK_ARG(a.N % BN == 0, "gemm: N=%d must be a multiple of 128", a.N);
  CK_ARG(a.K % 8 == 0, "gemm: K=%d must be a multiple of 8", a.K);
  CK_ARG(!a.trans_a || a.M % 8 == 0, "gemm: M=%d must be a multiple of 8 when trans_a", a.M);
  CK_ARG(a.lda % 8 == 0 && a.ldb % 8 == 0 && a.ldc % 8 == 0, "gemm: leading dims must be multiples of 8");
#if !defined(COCODR_ABL_ALIAS_LD)  // ablation builds may alias operand rows (L2-resident window)
  CK_ARG(a.lda >= (a.trans_a ? a.M : a.K) && a.ldb >= (a.trans_b ? a.N : a.K) && a.ldc >= a.N, "gemm: leading dim too small");
#endif
  CK_ARG(!(a.trans_a && !a.trans_b), "gemm: (trans_a=1, trans_b=0) is not used on this path");
  CK_ARG(a.epi >= COCODR_EPI_NONE && a.epi <= COCODR_EPI_DGELU, "gemm: bad epilogue %d", a.epi);
  CK_ARG(a.epi != COCODR_EPI_GELU || !a.out_f32, "gemm: EPI_GELU needs a bf16 output");
  CK_ARG((a.epi != COCODR_EPI_ADD && a.epi != COCODR_EPI_DGELU) || (a.R && a.ldr % 8 == 0 && a.ldr >= a.N), "gemm: epilogue needs R");
  CK_ARG((((uintptr_t)a.A | (uintptr_t)a.B | (uintptr_t)a.C | (uintptr_t)a.C2 | (uintptr_t)a.R | (uintptr_t)a.bias) & 15) == 0,
         "gemm: pointers must be 16-byte aligned");
  if (a.epi != COCODR_EPI_ADD && a.epi != COCODR_EPI_DGELU) a.R = nullptr;
  CK_ARG(a.drop.threshold == 0 || (a.epi == COCODR_EPI_ADD && a.batch == 1 && !a.colsum && !a.colsum_partial && a.drop.threshold < 65536),
         "gemm: dropout belongs to an EPI_ADD call with batch == 1 and no column sums");
  CK_ARG(a.split_ws == nullptr || (((uintptr_t)a.split_ws & 15) == 0), "gemm: split_ws must be 16-byte aligned");
  CK_ARG(!(a.colsum || a.colsum_partial) || (a.colsum_partial && a.batch == 1 && !a.out_f32),
         "gemm: column sums need colsum_partial, batch == 1 and a bf16 output");
  float* const cs_out = a.colsum;
  float* const cs_part = a.colsum_partial;
  const int ntm = (a.M + BM - 1) / BM, ntn = a.N / BN;
  dim3 grid(ntm * ntn, a.batch);
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(PROF_GEMM, st, 2.0 * a.M * a.N * (double)a.K * a.batch);
  if (a.ab_f16) {  // IEEE-half operands: the ping-pong pipeline's NT form with an fp32 result, nothing fused
    CK_ARG(!a.trans_a && !a.trans_b && a.out_f32 && a.epi == COCODR_EPI_NONE && !a.bias && !cs_part && a.N % 256 == 0 && a.K % BK == 0,
           "gemm: fp16 operands need the plain NT form with an fp32 result, N %% 256 == 0 and K %% 64 == 0");
    CK_ARG((size_t)a.M * a.lda * 2 < (1ull << 32) && (size_t)a.N * a.ldb * 2 < (1ull << 32), "gemm: fp16 operands: each operand must stay below 4 GiB per batch item");
    cocodr_gemm_pp_launch(a, 104, st);
    CK_LAUNCH("gemm(f16)");
    return COCODR_OK;
  }
  const int impl = select_impl(a);
  const int cs_rows = colsum_rows(impl, a.M);  // the kernels with 128-column tiles reduce in their epilogue
  CK_ARG(!cs_part || cs_out || cs_rows > 0, "gemm: deferred column sums (colsum == NULL) are not available on this pipeline; ask cocodr_gemm_colsum_rows first");
  if (cs_rows == 0) a.colsum_partial = nullptr;
  // (a pipeline forced through cocodr_gemm_set_impl takes any cut the plan allows - tests, sweeps; the automatic selection only the measured window)
  if (impl == 13 && (gemm_impl_override() != 0 || split_tail_wins(a)) && cocodr_gemm_pp_launch_split(a, st)) { /* whole rounds + a cut last round */ }
  else if (impl == 15) cocodr_gemm_a4_walk_launch(a, st);
  else if (impl == 14) cocodr_gemm_a4_launch(a, st);
  else if (impl >= 13) cocodr_gemm_pp_launch(a, impl == 18 ? 105 : 2, st);
  else if (impl == 12) launch_glds_any<256, 64, 2, 1, 4, 3>(a, st);
  else if (impl == 11) launch_glds_any<256, 32, 2, 4>(a, st);
  else if (impl == 10) launch_glds_any<256, 64, 4, 2, 4>(a, st);
  else if (impl == 9) launch_glds_any<256, 64, 2, 2, 4>(a, st);
  else if (impl == 8) launch_glds_any<128, 64, 2, 3>(a, st);
  else if (impl == 7) launch_glds_any<256, 64, 4>(a, st);
  else if (impl == 6) launch_glds_any<256, 32, 4>(a, st);
  else if (impl == 5) launch_glds_any<256, 32, 2>(a, st);
  else if (impl == 4) launch_glds_any<128, 32, 2>(a, st);
  else if (impl == 3) launch_glds_any<256, 64, 2>(a, st);
  else if (impl == 2) launch_glds_any<128, 64, 2>(a, st);
  else if (!a.trans_a && !a.trans_b) launch<0, 0>(a, grid, st);
  else if (!a.trans_a && a.trans_b) launch<0, 1>(a, grid, st);
  else launch<1, 1>(a, grid, st);
  CK_LAUNCH("gemm");
  if (cs_out) {
    if (cs_rows > 0) return cocodr_reduce_partials(cs_part, cs_out, nullptr, nullptr, cs_rows, 1, a.N, 1, 0, st);
    return cocodr_colsum((const uint16_t*)a.C, cs_out, cs_part, a.M, a.N, a.ldc, 1, 0, 0, stream);
  }
  return COCODR_OK;
}

extern "C" size_t cocodr_gemm_split_workspace_floats(void) { return cocodr_gemm_pp_split_ws_floats(); }

extern "C" size_t cocodr_gemm_colsum_partial_floats(int M, int N) {
  const size_t panels = (size_t)(M + 127) / 128;
  const size_t fused = panels * (size_t)(N > 0 ? N : 0);
  return std::max(fused, cocodr_colsum_partial_floats(M, N, 1));  // the pipelines without fused sums run cocodr_colsum on the result
}

// n independent (batched) weight-gradient problems - form TN, fp32 result, no epilogue - as ONE launch on the ping-pong pipeline
// when together they fill it (>= 400 tiles of 256 x 256); otherwise, or when a problem does not fit that form, n plain calls.
extern "C" size_t cocodr_gemm_multi_workspace_floats(void) { return cocodr_gemm_pp_multi_ws_floats(); }

namespace {
long long multi_min_tiles() {
  static const long long v = getenv("COCODR_GEMM_MULTI_MIN") ? atoll(getenv("COCODR_GEMM_MULTI_MIN")) : 400;  // tuning hook
  return v;
}
// tiles of the merged launch; *form_ok: every problem has the form the merged launch takes (and nothing switched it off)
long long multi_tiles(const cocodr_gemm_args* problems, int n, bool* form_ok) {
  static const bool off = getenv("COCODR_GEMM_NOMULTI") != nullptr;  // A/B switch
  bool ok = !off && n > 1 && gemm_impl_override() == 0;
  long long tiles = 0;
  for (int q = 0; q < n && ok; ++q) {
    const cocodr_gemm_args& a = problems[q];
    ok = a.trans_a && a.trans_b && a.out_f32 && a.epi == COCODR_EPI_NONE && !a.bias && !a.colsum &&
         !a.colsum_partial && !a.drop.threshold && !a.ab_f16 && a.M > 0 && a.N > 0 && a.K == problems[0].K && a.K > 0 &&
         a.N % 256 == 0 && a.M % 8 == 0 && a.K % 8 == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0 && a.ldc % 8 == 0 && a.lda >= a.M &&
         a.ldb >= a.N && a.ldc >= a.N && (size_t)a.K * a.lda * 2 < (1ull << 32) && (size_t)a.K * a.ldb * 2 < (1ull << 32);
    tiles += (long long)((a.M + 255) / 256) * (a.N / 256) * (a.batch > 0 ? a.batch : 1);
  }
  *form_ok = ok;
  return tiles;
}
}  // namespace

// workspace floats THIS call would use (0: the problems do not run merged, or the merged launch has no last round to cut);
// pointers are not examined, so layouts can ask with shapes alone
extern "C" size_t cocodr_gemm_multi_workspace_floats_for(const cocodr_gemm_args* problems, int n) {
  if (!problems || n < 1 || n > 4) return 0;
  bool ok = false;
  const long long tiles = multi_tiles(problems, n, &ok);
  if (!(ok && tiles >= multi_min_tiles() && tiles < (1ll << 30))) return 0;
  return cocodr_gemm_pp_multi_ws_floats_for(problems, n);
}

extern "C" int cocodr_gemm_multi(const cocodr_gemm_args* problems, int n, float* workspace, size_t workspace_floats,
                                 cocodr_stream_t stream) {
  CK_ARG(problems != nullptr && n >= 1 && n <= 4, "gemm_multi: 1..4 problems");
  CK_ARG(workspace == nullptr || (((uintptr_t)workspace & 15) == 0), "gemm_multi: workspace must be 16-byte aligned");
  bool ok = false;
  const long long tiles = multi_tiles(problems, n, &ok);
  for (int q = 0; q < n && ok; ++q) ok = problems[q].A && problems[q].B && problems[q].C;
  static const bool noa4 = getenv("COCODR_GEMM_NOA4") != nullptr;  // A/B switch
  bool a4 = ok && !noa4 && tiles >= multi_min_tiles() && tiles < (1ll << 30);
  for (int q = 0; q < n && a4; ++q) a4 = cocodr_gemm_a4_walk_ok(problems[q]);
  if (a4) {  // the hand-scheduled kernel walks the tiles of all problems (gemm_a4.hip): no partial last round to cut, no workspace
    double flops = 0.0;
    for (int q = 0; q < n; ++q) flops += 2.0 * problems[q].M * problems[q].N * (double)problems[q].K * (problems[q].batch > 0 ? problems[q].batch : 1);
    ProfScope prof(PROF_GEMM, (hipStream_t)stream, flops);
    cocodr_gemm_a4_walk_launch_multi(problems, n, (hipStream_t)stream);
    CK_LAUNCH("gemm_multi(a4)");
    return COCODR_OK;
  }
  if (ok && tiles >= multi_min_tiles() && tiles < (1ll << 30)) {
    cocodr_gemm_args copy[4];
    double flops = 0.0;
    for (int q = 0; q < n; ++q) {
      copy[q] = problems[q];
      if (copy[q].batch <= 0) copy[q].batch = 1;
      flops += 2.0 * copy[q].M * copy[q].N * (double)copy[q].K * copy[q].batch;
    }
    ProfScope prof(PROF_GEMM, (hipStream_t)stream, flops);  // bench.py's roofline sample: one launch, the FLOPs of all problems
    cocodr_gemm_pp_launch_multi(copy, n, workspace, workspace_floats, (hipStream_t)stream);
    CK_LAUNCH("gemm_multi");
    return COCODR_OK;
  }
  for (int q = 0; q < n; ++q) {
    const int rc = cocodr_gemm(&problems[q], stream);
    if (rc != COCODR_OK) return rc;
  }
  return COCODR_OK;
}
