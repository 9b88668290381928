// Device-side building blocks shared by the direct-to-LDS GEMM pipelines (gemm.hip, gemm_pp.hip): counted waits, LDS
// fragment reads issued from inline asm, the chunk swizzles of the operand tile images, the per-lane DMA source offsets
// and the per-row epilogue.  See gemm.hip for the design notes.
#pragma once
#include <type_traits>

#include "common.h"

namespace cocodr_gemm_v2 {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// "all but the k most recently issued stages (NP loads each) have landed", k wave-uniform in [0, KMAX]
template <int NP, int KMAX, int K = 0>
__device__ __forceinline__ void wait_vmcnt_stages(int k) {
  static_assert(NP * KMAX <= 63, "vmcnt is a 6-bit counter");
  if constexpr (K >= KMAX) {
    wait_vmcnt<NP * KMAX>();
  } else {
    if (k == K) wait_vmcnt<NP * K>();
    else wait_vmcnt_stages<NP, KMAX, K + 1>(k);
  }
}
template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <int OFF>
__device__ __forceinline__ void asm_ds_read_b128(v4i& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void asm_ds_read_tr16(v2i& dst, uint32_t addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

// L2-aware tile order (each XCD has a private 4 MiB L2 and receives a contiguous range of tile ids, see
// xcd_remap): ids walk GM row-panels down, then across the columns, so any ~32 consecutive ids (= the tiles an
// XCD's CUs hold at once) form a GM x (32/GM) rectangle whose A and B panels fit the L2 together; the next
// rectangle reuses the same A panels.
__device__ __forceinline__ void grouped_tile(int id, int ntm, int ntn, int gm, int& tm, int& tn) {
  const int per_group = gm * ntn;
  const int grp = id / per_group;
  const int first = grp * gm;
  const int rows = min(gm, ntm - first);
  const int r = id - grp * per_group;
  tm = first + r % rows;
  tn = r / rows;
}

template <int LO, int HI, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (LO < HI) {
    f(std::integral_constant<int, LO>{});
    static_for<LO + 1, HI>(f);
  }
}

// chunk swizzle of a row-major [rows][BK] tile (16-B chunks): conflict-free ds_read_b128 fragment reads
template <int BKv>
__device__ __forceinline__ int swz_rows(int row) {
  return BKv == 64 ? swz64(row) : ((row >> 2) & 3);
}
template <int BKv>
__device__ __forceinline__ int tile_rows_off(int row, int ch) {
  return row * (BKv * 2) + ((ch ^ swz_rows<BKv>(row)) << 4);
}

// chunk swizzle of a [k][COLS] tile read with ds_read_b64_tr_b16 (4 rows x 64 B per 32 lanes): rows r..r+3 must fall
// into the four 64-B windows of a 256-B bank row.  256-B / 512-B rows: XOR the chunk with (row & 3) << 2; 384-B rows
// (COLS = 192) already alternate 128-B halves, so only bit 2 is flipped on rows 2,3 (keeps chunk < 24).
// 192-B rows (COLS = 96) step through the four windows by themselves (0, 192, 128, 64 mod 256): no swizzle.
template <int COLS>
__device__ __forceinline__ int swz_cols(int row) {
  return COLS == 96 ? 0 : (COLS == 192 ? (((row >> 1) & 1) << 2) : ((row & 3) << 2));
}

// byte offset (relative to the matrix base) of the 16-B chunk that must land at linear LDS chunk p of a tile
template <int TR, int COLS /* tile width when stored [k][COLS] */, int BKv>
__device__ __forceinline__ uint32_t glds_src_off(int p, int r0, int ld) {
  if (TR == 0) {
    constexpr int CPR = BKv / 8;
    const int row = p / CPR, ch = (p % CPR) ^ swz_rows<BKv>(row);
    return (uint32_t)(((r0 + row) * ld + ch * 8) * 2);
  } else {
    constexpr int CPR = COLS / 8;
    const int row = p / CPR, ch = (p % CPR) ^ swz_cols<COLS>(row);
    return (uint32_t)((row * ld + r0 + ch * 8) * 2);
  }
}

template <int TR, int NF>
struct FragSet {  // the NF 32-row fragments of one operand for one K-sub-step
  v4i q[NF];
  v2i lo[NF], hi[NF];
};

// per-lane LDS byte offsets inside an operand tile: TR=0 -> one per K-sub-step (fragment a adds 32 rows),
// TR=1 -> one per fragment a (sub-step s and the +4-row half are immediates)
template <int TR, int COLS, int BKv, int NF>
__device__ __forceinline__ void frag_addrs(int r0, int lane, uint32_t (&ad)[4]) {
  ad[0] = ad[1] = ad[2] = ad[3] = 0;
  if (TR == 0) {
#pragma unroll
    for (int s = 0; s < BKv / 16; ++s) ad[s] = (uint32_t)tile_rows_off<BKv>(r0 + (lane & 31), 2 * s + (lane >> 5));
  } else {
    const int g = lane >> 4, c = lane & 15;
    const int row = ((g >> 1) << 3) + (c >> 2);
#pragma unroll
    for (int a = 0; a < NF; ++a) {
      const int col = r0 + a * 32 + ((g & 1) << 4) + ((c & 3) << 2);
      ad[a] = (uint32_t)(row * (COLS * 2) + (((col >> 3) ^ swz_cols<COLS>(row)) << 4) + ((col & 7) << 1));
    }
  }
}

template <int TR, int COLS, int BKv, int NF, int S, int A = 0>
__device__ __forceinline__ void frags_issue(const uint32_t (&cur)[4], FragSet<TR, NF>& f) {
#if defined(COCODR_ABL_NO_LDSREAD)
  return;
#endif
  if constexpr (A < NF) {
    if constexpr (TR == 0) {
      asm_ds_read_b128<A * 32 * BKv * 2>(f.q[A], cur[S]);
    } else {
      constexpr int o = S * 16 * COLS * 2;
      asm_ds_read_tr16<o>(f.lo[A], cur[A]);
      asm_ds_read_tr16<o + 4 * COLS * 2>(f.hi[A], cur[A]);
    }
    frags_issue<TR, COLS, BKv, NF, S, A + 1>(cur, f);
  }
}
template <int TR, int NF>
__device__ __forceinline__ bf16x8 frag_get(const FragSet<TR, NF>& f, int a) {
  if constexpr (TR == 0) {
    return __builtin_bit_cast(bf16x8, f.q[a]);
  } else {
    const v4i v = {f.lo[a][0], f.lo[a][1], f.hi[a][0], f.hi[a][1]};
    return __builtin_bit_cast(bf16x8, v);
  }
}
template <int TA, int TB, int WTM, int WTN>
__device__ __forceinline__ void mfma_step(const FragSet<TA, WTM>& fa, const FragSet<TB, WTN>& fb, f32x16 (&acc)[WTM][WTN]) {
#if defined(COCODR_ABL_NO_MFMA)
  return;
#endif
  bf16x8 a[WTM], b[WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i) a[i] = frag_get<TA, WTM>(fa, i);
#pragma unroll
  for (int j = 0; j < WTN; ++j) b[j] = frag_get<TB, WTN>(fb, j);
  // operands swapped: D[i = n][j = m], so a lane ends up with 4 consecutive n of one m
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
}

// one 8-column slice of an output row: bias / GELU / residual / GELU' and the store
template <bool OUT_F32, bool RPRE = false, bool BPRE = false>
__device__ __forceinline__ void epilogue_store8(const cocodr_gemm_args& p, int z, const float* __restrict__ bias,
                                                const uint16_t* __restrict__ R_, int gm, int gn, float (&v)[8],
                                                uint4 rpre = make_uint4(0, 0, 0, 0), const float* bpre = nullptr) {
  if constexpr (BPRE) {  // the caller's chunks all sit in the same 8 columns: their bias was fetched once (zeros without a bias)
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += bpre[j];
  } else if (bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(bias + gn);
    const float4 b1 = *reinterpret_cast<const float4*>(bias + gn + 4);
    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
  }
  if (p.epi == COCODR_EPI_GELU && p.C2 == nullptr) {  // inference: nobody needs the derivative
    gelu_erf8(v);
  } else if (p.epi == COCODR_EPI_GELU) {
    float gp[8], u[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) u[j] = v[j];
    gelu_erf_both8(u, v, gp);
    {  // GELU' is read by the backward only: streamed past the caches (non-temporal), so it does not evict the h tile FFN2 reads next
       // (same-box A/B of the two builds, round 5: GEMM class +0.4 % on the BERT-base step, +0.5 % at 256 padded BERT-large sequences)
      typedef uint32_t u4nt __attribute__((ext_vector_type(4)));
      const uint4 g4 = pack8(gp);
      const u4nt v = {g4.x, g4.y, g4.z, g4.w};
      __builtin_nontemporal_store(v, reinterpret_cast<u4nt*>(p.C2 + (size_t)z * p.strideC + (size_t)gm * p.ldc + gn));
    }
  } else if (p.epi == COCODR_EPI_ADD) {
    if (p.drop.threshold) drop_apply<8>(v, (uint64_t)gm * p.N + gn, p.drop);  // hf BertSelfOutput / BertOutput: dropout(dense(x)) + residual
    float r[8];
    unpack8(RPRE ? rpre : *reinterpret_cast<const uint4*>(R_ + (size_t)gm * p.ldr + gn), r);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += r[j];
  } else if (p.epi == COCODR_EPI_DGELU) {
    float r[8];
    unpack8(RPRE ? rpre : *reinterpret_cast<const uint4*>(R_ + (size_t)gm * p.ldr + gn), r);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= r[j];
  }
#if defined(COCODR_ABL_EPI_NOSTORE)  // ablation: everything but the result store
  asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
  return;
#endif
  if (OUT_F32) {
    float* C = reinterpret_cast<float*>(p.C) + (size_t)z * p.strideC + (size_t)gm * p.ldc + gn;
    *reinterpret_cast<float4*>(C) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(C + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    uint16_t* C = reinterpret_cast<uint16_t*>(p.C) + (size_t)z * p.strideC + (size_t)gm * p.ldc + gn;
    *reinterpret_cast<uint4*>(C) = pack8(v);
  }
}


}  // namespace cocodr_gemm_v2
