// Fused self-attention forward / backward for gfx950, head_dim = 64, key-padding mask only
// (hf: BertSelfAttention + eager_attention_forward, modeling_bert.py:111-203).
//
// Data layout: the fused QKV projection output [B*L, 3H] (Q | K | V column blocks) is read in
// place; one workgroup stages the K/V (and for the backward Q/dO) rows of ONE (batch, head) into
// swizzled LDS tiles and its 4 waves each own 32 query rows (or 32 keys in the dK/dV phase).
//
// MFMA use (v_mfma_f32_32x32x16_bf16, 64-lane waves):
//  * scores are computed TRANSPOSED, S^T = K Q^T, so a lane holds one query column: the softmax
//    max / sum are lane-local plus one exchange with lane^32;
//  * the fp32 S^T accumulator registers are converted to bf16 and fed straight back as the B
//    operand of O^T = V^T P^T: which 8 keys a lane contributes per MFMA is fixed by the
//    accumulator layout, and the matching rows of V are fetched with the transposing LDS read
//    (ds_read_b64_tr_b16), so P never goes through LDS or cross-lane shuffles.
#include <stdlib.h>
#include <algorithm>
#include "common.h"
#include "prof.h"

namespace {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kScale = 0.125f;  // 1/sqrt(64)
constexpr float kMaskNeg = -1e30f;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// B/A operand fragment of a row-major [rows][64] tile: row r0 + (l & 31), chunk 2s + (l >> 5)
__device__ __forceinline__ bf16x8 frag_rows(const char* t, int r0, int s, int lane) {
  return lds_read_b128(t, tile64_off(r0 + (lane & 31), 2 * s + (lane >> 5)));
}
// A operand = transpose of a [rows][64] tile: output row i = column dt*32 + (l & 31), contraction
// slots = tile rows rbase + 4*(l>>5) + {0..3} and + 8 + {0..3}  (the accumulator-register order of
// a 32x32 MFMA result, see file header).
__device__ __forceinline__ bf16x8 frag_cols_tr(const char* t, int rbase, int dt, int lane) {
  const int g = lane >> 4, c = lane & 15;
  const int col = dt * 32 + ((g & 1) << 4) + ((c & 3) << 2);
  const int row = rbase + ((g >> 1) << 2) + (c >> 2);
  const int within = (col & 7) << 1;
  const s16x4 a = lds_read_tr16(t, tile64_off(row, col >> 3) + within);
  const s16x4 b = lds_read_tr16(t, tile64_off(row + 8, col >> 3) + within);
  return join_tr(a, b);
}
// accumulator registers j*8 .. j*8+7 -> bf16x8 B operand
__device__ __forceinline__ bf16x8 pack_acc(const float* p, int j) {
  float t[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = p[j * 8 + e];
  return as_bf16x8(pack8(t));
}

// Same tile, fetched with the LDS DMA (buffer_load ... lds, 1 KiB = 8 rows per wave instruction, no VGPR round trip).
// The DMA writes lane-linear, so the chunk swizzle is applied to the per-lane SOURCE address (the XOR is an involution).
// Wave `wid` of `nw` issues pieces wid, wid + nw, ...; completion = s_waitcnt vmcnt(0) + barrier by the caller.
// Le <= L rows exist (a packed sequence whose extent is not a multiple of 32): the descriptor ends behind row Le - 1, so the rows
// of the last block that belong to the NEXT sequence (or lie behind the buffer) arrive as zeros through the bounds check.
__device__ __forceinline__ void stage_tile_dma(char* dst, const uint16_t* src, int ld, int L, int lane, int wid, int nw, int Le) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (uint32_t)(((size_t)(Le - 1) * ld + 64) * 2), 0x00020000);
  for (int p = wid; p < L / 8; p += nw) {
    const int q = p * 64 + lane;
    const int row = q >> 3, ch = (q & 7) ^ swz64(row);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_PTR(void))(dst + p * 1024), 16, (uint32_t)((row * ld + ch * 8) * 2), 0, 0, 0);
  }
#endif
}

// O^T / dQ^T / dK^T / dV^T accumulator (lane: row index l & 31, 4 consecutive d per register group).  Lanes l and
// l ^ 32 hold the two 4-wide halves of an 8-column group: they trade halves so that every lane issues 16-B stores.
// rows: how many of the block's 32 rows exist in this sequence (>= 32: all)
__device__ __forceinline__ void store_acc_T16(uint16_t* dst, int ld, const f32x16 (&o)[2], float mul, int lane, int rows) {
  const int half = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {  // register-group pair (2 rp, 2 rp + 1): half 0 stores group 2 rp, half 1 stores 2 rp + 1
      float mine[2][4];
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) mine[g][e] = o[dt][(2 * rp + g) * 4 + e] * mul;
      // both groups are packed with static indices and the packed words selected by lane half: indexing mine[half]
      // turns into an eight-way select chain per element (hipcc cannot index registers by a per-lane value)
      const uint2 g0 = pack4(mine[0]), g1 = pack4(mine[1]);
      const uint2 keep = half ? g1 : g0;           // my 4 columns of the group I store
      const uint2 give = half ? g0 : g1;           // my 4 columns of the group the partner stores
      uint2 got;
      got.x = __shfl_xor((int)give.x, 32, 64);
      got.y = __shfl_xor((int)give.y, 32, 64);
      const int d = dt * 32 + 8 * (2 * rp + half);  // columns d .. d+7: half-0 lane owns d..d+3, half-1 lane d+4..d+7
      const uint4 v = half == 0 ? make_uint4(keep.x, keep.y, got.x, got.y) : make_uint4(got.x, got.y, keep.x, keep.y);
      if ((lane & 31) < rows) *reinterpret_cast<uint4*>(dst + (size_t)(lane & 31) * ld + d) = v;
    }
}

// 8-byte store form (the forward: the 16-byte form costs it a wave of occupancy, 152 vs 94 registers, and measures slower)
__device__ __forceinline__ void store_acc_T(uint16_t* dst, int ld, const f32x16 (&o)[2], float mul, int lane, int rows) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      float t[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = o[dt][rg * 4 + e] * mul;
      const int d = dt * 32 + 8 * rg + 4 * (lane >> 5);
      if ((lane & 31) < rows) *reinterpret_cast<uint2*>(dst + (size_t)(lane & 31) * ld + d) = pack4(t);
    }
}

// ---- dropout on the attention probabilities (include/cocodr.h "Dropout"): element (b, h, q, k) has flat index
// ((b heads + h) L + q) L + k; pairs run along k.  Masks are regenerated wherever P is formed, never stored.
__device__ __forceinline__ uint32_t prob_row_pair(int bh, int q, int L) { return (uint32_t)((((uint64_t)bh * L + q) * L) >> 1); }
// lane = query layout (S^T accumulators): register rg*4 + e is key k0 + 8 rg + 4 half + e of this lane's row;
// pairbase = pair of (row, k0 + 4 half).  f(r, keep) with r a compile-time register index.
template <class F>
__device__ __forceinline__ void for_keep_qlane(uint32_t pairbase, const cocodr_dropout_mask& dm, F&& f) {
#pragma unroll
  for (int rg = 0; rg < 4; ++rg)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t w = drop_word(pairbase + rg * 4 + j, dm.k0, dm.k1);
      f(rg * 4 + 2 * j, drop_keep_lo(w, dm.threshold));
      f(rg * 4 + 2 * j + 1, drop_keep_hi(w, dm.threshold));
    }
}
// lane = key layout (S accumulators): register rg*4 + e is query q0 + 8 rg + 4 half + e, this lane's key is fixed;
// pairbase = pair of (q0 + 4 half, key), Lh = L / 2 pairs per row, shift = 16 for odd keys
template <class F>
__device__ __forceinline__ void for_keep_klane(uint32_t pairbase, uint32_t Lh, uint32_t shift, const cocodr_dropout_mask& dm, F&& f) {
#pragma unroll
  for (int rg = 0; rg < 4; ++rg)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t w = drop_word(pairbase + (uint32_t)(8 * rg + e) * Lh, dm.k0, dm.k1);
      f(rg * 4 + e, ((w >> shift) & 0xffffu) >= dm.threshold);
    }
}

// Packed (variable-length) batches: sequence b occupies rows [seq_off[b], seq_off[b + 1]) of the [T, .] activations - an extent
// of Le >= 1 rows, ANY number (rows behind a sequence's tokens inside its extent are padding with mask 0); lse is [heads, T].
// The kernels walk L = ceil32(Le) rows in 32-row blocks: rows >= Le of the last block (the next sequence's, or behind the
// buffer) are read as zeros / not read, masked as keys, given a probability of exactly 0 as queries, and never stored.
// seq_off == NULL: the padded layout, B sequences of Le = L = Lmax rows.  drop_L: the padded length the dropout indices are defined on, so a packed
// and a padded run of the same batch draw the same masks.
constexpr float kLseNoRow = 1.0e30f;  // log-sum-exp (in log2 units) of a query row that does not exist: its probabilities are exactly 0
// order (optional): the sequence the y-th workgroup row takes - the host passes the sequences longest first, so that the
// workgroups of the last, partial round over the CUs are the cheap ones (a workgroup costs ~ its number of 32-row blocks squared).
struct AttnPacked {
  const int32_t* seq_off;
  int T, drop_L;
  const int32_t* order;
};
#define ATTN_EXTENT(Lmax, pk)                                                                     \
  const int h = blockIdx.x, b = (pk).order ? (pk).order[blockIdx.y] : (int)blockIdx.y, heads = gridDim.x; \
  int L = (Lmax), Le = (Lmax), dropL = (Lmax);                                                    \
  size_t row0 = (size_t)b * (Lmax), lse0 = ((size_t)b * heads + h) * (Lmax);                      \
  if ((pk).seq_off) {                                                                             \
    const int o_ = (pk).seq_off[b];                                                               \
    Le = (pk).seq_off[b + 1] - o_;                                                                \
    L = (Le + 31) & ~31;                                                                          \
    row0 = (size_t)o_;                                                                            \
    lse0 = (size_t)h * (pk).T + o_;                                                               \
    dropL = (pk).drop_L;                                                                          \
  }

// One wave's 32 query rows q0 .. q0 + 31 of one (sequence, head) against the K / V tiles in LDS: online softmax over the L / 32 key
// blocks, O^T accumulated in registers, result rows and log-sum-exp stored (rows behind the extent Le are computed and not stored).
template <bool DROP>
__device__ __forceinline__ void attn_fwd_rows(const char* Kt, const char* Vt, const float* madd, const bf16x8 (&qf)[4], int L, int Le, int q0,
                                              int lane, int bh, int dropL, const cocodr_dropout_mask& dm, uint16_t* out, int H, float* lse_out) {
  const int half = lane >> 5;
  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m = kMaskNeg, lsum = 0.f;
  const float sl2 = kScale * kLog2e;
  const uint32_t rowpair = DROP ? prob_row_pair(bh, q0 + (lane & 31), dropL) + 2 * half : 0u;
  for (int kb = 0; kb < L / 32; ++kb) {
    f32x16 sacc;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {  // S starts at the key mask: the MFMAs add K.Q on top
      const float4 ma = *reinterpret_cast<const float4*>(madd + kb * 32 + 8 * rg + 4 * half);
      sacc[rg * 4 + 0] = ma.x; sacc[rg * 4 + 1] = ma.y; sacc[rg * 4 + 2] = ma.z; sacc[rg * 4 + 3] = ma.w;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Kt, kb * 32, s, lane), qf[s], sacc, 0, 0, 0);
    float p[16];
    float bmax = sacc[0];  // block maximum on the raw scores (the scale is positive), one multiply per row afterwards
#pragma unroll
    for (int r = 1; r < 16; ++r) bmax = fmaxf(bmax, sacc[r]);
    bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
    const float mnew = fmaxf(m, bmax * sl2);
    const float alpha = fast_exp2(m - mnew);
    m = mnew;
    lsum *= alpha;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = fast_exp2(fmaf(sacc[r], sl2, -mnew));
      lsum += p[r];
    }
    // the normaliser sums the un-dropped probabilities; the 1 / (1 - p) scale rides on the final 1 / l
    if constexpr (DROP) for_keep_qlane(rowpair + kb * 16, dm, [&](int r, bool keep) { p[r] = keep ? p[r] : 0.f; });
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bf16x8 pf = pack_acc(p, j);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_tr(Vt, kb * 32 + j * 16, dt, lane), pf, o[dt], 0, 0, 0);
    }
  }
  const float ltot = lsum + __shfl_xor(lsum, 32, 64);
  if (lane < 32 && q0 + lane < Le) lse_out[lane] = (m + __log2f(ltot)) * kLn2;
  store_acc_T(out, H, o, (DROP ? dm.scale : 1.0f) / ltot, lane, Le - q0);
}


// three waves per SIMD (168 registers): at the default bound hipcc parks the O accumulators in AGPRs and pays an
// accvgpr read + write per element for every online-softmax rescale
template <bool DROP>
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ mask,
                                                       uint16_t* __restrict__ ctx, float* __restrict__ lse, int Lmax, int H,
                                                       const cocodr_dropout_mask dm, const AttnPacked pk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ATTN_EXTENT(Lmax, pk)
  if ((int)blockIdx.z * 128 >= L) return;  // (packed batches: a sequence shorter than the longest has fewer query blocks)
  char* Kt = smem;
  char* Vt = smem + L * 128;
  float* madd = reinterpret_cast<float*>(smem + 2 * L * 128);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ld = 3 * H;
  const uint16_t* base = qkv + row0 * ld + h * 64;
  stage_tile_dma(Kt, base + H, ld, L, lane, wid, 4, Le);
  stage_tile_dma(Vt, base + 2 * H, ld, L, lane, wid, 4, Le);
  // additive key mask in units of the raw q.k scores; it seeds the S accumulators, so no add per element later.  -2e5 raw
  // = -3.6e4 in the exponent: exp2 underflows to an exact 0 against any real score, and a row whose keys are ALL masked
  // still gets a finite softmax over its raw scores (what adding finfo.min to every key gives the reference); a seed
  // of -1e30 would leave the fma below with a rounding residue of ~1e22 there.
  for (int i = tid; i < L; i += 256) madd[i] = (i < Le && mask[row0 + i] != 0) ? 0.f : -2.0e5f;
  const int q0 = blockIdx.z * 128 + wid * 32;
  const int half = lane >> 5;
  bf16x8 qf[4];
  if (q0 < L) {
    const bool qok = q0 + (lane & 31) < Le;  // (query rows behind the extent: zeros, computed and not stored)
#pragma unroll
    for (int s = 0; s < 4; ++s)
      qf[s] = as_bf16x8(qok ? *reinterpret_cast<const uint4*>(base + (size_t)(q0 + (lane & 31)) * ld + (2 * s + half) * 8) : make_uint4(0u, 0u, 0u, 0u));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (q0 >= L) return;

  attn_fwd_rows<DROP>(Kt, Vt, madd, qf, L, Le, q0, lane, b * heads + h, dropL, dm, ctx + (row0 + q0) * H + h * 64, H, lse + lse0 + q0);
}

#if defined(COCODR_ABL_TIMELINE)  // tools/attn_timeline.py builds: per-workgroup phase stamps (100 MHz wall clock)
__device__ unsigned long long* g_attn_tl = nullptr;
#define ATTN_STAMP(i) do { if (g_attn_tl && threadIdx.x == 0) g_attn_tl[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define ATTN_STAMP(i) do { } while (0)
#endif

// Backward.  All four [L][64] tiles of one (batch, head) live in LDS (L <= 256 -> 128 KiB).
// Column sums of a transposed [32 rows x 64 columns] fp32 accumulator tile (the query / key bias gradient).  A lane holds
// 32 columns of one row.  Per 16-column group the lanes of a 16-lane row pair up (lane ^ 8, then 7 - lane within 8 lanes):
// both partners get the pair sums and keep the half of the value list their own lane bit selects; the quads then sum
// their four remaining entries and each lane keeps one.  All of it is DPP operands of v_add_f32 - no LDS traffic - and
// one cross-row add (lane ^ 16) finishes a group, after which lane l owns the sum of ONE column: list index
// c = l & 31 (bit 4 = group) -> column qk_col(l).
#ifdef COCODR_ABL_QK_NOSUM
constexpr bool QK_ABL_NOSUM = true;
#else
constexpr bool QK_ABL_NOSUM = false;
#endif
// r = own + partner's value of the same list entry, where the quads in bank mask LO take entry a and the quads in HI take
// entry b: two DPP adds whose bank masks do the selecting (hipcc does not fold a bank-masked update_dpp into the add, and
// inline asm is invisible to its hazard recognizer: the s_nop covers the two wait states between a VALU write and a DPP read)
#define COLSUM_PAIR(r, a, b, CTRL, LO, HI)                                                   \
  asm volatile("s_nop 1\n\t"                                                                 \
               "v_add_f32_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:" LO "\n\t"           \
               "v_add_f32_dpp %0, %2, %2 " CTRL " row_mask:0xf bank_mask:" HI                  \
               : "=&v"(r) : "v"(a), "v"(b))
#define COLSUM_QUAD(x, CTRL) \
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(x))
__device__ __forceinline__ float acc_colsum32(const f32x16 (&o)[2], float mul, int lane) {
  float out[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    float v[8], w[4];
#pragma unroll
    for (int i = 0; i < 8; ++i)  // partner lane ^ 8; lanes 8..15 of a row (quads 2, 3) keep the upper half of the list
      COLSUM_PAIR(v[i], o[dt][i], o[dt][i + 8], "row_ror:8", "0x3", "0xc");
#pragma unroll
    for (int i = 0; i < 4; ++i)  // partner 7 - lane within 8 lanes (flips bit 2); quads 1, 3 keep the upper half
      COLSUM_PAIR(w[i], v[i], v[i + 4], "row_half_mirror", "0x5", "0xa");
    // the four lanes of a quad now hold the same four columns: full quad sums, then every lane keeps column (lane & 3)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      COLSUM_QUAD(w[i], "quad_perm:[1,0,3,2]");
      COLSUM_QUAD(w[i], "quad_perm:[2,3,0,1]");
    }
    const float a = (lane & 1) ? w[1] : w[0], c = (lane & 1) ? w[3] : w[2];
    const float q = (lane & 2) ? c : a;
    out[dt] = q + __shfl_xor(q, 16, 64);
  }
  return ((lane & 16) ? out[1] : out[0]) * mul;
}
__device__ __forceinline__ int qk_col(int lane) {  // list index dt*16 + rg*4 + e of half (lane >> 5) -> d = dt*32 + 8*rg + 4*half + e
  const int c = lane & 31;
  return (c >> 4) * 32 + ((c >> 2) & 3) * 8 + 4 * (lane >> 5) + (c & 3);
}
// every wave leaves its own row: partial is [B][4 waves][2H] floats (query half | key half), no LDS pass and no barrier at
// the end of the workgroup; a wave without a query block of its own writes zeros.
// The key half is written as exact zeros: sum_k dK[k] = scale * sum_q Q[q] * (sum_k dS[q][k]) and every row of dS sums to
// P.dP - delta * sum(P) = 0 (softmax shift invariance: the scores do not depend on the key bias).  The fp32 reference gets
// rounding noise ~1e-7 of the other gradients there, far below Adam's eps; summing the bf16 pipeline's dK would instead
// hand the optimizer noise it treats as a gradient.
__device__ __forceinline__ void qk_bias_store(float* partial, float qacc, int b, int h, int H, int tid) {
  float* row = partial + (size_t)(b * 4 + (tid >> 6)) * 2 * H + h * 64 + qk_col(tid & 63);
  row[0] = qacc;
  row[H] = 0.f;
}
//  phase A: wave <-> 32 queries,  S^T/dP^T layout (lane = query):  dQ^T += K^T dS^T
//  phase B: wave <-> 32 keys,     S / dP layout   (lane = key):    dV^T += dO^T P,  dK^T += Q^T dS
// DROP (probabilities dropped in the forward, mask m, scale s): dV = (s m P)^T dO and dS = P (s m dP' - delta) with
// dP' = dO V^T, delta = rowsum(dO O) as before (O already carries the mask).  The dP accumulators are seeded with
// -delta / s, so dS / s = P (kept ? acc : seed): one select per element, and s rides on the final scales of dQ, dK, dV.
template <bool QKSUM, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ mask,
                                                          const uint16_t* __restrict__ ctx, const uint16_t* __restrict__ dctx,
                                                          const float* __restrict__ lse, uint16_t* __restrict__ dqkv, int Lmax,
                                                          int H, int stagger, float* __restrict__ qk_partial,
                                                          const cocodr_dropout_mask dm, const AttnPacked pk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ATTN_EXTENT(Lmax, pk)
  char* Qt = smem;
  char* Kt = smem + L * 128;
  char* Vt = smem + 2 * L * 128;
  char* Dt = smem + 3 * L * 128;
  float qacc = 0.f;
  float* madd = reinterpret_cast<float*>(smem + 4 * L * 128);
  float* lse2 = madd + L;
  float* delta = lse2 + L;
  const float inv_s = DROP ? 1.0f / dm.scale : 1.0f, out_s = DROP ? dm.scale : 1.0f;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ld = 3 * H;
  const uint16_t* base = qkv + row0 * ld + h * 64;
  const uint16_t* obase = ctx + row0 * H + h * 64;
  const uint16_t* dobase = dctx + row0 * H + h * 64;
  // Two workgroups share a CU and all of them take the same time: un-staggered, both stage (an HBM burst of the whole
  // grid) and then both compute, round after round.  The first-round workgroup in the upper LDS slot starts
  // `stagger` x 4096 clocks late once, so from then on one workgroup's loads run under the other's MFMAs.
  if (stagger > 0 && (int)(blockIdx.y * gridDim.x + blockIdx.x) < 512 && __builtin_amdgcn_s_getreg(6 | (11 << 11)) != 0)
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(64);
  ATTN_STAMP(0);
  // Q, K, V go HBM -> LDS by DMA; dO and O pass through registers (delta = rowsum(dO * O) needs them there anyway).
  // Every global load of the prologue is in flight before the first one is consumed.
  stage_tile_dma(Qt, base, ld, L, lane, wid, 4, Le);
  stage_tile_dma(Kt, base + H, ld, L, lane, wid, 4, Le);
  stage_tile_dma(Vt, base + 2 * H, ld, L, lane, wid, 4, Le);
  constexpr int kMaxIt = 8;  // L <= 256 -> L * 8 / 256 <= 8 chunks per thread
  const int nit = L * 8 / 256;  // whole waves stay converged (L % 32 == 0)
  uint4 dreg[kMaxIt], oreg[kMaxIt];
#pragma unroll
  for (int i = 0; i < kMaxIt; ++i)
    if (i < nit) {
      const int q = tid + i * 256, row = q >> 3, ch = q & 7;
      dreg[i] = oreg[i] = make_uint4(0u, 0u, 0u, 0u);
      if (row < Le) {
        dreg[i] = *reinterpret_cast<const uint4*>(dobase + (size_t)row * H + ch * 8);
        oreg[i] = *reinterpret_cast<const uint4*>(obase + (size_t)row * H + ch * 8);
      }
    }
  for (int i = tid; i < L; i += 256) {
    madd[i] = (i < Le && mask[row0 + i] != 0) ? 0.f : kMaskNeg;
    lse2[i] = i < Le ? lse[lse0 + i] * kLog2e : kLseNoRow;  // (a query row behind the extent: P = exp2(.. - 1e30) = 0, so dS = 0)
  }
#pragma unroll
  for (int i = 0; i < kMaxIt; ++i)
    if (i < nit) {
      const int q = tid + i * 256, row = q >> 3, ch = q & 7;
      *reinterpret_cast<uint4*>(Dt + tile64_off(row, ch)) = dreg[i];
      float df[8], of[8];
      unpack8(dreg[i], df);
      unpack8(oreg[i], of);
      float part = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) part += df[e] * of[e];
      part += dpp_partner<0xB1>(part);   // lane ^ 1
      part += dpp_partner<0x4E>(part);   // lane ^ 2
      part += dpp_partner<0x141>(part);  // the other quad of the 8 lanes that hold a row (DPP operands, no LDS round trip)
      if (ch == 0) delta[row] = DROP ? -part * inv_s : -part;  // negated: it seeds the dP accumulators below
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ATTN_STAMP(1);

  const int half = lane >> 5;
  const float sl2 = kScale * kLog2e;
  const int nblk = L / 32;

  // ---------------- phase A: dQ
  for (int qb = wid; qb < nblk; qb += 4) {
    bf16x8 qf[4], dof[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = frag_rows(Qt, qb * 32, s, lane);
      dof[s] = frag_rows(Dt, qb * 32, s, lane);
    }
    const float my_lse = lse2[qb * 32 + (lane & 31)];
    const float my_ndelta = delta[qb * 32 + (lane & 31)];  // -delta of this lane's query (DROP: / s)
    const uint32_t rowpair = DROP ? prob_row_pair(b * heads + h, qb * 32 + (lane & 31), dropL) + 2 * half : 0u;
    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    for (int kb = 0; kb < nblk; ++kb) {
      f32x16 sacc, dpacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = my_ndelta; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Kt, kb * 32, s, lane), qf[s], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Vt, kb * 32, s, lane), dof[s], dpacc, 0, 0, 0);
      }
      float ds[16];
      [[maybe_unused]] float pa[DROP ? 16 : 1];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float4 ma = *reinterpret_cast<const float4*>(madd + kb * 32 + 8 * rg + 4 * half);
        const float mm[4] = {ma.x, ma.y, ma.z, ma.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = rg * 4 + e;
          const float p = fast_exp2(sacc[r] * sl2 + mm[e] - my_lse);
          ds[r] = p * dpacc[r];  // dpacc was seeded with -delta
          if constexpr (DROP) pa[r] = p;
        }
      }
      if constexpr (DROP)  // dropped elements: dP = 0, i.e. dS / s = P * seed
        for_keep_qlane(rowpair + kb * 16, dm, [&](int r, bool keep) { ds[r] = keep ? ds[r] : pa[r] * my_ndelta; });
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 dsf = pack_acc(ds, j);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_tr(Kt, kb * 32 + j * 16, dt, lane), dsf, dq[dt], 0, 0, 0);
      }
    }
    ATTN_STAMP(2);
    // (in front of the store: behind it hipcc interleaves the two and spills 72 SGPRs of lane masks instead of 9)
    if constexpr (QKSUM && !QK_ABL_NOSUM) qacc += acc_colsum32(dq, kScale * out_s, lane);
    store_acc_T16(dqkv + (row0 + qb * 32) * ld + h * 64, ld, dq, kScale * out_s, lane, Le - qb * 32);
  }
  ATTN_STAMP(3);
  if constexpr (QKSUM) qk_bias_store(qk_partial, qacc, b, h, H, tid);

  // ---------------- phase B: dK, dV
  for (int kb = wid; kb < nblk; kb += 4) {
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = frag_rows(Kt, kb * 32, s, lane);
      vf[s] = frag_rows(Vt, kb * 32, s, lane);
    }
    const float my_madd = madd[kb * 32 + (lane & 31)];
    // pair of (query 4 half, this lane's key); a query step is L / 2 pairs
    const uint32_t Lh = (uint32_t)dropL >> 1, kshift = (lane & 1) << 4;
    const uint32_t keypair = DROP ? prob_row_pair(b * heads + h, 4 * half, dropL) + (uint32_t)((kb * 32 + (lane & 31)) >> 1) : 0u;
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    for (int qb = 0; qb < nblk; ++qb) {
      f32x16 sacc, dpacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {  // dP accumulators start at -delta of their query rows
        const float4 d4 = *reinterpret_cast<const float4*>(delta + qb * 32 + 8 * rg + 4 * half);
        dpacc[rg * 4 + 0] = d4.x; dpacc[rg * 4 + 1] = d4.y; dpacc[rg * 4 + 2] = d4.z; dpacc[rg * 4 + 3] = d4.w;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Qt, qb * 32, s, lane), kf[s], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Dt, qb * 32, s, lane), vf[s], dpacc, 0, 0, 0);
      }
      float p[16], ds[16];
      [[maybe_unused]] float sd[DROP ? 16 : 1];  // dS / s of a dropped element: P * seed
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float4 l4 = *reinterpret_cast<const float4*>(lse2 + qb * 32 + 8 * rg + 4 * half);
        const float ll[4] = {l4.x, l4.y, l4.z, l4.w};
        [[maybe_unused]] float dseed[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (DROP) {
          const float4 d4 = *reinterpret_cast<const float4*>(delta + qb * 32 + 8 * rg + 4 * half);
          dseed[0] = d4.x; dseed[1] = d4.y; dseed[2] = d4.z; dseed[3] = d4.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = rg * 4 + e;
          p[r] = fast_exp2(sacc[r] * sl2 + my_madd - ll[e]);
          ds[r] = p[r] * dpacc[r];  // dpacc was seeded with -delta
          if constexpr (DROP) sd[r] = p[r] * dseed[e];
        }
      }
      if constexpr (DROP)
        for_keep_klane(keypair + (uint32_t)(qb * 32) * Lh, Lh, kshift, dm, [&](int r, bool keep) {
          ds[r] = keep ? ds[r] : sd[r];
          p[r] = keep ? p[r] : 0.f;
        });
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 pf = pack_acc(p, j), dsf = pack_acc(ds, j);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_tr(Dt, qb * 32 + j * 16, dt, lane), pf, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_tr(Qt, qb * 32 + j * 16, dt, lane), dsf, dk[dt], 0, 0, 0);
        }
      }
    }
    ATTN_STAMP(4);
    uint16_t* out0 = dqkv + (row0 + kb * 32) * ld + h * 64;
    store_acc_T16(out0 + H, ld, dk, kScale * out_s, lane, Le - kb * 32);
    store_acc_T16(out0 + 2 * H, ld, dv, out_s, lane, Le - kb * 32);
  }
  ATTN_STAMP(5);
}


// ---------------------------------------------------------------------------------------------------------------
// L <= 128 (at most four key blocks = one per wave): ONE pass over the (query block, key block) pairs.  The two-phase kernel
// above forms S and dP - and evaluates the softmax arithmetic: exp2 at quarter rate, the dS product, the bf16 packs - twice, once
// per layout (28 MFMAs per block pair, 16 exp2 per lane twice); what fills the SIMDs there is that arithmetic, not the MFMAs
// (profiles/r03_attention_pipelining.md).  Here every wave owns ONE key block (lane = key) and walks the query blocks once:
//   S, dP (8 MFMAs) -> P, dS -> dV^T += dO^T P, dK^T += Q^T dS (8 MFMAs) -> dS (bf16) into an LDS stage tile [keys][64 queries]
// and after every two query blocks the waves meet, and dQ^T = K^T dS^T for those 64 queries is 4 MFMAs per block pair with the
// dS operand read back through the transposing LDS read (wave = (query block of the pair, 32-column half of d)): 20 MFMAs per
// block pair, the softmax arithmetic once.  dS needs no cross-wave reduction and no atomics: deterministic.
// LDS: Q, K, dO tiles + the dS stage tile (4 x L x 128 B) + lse / delta rows - what the two-phase kernel takes, so two workgroups
// share a CU as before; V is not staged (a wave reads the four fragments of its own key block from global memory once).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float acc_colsum16(const f32x16& acc, float mul, int lane, int dt) {  // acc_colsum32 for one 32-column half
  // The accumulator arrives straight from the LAST MFMA of a dependent chain, and hipcc's hazard recognizer does not look inside
  // an asm statement: a DPP read there would see registers the matrix pipe has not written back yet (measured: the list
  // entries read first came out stale).  An ordinary VALU instruction - the scale - takes the MFMA -> VALU wait states instead.
  float o[16], v[8], w[4];
#pragma unroll
  for (int i = 0; i < 16; ++i) o[i] = acc[i] * mul;
  mul = 1.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) COLSUM_PAIR(v[i], o[i], o[i + 8], "row_ror:8", "0x3", "0xc");
#pragma unroll
  for (int i = 0; i < 4; ++i) COLSUM_PAIR(w[i], v[i], v[i + 4], "row_half_mirror", "0x5", "0xa");
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    COLSUM_QUAD(w[i], "quad_perm:[1,0,3,2]");
    COLSUM_QUAD(w[i], "quad_perm:[2,3,0,1]");
  }
  const float a = (lane & 1) ? w[1] : w[0], c = (lane & 1) ? w[3] : w[2];
  const float q = (lane & 2) ? c : a;
  const float out = q + __shfl_xor(q, 16, 64);
  return (((lane >> 4) & 1) == dt) ? out * mul : 0.f;  // the lanes that own a column of this half (qk_col)
}
__device__ __forceinline__ void store_acc_T16_half(uint16_t* dst, int ld, const f32x16& o, int dt, float mul, int lane, int rows) {
  const int half = lane >> 5;
#pragma unroll
  for (int rp = 0; rp < 2; ++rp) {
    float mine[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) mine[g][e] = o[(2 * rp + g) * 4 + e] * mul;
    const uint2 g0 = pack4(mine[0]), g1 = pack4(mine[1]);
    const uint2 keep = half ? g1 : g0;
    const uint2 give = half ? g0 : g1;
    uint2 got;
    got.x = __shfl_xor((int)give.x, 32, 64);
    got.y = __shfl_xor((int)give.y, 32, 64);
    const int d = dt * 32 + 8 * (2 * rp + half);
    const uint4 v = half == 0 ? make_uint4(keep.x, keep.y, got.x, got.y) : make_uint4(got.x, got.y, keep.x, keep.y);
    if ((lane & 31) < rows) *reinterpret_cast<uint4*>(dst + (size_t)(lane & 31) * ld + d) = v;
  }
}

template <bool QKSUM, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd1_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ mask,
                                                           const uint16_t* __restrict__ ctx, const uint16_t* __restrict__ dctx,
                                                           const float* __restrict__ lse, uint16_t* __restrict__ dqkv, int Lmax,
                                                           int H, int stagger, float* __restrict__ qk_partial,
                                                           const cocodr_dropout_mask dm, const AttnPacked pk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ATTN_EXTENT(Lmax, pk)
  char* Qt = smem;
  char* Kt = smem + L * 128;
  char* Dt = smem + 2 * L * 128;
  char* St = smem + 3 * L * 128;  // dS of the current pair of query blocks: [L keys][64 queries] bf16, tile64 layout
  float* lse2 = reinterpret_cast<float*>(smem + 4 * L * 128);
  float* delta = lse2 + L;
  float qacc = 0.f;
  const float inv_s = DROP ? 1.0f / dm.scale : 1.0f, out_s = DROP ? dm.scale : 1.0f;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ld = 3 * H;
  const uint16_t* base = qkv + row0 * ld + h * 64;
  const uint16_t* obase = ctx + row0 * H + h * 64;
  const uint16_t* dobase = dctx + row0 * H + h * 64;
  if (stagger > 0 && (int)(blockIdx.y * gridDim.x + blockIdx.x) < 512 && __builtin_amdgcn_s_getreg(6 | (11 << 11)) != 0)
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(64);
  const int half = lane >> 5;
  const int nblk = L / 32;
  const int kb = wid;               // this wave's key block
  const bool own = kb < nblk;       // (a short sequence of a packed batch has fewer key blocks than the workgroup has waves)
  stage_tile_dma(Qt, base, ld, L, lane, wid, 4, Le);
  stage_tile_dma(Kt, base + H, ld, L, lane, wid, 4, Le);
  // V fragments and the mask of this wave's own keys straight from global memory (layout of frag_rows)
  uint4 vraw[4] = {};
  float my_madd = 0.f;
  if (own) {
    const bool kok = kb * 32 + (lane & 31) < Le;  // (key rows behind the extent: zero V, masked)
    const uint16_t* vrow = base + 2 * H + (size_t)(kb * 32 + (lane & 31)) * ld;
    if (kok) {
#pragma unroll
      for (int s = 0; s < 4; ++s) vraw[s] = *reinterpret_cast<const uint4*>(vrow + (2 * s + half) * 8);
    }
    my_madd = (kok && mask[row0 + kb * 32 + (lane & 31)] != 0) ? 0.f : kMaskNeg;
  }
  constexpr int kMaxIt = 4;  // L <= 128 -> L * 8 / 256 <= 4 chunks per thread
  const int nit = L * 8 / 256;
  uint4 dreg[kMaxIt], oreg[kMaxIt];
#pragma unroll
  for (int i = 0; i < kMaxIt; ++i)
    if (i < nit) {
      const int q = tid + i * 256, row = q >> 3, ch = q & 7;
      dreg[i] = oreg[i] = make_uint4(0u, 0u, 0u, 0u);
      if (row < Le) {
        dreg[i] = *reinterpret_cast<const uint4*>(dobase + (size_t)row * H + ch * 8);
        oreg[i] = *reinterpret_cast<const uint4*>(obase + (size_t)row * H + ch * 8);
      }
    }
  for (int i = tid; i < L; i += 256) lse2[i] = i < Le ? lse[lse0 + i] * kLog2e : kLseNoRow;
#pragma unroll
  for (int i = 0; i < kMaxIt; ++i)
    if (i < nit) {
      const int q = tid + i * 256, row = q >> 3, ch = q & 7;
      *reinterpret_cast<uint4*>(Dt + tile64_off(row, ch)) = dreg[i];
      float df[8], of[8];
      unpack8(dreg[i], df);
      unpack8(oreg[i], of);
      float part = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) part += df[e] * of[e];
      part += dpp_partner<0xB1>(part);
      part += dpp_partner<0x4E>(part);
      part += dpp_partner<0x141>(part);
      if (ch == 0) delta[row] = DROP ? -part * inv_s : -part;  // negated: it seeds the dP accumulators
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const float sl2 = kScale * kLog2e;
  bf16x8 kf[4], vf[4];
  if (own) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = frag_rows(Kt, kb * 32, s, lane);
      vf[s] = as_bf16x8(vraw[s]);
    }
  }
  const uint32_t Lh = (uint32_t)dropL >> 1, kshift = (lane & 1) << 4;
  const uint32_t keypair = DROP ? prob_row_pair(b * heads + h, 4 * half, dropL) + (uint32_t)((kb * 32 + (lane & 31)) >> 1) : 0u;
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
  const int aq = wid & 1, adt = wid >> 1;  // dQ phase: this wave's query block of the pair and its 32-column half of d

  for (int c = 0; 2 * c < nblk; ++c) {
    if (own) {
#pragma unroll 1
      for (int qbl = 0; qbl < 2; ++qbl) {
        const int qb = 2 * c + qbl;
        if (qb >= nblk) break;
        f32x16 sacc, dpacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {  // dP accumulators start at -delta of their query rows
          const float4 d4 = *reinterpret_cast<const float4*>(delta + qb * 32 + 8 * rg + 4 * half);
          dpacc[rg * 4 + 0] = d4.x; dpacc[rg * 4 + 1] = d4.y; dpacc[rg * 4 + 2] = d4.z; dpacc[rg * 4 + 3] = d4.w;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Qt, qb * 32, s, lane), kf[s], sacc, 0, 0, 0);
          dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Dt, qb * 32, s, lane), vf[s], dpacc, 0, 0, 0);
        }
        float p[16], ds[16];
        [[maybe_unused]] float sd[DROP ? 16 : 1];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const float4 l4 = *reinterpret_cast<const float4*>(lse2 + qb * 32 + 8 * rg + 4 * half);
          const float ll[4] = {l4.x, l4.y, l4.z, l4.w};
          [[maybe_unused]] float dseed[4] = {0.f, 0.f, 0.f, 0.f};
          if constexpr (DROP) {
            const float4 d4 = *reinterpret_cast<const float4*>(delta + qb * 32 + 8 * rg + 4 * half);
            dseed[0] = d4.x; dseed[1] = d4.y; dseed[2] = d4.z; dseed[3] = d4.w;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = rg * 4 + e;
            p[r] = fast_exp2(sacc[r] * sl2 + my_madd - ll[e]);
            ds[r] = p[r] * dpacc[r];
            if constexpr (DROP) sd[r] = p[r] * dseed[e];
          }
        }
        if constexpr (DROP)
          for_keep_klane(keypair + (uint32_t)(qb * 32) * Lh, Lh, kshift, dm, [&](int r, bool keep) {
            ds[r] = keep ? ds[r] : sd[r];
            p[r] = keep ? p[r] : 0.f;
          });
        // dS of (this key, queries 8 rg + 4 half .. + 3 of block qbl) -> stage tile row = key, 8 bytes at column qbl 32 + 8 rg + 4 half
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const float t4[4] = {ds[rg * 4 + 0], ds[rg * 4 + 1], ds[rg * 4 + 2], ds[rg * 4 + 3]};
          *reinterpret_cast<uint2*>(St + tile64_off(kb * 32 + (lane & 31), qbl * 4 + rg) + 8 * half) = pack4(t4);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 pf = pack_acc(p, j), dsf = pack_acc(ds, j);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_tr(Dt, qb * 32 + j * 16, dt, lane), pf, dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_tr(Qt, qb * 32 + j * 16, dt, lane), dsf, dk[dt], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();  // the pair's dS tile is complete (every key block)
    if (2 * c + aq < nblk) {  // dQ^T[adt half of d, query block 2 c + aq] = sum over ALL keys
      f32x16 dq;
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[r] = 0.f;
      for (int k2 = 0; k2 < nblk; ++k2)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_tr(Kt, k2 * 32 + j * 16, adt, lane), frag_cols_tr(St, k2 * 32 + j * 16, aq, lane),
                                                       dq, 0, 0, 0);
      if constexpr (QKSUM && !QK_ABL_NOSUM) qacc += acc_colsum16(dq, kScale * out_s, lane, adt);
      store_acc_T16_half(dqkv + (row0 + (2 * c + aq) * 32) * ld + h * 64, ld, dq, adt, kScale * out_s, lane, Le - (2 * c + aq) * 32);
    }
    if (2 * (c + 1) < nblk) __syncthreads();  // the next pair overwrites the stage tile
  }
  if constexpr (QKSUM) qk_bias_store(qk_partial, qacc, b, h, H, tid);
  if (own) {
    uint16_t* out0 = dqkv + (row0 + kb * 32) * ld + h * 64;
    store_acc_T16(out0 + H, ld, dk, kScale * out_s, lane, Le - kb * 32);
    store_acc_T16(out0 + 2 * H, ld, dv, out_s, lane, Le - kb * 32);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 256 < L <= 512: the four [L,64] tiles no longer fit the LDS together, so the two phases become two kernels that
// each keep only the pair of tiles they sweep (K,V for dQ; Q,dO for dK/dV, 128 KiB at L = 512) and fetch the fragments
// of their own 32 rows straight from global memory.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 frag_rows_global(const uint16_t* base, int ld, int r0, int s, int lane, int Le) {  // rows >= Le: zeros
  const bool ok = r0 + (lane & 31) < Le;
  return as_bf16x8(ok ? *reinterpret_cast<const uint4*>(base + (size_t)(r0 + (lane & 31)) * ld + (2 * s + (lane >> 5)) * 8) : make_uint4(0u, 0u, 0u, 0u));
}

template <bool QKSUM, bool DROP>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ mask,
                                                             const uint16_t* __restrict__ ctx, const uint16_t* __restrict__ dctx,
                                                             const float* __restrict__ lse, uint16_t* __restrict__ dqkv, int Lmax, int H,
                                                             float* __restrict__ qk_partial, const cocodr_dropout_mask dm, const AttnPacked pk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ATTN_EXTENT(Lmax, pk)
  const float inv_s = DROP ? 1.0f / dm.scale : 1.0f, out_s = DROP ? dm.scale : 1.0f;
  float qacc = 0.f;
  char* Kt = smem;
  char* Vt = smem + L * 128;
  float* madd = reinterpret_cast<float*>(smem + 2 * L * 128);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, half = lane >> 5;
  const int ld = 3 * H;
  const uint16_t* base = qkv + row0 * ld + h * 64;
  const uint16_t* obase = ctx + row0 * H + h * 64;
  const uint16_t* dobase = dctx + row0 * H + h * 64;
  stage_tile_dma(Kt, base + H, ld, L, lane, wid, 4, Le);
  stage_tile_dma(Vt, base + 2 * H, ld, L, lane, wid, 4, Le);
  for (int i = tid; i < L; i += 256) madd[i] = (i < Le && mask[row0 + i] != 0) ? 0.f : kMaskNeg;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const float sl2 = kScale * kLog2e;
  const int nblk = L / 32;
  for (int qb = wid; qb < nblk; qb += 4) {
    bf16x8 qf[4], dof[4];
    float dpart = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = frag_rows_global(base, ld, qb * 32, s, lane, Le);
      dof[s] = frag_rows_global(dobase, H, qb * 32, s, lane, Le);
      const bf16x8 of = frag_rows_global(obase, H, qb * 32, s, lane, Le);
      float df[8], ofv[8];
      unpack8(__builtin_bit_cast(uint4, dof[s]), df);
      unpack8(__builtin_bit_cast(uint4, of), ofv);
#pragma unroll
      for (int e = 0; e < 8; ++e) dpart += df[e] * ofv[e];
    }
    const float my_ndelta = -(dpart + __shfl_xor(dpart, 32, 64)) * inv_s;  // lanes l and l^32 hold the two halves of row l & 31
    const uint32_t rowpair = DROP ? prob_row_pair(b * heads + h, qb * 32 + (lane & 31), dropL) + 2 * half : 0u;
    const float my_lse = qb * 32 + (lane & 31) < Le ? lse[lse0 + qb * 32 + (lane & 31)] * kLog2e : kLseNoRow;
    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    for (int kb = 0; kb < nblk; ++kb) {
      f32x16 sacc, dpacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = my_ndelta; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Kt, kb * 32, s, lane), qf[s], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Vt, kb * 32, s, lane), dof[s], dpacc, 0, 0, 0);
      }
      float ds[16];
      [[maybe_unused]] float pa[DROP ? 16 : 1];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float4 ma = *reinterpret_cast<const float4*>(madd + kb * 32 + 8 * rg + 4 * half);
        const float mm[4] = {ma.x, ma.y, ma.z, ma.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = rg * 4 + e;
          const float p = fast_exp2(sacc[r] * sl2 + mm[e] - my_lse);
          ds[r] = p * dpacc[r];  // dpacc was seeded with -delta
          if constexpr (DROP) pa[r] = p;
        }
      }
      if constexpr (DROP)  // dropped elements: dP = 0, i.e. dS / s = P * seed
        for_keep_qlane(rowpair + kb * 16, dm, [&](int r, bool keep) { ds[r] = keep ? ds[r] : pa[r] * my_ndelta; });
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 dsf = pack_acc(ds, j);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_tr(Kt, kb * 32 + j * 16, dt, lane), dsf, dq[dt], 0, 0, 0);
      }
    }
    if constexpr (QKSUM && !QK_ABL_NOSUM) qacc += acc_colsum32(dq, kScale * out_s, lane);
    store_acc_T16(dqkv + (row0 + qb * 32) * ld + h * 64, ld, dq, kScale * out_s, lane, Le - qb * 32);
  }
  if constexpr (QKSUM) qk_bias_store(qk_partial, qacc, b, h, H, tid);
}

template <bool DROP>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ mask,
                                                              const uint16_t* __restrict__ ctx, const uint16_t* __restrict__ dctx,
                                                              const float* __restrict__ lse, uint16_t* __restrict__ dqkv, int Lmax, int H,
                                                              const cocodr_dropout_mask dm, const AttnPacked pk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ATTN_EXTENT(Lmax, pk)
  const float inv_s = DROP ? 1.0f / dm.scale : 1.0f, out_s = DROP ? dm.scale : 1.0f;
  char* Qt = smem;
  char* Dt = smem + L * 128;
  float* lse2 = reinterpret_cast<float*>(smem + 2 * L * 128);
  float* delta = lse2 + L;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, half = lane >> 5;
  const int ld = 3 * H;
  const uint16_t* base = qkv + row0 * ld + h * 64;
  const uint16_t* obase = ctx + row0 * H + h * 64;
  const uint16_t* dobase = dctx + row0 * H + h * 64;
  stage_tile_dma(Qt, base, ld, L, lane, wid, 4, Le);
  for (int q0 = 0; q0 < L * 8; q0 += 256 * 4) {  // dO through registers (delta = rowsum(dO * O)), four 16-B chunks per thread a round
    uint4 dreg[4], oreg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = q0 + tid + i * 256, row = q >> 3, ch = q & 7;
      dreg[i] = oreg[i] = make_uint4(0u, 0u, 0u, 0u);
      if (q < L * 8 && row < Le) {  // L * 8 is a multiple of 256: whole waves are in or out of the first test
        dreg[i] = *reinterpret_cast<const uint4*>(dobase + (size_t)row * H + ch * 8);
        oreg[i] = *reinterpret_cast<const uint4*>(obase + (size_t)row * H + ch * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = q0 + tid + i * 256, row = q >> 3, ch = q & 7;
      if (q >= L * 8) continue;
      *reinterpret_cast<uint4*>(Dt + tile64_off(row, ch)) = dreg[i];
      float df[8], of[8];
      unpack8(dreg[i], df);
      unpack8(oreg[i], of);
      float part = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) part += df[e] * of[e];
      part += dpp_partner<0xB1>(part);   // lane ^ 1
      part += dpp_partner<0x4E>(part);   // lane ^ 2
      part += dpp_partner<0x141>(part);  // the other quad of the 8 lanes that hold a row (DPP operands, no LDS round trip)
      if (ch == 0) delta[row] = DROP ? -part * inv_s : -part;  // negated: it seeds the dP accumulators below
    }
  }
  for (int i = tid; i < L; i += 256) lse2[i] = i < Le ? lse[lse0 + i] * kLog2e : kLseNoRow;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const float sl2 = kScale * kLog2e;
  const int nblk = L / 32;
  for (int kb = wid; kb < nblk; kb += 4) {
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = frag_rows_global(base + H, ld, kb * 32, s, lane, Le);
      vf[s] = frag_rows_global(base + 2 * H, ld, kb * 32, s, lane, Le);
    }
    const float my_madd = (kb * 32 + (lane & 31) < Le && mask[row0 + kb * 32 + (lane & 31)] != 0) ? 0.f : kMaskNeg;
    const uint32_t Lh = (uint32_t)dropL >> 1, kshift = (lane & 1) << 4;
    const uint32_t keypair = DROP ? prob_row_pair(b * heads + h, 4 * half, dropL) + (uint32_t)((kb * 32 + (lane & 31)) >> 1) : 0u;
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    for (int qb = 0; qb < nblk; ++qb) {
      f32x16 sacc, dpacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {  // dP accumulators start at -delta of their query rows
        const float4 d4 = *reinterpret_cast<const float4*>(delta + qb * 32 + 8 * rg + 4 * half);
        dpacc[rg * 4 + 0] = d4.x; dpacc[rg * 4 + 1] = d4.y; dpacc[rg * 4 + 2] = d4.z; dpacc[rg * 4 + 3] = d4.w;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Qt, qb * 32, s, lane), kf[s], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Dt, qb * 32, s, lane), vf[s], dpacc, 0, 0, 0);
      }
      float p[16], ds[16];
      [[maybe_unused]] float sd[DROP ? 16 : 1];  // dS / s of a dropped element: P * seed
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float4 l4 = *reinterpret_cast<const float4*>(lse2 + qb * 32 + 8 * rg + 4 * half);
        const float ll[4] = {l4.x, l4.y, l4.z, l4.w};
        [[maybe_unused]] float dseed[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (DROP) {
          const float4 d4 = *reinterpret_cast<const float4*>(delta + qb * 32 + 8 * rg + 4 * half);
          dseed[0] = d4.x; dseed[1] = d4.y; dseed[2] = d4.z; dseed[3] = d4.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = rg * 4 + e;
          p[r] = fast_exp2(sacc[r] * sl2 + my_madd - ll[e]);
          ds[r] = p[r] * dpacc[r];  // dpacc was seeded with -delta
          if constexpr (DROP) sd[r] = p[r] * dseed[e];
        }
      }
      if constexpr (DROP)
        for_keep_klane(keypair + (uint32_t)(qb * 32) * Lh, Lh, kshift, dm, [&](int r, bool keep) {
          ds[r] = keep ? ds[r] : sd[r];
          p[r] = keep ? p[r] : 0.f;
        });
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 pf = pack_acc(p, j), dsf = pack_acc(ds, j);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_tr(Dt, qb * 32 + j * 16, dt, lane), pf, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_tr(Qt, qb * 32 + j * 16, dt, lane), dsf, dk[dt], 0, 0, 0);
        }
      }
    }
    uint16_t* out0 = dqkv + (row0 + kb * 32) * ld + h * 64;
    store_acc_T16(out0 + H, ld, dk, kScale * out_s, lane, Le - kb * 32);
    store_acc_T16(out0 + 2 * H, ld, dv, out_s, lane, Le - kb * 32);
  }
}

}  // namespace

#if defined(COCODR_ABL_TIMELINE)
extern "C" int cocodr_debug_attn_timeline(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_tl), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

namespace {
const cocodr_dropout_mask kNoDrop = {0, 0, 0, 1.0f};
template <class K>
void lds_attr(K kern) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
}  // namespace

namespace {
int attn_fwd_any(const uint16_t* qkv, const int32_t* mask, uint16_t* ctx, float* lse, int B, int L, int heads, const cocodr_dropout_mask* drop,
                 const AttnPacked& pk, cocodr_stream_t stream) {
  CK_ARG(qkv && mask && ctx && lse, "attn_fwd: null pointer");
  CK_ARG(B > 0 && heads > 0, "attn_fwd: bad shape");
  CK_ARG(L % 32 == 0 && L >= 32 && L <= 512, "attn_fwd: L=%d must be a multiple of 32 in [32,512]", L);
  const bool dropping = drop != nullptr && drop->threshold != 0;
  const int dl = pk.seq_off ? pk.drop_L : L;
  CK_ARG(!dropping || ((unsigned long long)B * heads * dl * dl <= (1ull << 33) && drop->threshold < 65536),
         "attn_fwd: dropout indexes at most 2^33 probabilities");
  const int H = heads * 64;
  const size_t lds = (size_t)2 * L * 128 + (size_t)L * 4;
  static cocodr_lds_once attr_done;
  if (attr_done.pending()) {
    lds_attr(attn_fwd_kernel<false>);
    lds_attr(attn_fwd_kernel<true>);
    attr_done.done();
  }
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(PROF_ATTN, st, 4.0 * B * heads * (double)L * L * 64);
  hipLaunchKernelGGL(dropping ? attn_fwd_kernel<true> : attn_fwd_kernel<false>, dim3(heads, B, (L + 127) / 128), dim3(256), lds, st, qkv,
                     mask, ctx, lse, L, H, dropping ? *drop : kNoDrop, pk);
  CK_LAUNCH("attn_fwd");
  return COCODR_OK;
}
int check_packed(const int32_t* seq_off, int B, int T, int max_len, int drop_L) {
  CK_ARG(seq_off != nullptr && B > 0 && T >= B && T % 32 == 0, "attn(packed): need seq_off, T %% 32 == 0 and T >= B (T=%d, B=%d)", T, B);
  CK_ARG(max_len % 32 == 0 && max_len >= 32 && max_len <= 512 && drop_L >= max_len, "attn(packed): max_len=%d must be a multiple of 32 in [32,512] and <= drop_L=%d", max_len, drop_L);
  return COCODR_OK;
}
}  // namespace

extern "C" int cocodr_attn_fwd_drop(const uint16_t* qkv, const int32_t* mask, uint16_t* ctx, float* lse, int B, int L, int heads,
                                    const cocodr_dropout_mask* drop, cocodr_stream_t stream) {
  return attn_fwd_any(qkv, mask, ctx, lse, B, L, heads, drop, AttnPacked{nullptr, 0, 0, nullptr}, stream);
}
extern "C" int cocodr_attn_fwd_packed(const uint16_t* qkv, const int32_t* mask, uint16_t* ctx, float* lse, const int32_t* seq_off,
                                      const int32_t* seq_order, int B, int T, int max_len, int heads, const cocodr_dropout_mask* drop,
                                      int drop_L, cocodr_stream_t stream) {
  if (int rc = check_packed(seq_off, B, T, max_len, drop_L)) return rc;
  return attn_fwd_any(qkv, mask, ctx, lse, B, max_len, heads, drop, AttnPacked{seq_off, T, drop_L, seq_order}, stream);
}
extern "C" int cocodr_attn_fwd(const uint16_t* qkv, const int32_t* mask, uint16_t* ctx, float* lse, int B, int L, int heads,
                               cocodr_stream_t stream) {
  return cocodr_attn_fwd_drop(qkv, mask, ctx, lse, B, L, heads, nullptr, stream);
}

namespace {
int attn_bwd_any(const uint16_t* qkv, const int32_t* mask, const uint16_t* ctx, const uint16_t* dctx, const float* lse, uint16_t* dqkv,
                 float* qk_bias_partial, int B, int L, int heads, const cocodr_dropout_mask* drop, const AttnPacked& pk,
                 cocodr_stream_t stream) {
  CK_ARG(qkv && mask && ctx && dctx && lse && dqkv, "attn_bwd: null pointer");
  CK_ARG(B > 0 && heads > 0, "attn_bwd: bad shape");
  CK_ARG(L % 32 == 0 && L >= 32 && L <= 512, "attn_bwd: L=%d must be a multiple of 32 in [32,512]", L);
  const bool dropping = drop != nullptr && drop->threshold != 0;
  const int dl = pk.seq_off ? pk.drop_L : L;
  CK_ARG(!dropping || ((unsigned long long)B * heads * dl * dl <= (1ull << 33) && drop->threshold < 65536),
         "attn_bwd: dropout indexes at most 2^33 probabilities");
  const cocodr_dropout_mask dm = dropping ? *drop : kNoDrop;
  const int H = heads * 64;
  const size_t lds = (size_t)4 * L * 128 + (size_t)3 * L * 4;
  static cocodr_lds_once attr_done;
  if (attr_done.pending()) {
    lds_attr(attn_bwd_kernel<false, false>); lds_attr(attn_bwd_kernel<true, false>);
    lds_attr(attn_bwd_kernel<false, true>); lds_attr(attn_bwd_kernel<true, true>);
    lds_attr(attn_bwd1_kernel<false, false>); lds_attr(attn_bwd1_kernel<true, false>);
    lds_attr(attn_bwd1_kernel<false, true>); lds_attr(attn_bwd1_kernel<true, true>);
    lds_attr(attn_bwd_dq_kernel<false, false>); lds_attr(attn_bwd_dq_kernel<true, false>);
    lds_attr(attn_bwd_dq_kernel<false, true>); lds_attr(attn_bwd_dq_kernel<true, true>);
    lds_attr(attn_bwd_dkv_kernel<false>); lds_attr(attn_bwd_dkv_kernel<true>);
    attr_done.done();
  }
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(PROF_ATTN, st, 10.0 * B * heads * (double)L * L * 64);
  const bool qks = qk_bias_partial != nullptr;
  if (L > 256) {  // two kernels, each with the pair of [L,64] tiles it sweeps resident
    const size_t lds_q = (size_t)2 * L * 128 + (size_t)L * 4, lds_kv = (size_t)2 * L * 128 + (size_t)2 * L * 4;
    auto kq = dropping ? (qks ? attn_bwd_dq_kernel<true, true> : attn_bwd_dq_kernel<false, true>)
                       : (qks ? attn_bwd_dq_kernel<true, false> : attn_bwd_dq_kernel<false, false>);
    hipLaunchKernelGGL(kq, dim3(heads, B), dim3(256), lds_q, st, qkv, mask, ctx, dctx, lse, dqkv, L, H, qk_bias_partial, dm, pk);
    CK_LAUNCH("attn_bwd(dq)");
    hipLaunchKernelGGL(dropping ? attn_bwd_dkv_kernel<true> : attn_bwd_dkv_kernel<false>, dim3(heads, B), dim3(256), lds_kv, st, qkv, mask,
                       ctx, dctx, lse, dqkv, L, H, dm, pk);
    CK_LAUNCH("attn_bwd(dkv)");
    return COCODR_OK;
  }
  static int stagger = -1;  // x 4096 clocks; COCODR_ATTN_STAGGER overrides (0 disables)
  if (stagger < 0) {
    const char* e = getenv("COCODR_ATTN_STAGGER");
    stagger = e ? atoi(e) : 2;
  }
  static const bool two_phase = getenv("COCODR_ATTN_TWO_PHASE") != nullptr;  // A/B switch: the round-1..3 kernel at L <= 128 as well
  if (L <= 128 && !two_phase) {  // one pass: every wave one key block, dS staged for the dQ product (see attn_bwd1_kernel)
    const size_t lds1 = (size_t)4 * L * 128 + (size_t)2 * L * 4;
    auto k1 = dropping ? (qks ? attn_bwd1_kernel<true, true> : attn_bwd1_kernel<false, true>)
                       : (qks ? attn_bwd1_kernel<true, false> : attn_bwd1_kernel<false, false>);
    hipLaunchKernelGGL(k1, dim3(heads, B), dim3(256), lds1, st, qkv, mask, ctx, dctx, lse, dqkv, L, H,
                       2 * lds1 <= 160 * 1024 && heads * B > 512 && !pk.seq_off ? stagger : 0, qk_bias_partial, dm, pk);
    CK_LAUNCH("attn_bwd(one pass)");
    return COCODR_OK;
  }
  auto kf = dropping ? (qks ? attn_bwd_kernel<true, true> : attn_bwd_kernel<false, true>)
                     : (qks ? attn_bwd_kernel<true, false> : attn_bwd_kernel<false, false>);
  hipLaunchKernelGGL(kf, dim3(heads, B), dim3(256), lds, st, qkv, mask, ctx, dctx, lse, dqkv, L, H,
                     2 * lds <= 160 * 1024 && heads * B > 512 && !pk.seq_off ? stagger : 0, qk_bias_partial, dm, pk);
  CK_LAUNCH("attn_bwd");
  return COCODR_OK;
}
}  // namespace

extern "C" int cocodr_attn_bwd_drop(const uint16_t* qkv, const int32_t* mask, const uint16_t* ctx, const uint16_t* dctx,
                                    const float* lse, uint16_t* dqkv, float* qk_bias_partial, int B, int L, int heads,
                                    const cocodr_dropout_mask* drop, cocodr_stream_t stream) {
  return attn_bwd_any(qkv, mask, ctx, dctx, lse, dqkv, qk_bias_partial, B, L, heads, drop, AttnPacked{nullptr, 0, 0, nullptr}, stream);
}
extern "C" int cocodr_attn_bwd_packed(const uint16_t* qkv, const int32_t* mask, const uint16_t* ctx, const uint16_t* dctx,
                                      const float* lse, uint16_t* dqkv, float* qk_bias_partial, const int32_t* seq_off,
                                      const int32_t* seq_order, int B, int T, int max_len, int heads, const cocodr_dropout_mask* drop,
                                      int drop_L, cocodr_stream_t stream) {
  if (int rc = check_packed(seq_off, B, T, max_len, drop_L)) return rc;
  return attn_bwd_any(qkv, mask, ctx, dctx, lse, dqkv, qk_bias_partial, B, max_len, heads, drop, AttnPacked{seq_off, T, drop_L, seq_order}, stream);
}
extern "C" int cocodr_attn_bwd(const uint16_t* qkv, const int32_t* mask, const uint16_t* ctx, const uint16_t* dctx,
                               const float* lse, uint16_t* dqkv, float* qk_bias_partial, int B, int L, int heads,
                               cocodr_stream_t stream) {
  return cocodr_attn_bwd_drop(qkv, mask, ctx, dctx, lse, dqkv, qk_bias_partial, B, L, heads, nullptr, stream);
}
