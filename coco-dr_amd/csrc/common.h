// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the COCO-DR hot path.
// wave = 64 lanes everywhere; no CUDA-compat paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/cocodr.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LDS_PTR(T) __attribute__((address_space(3))) T*

// ------------------------------------------------------------------ error plumbing (host)
void cocodr_set_error(const char* fmt, ...);
#define CK_ARG(cond, ...)                 \
  do {                                    \
    if (!(cond)) {                        \
      cocodr_set_error(__VA_ARGS__);      \
      return COCODR_ERR_INVALID;          \
    }                                     \
  } while (0)
#define CK_LAUNCH(name)                                                        \
  do {                                                                         \
    hipError_t e_ = hipGetLastError();                                         \
    if (e_ != hipSuccess) {                                                    \
      cocodr_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
      return COCODR_ERR_LAUNCH;                                                \
    }                                                                          \
  } while (0)

// ------------------------------------------------------------------ bf16 <-> f32
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round to nearest even: gfx950 converts a pair per instruction (v_cvt_pk_bf16_f32); the integer form
// (add 0x7fff + lsb, shift) costs ~7 VALU operations per pair, which the VALU-bound epilogues and the attention inner
// loops feel
typedef __bf16 cocodr_bf16x2 __attribute__((ext_vector_type(2)));
typedef float cocodr_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const cocodr_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, cocodr_bf16x2));
}
__device__ __forceinline__ uint16_t f2bf(float f) { return (uint16_t)(pack2bf(f, 0.f) & 0xffffu); }
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  return make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
}
__device__ __forceinline__ void unpack4(const uint2& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
}
__device__ __forceinline__ uint2 pack4(const float* f) { return make_uint2(pack2bf(f[0], f[1]), pack2bf(f[2], f[3])); }

// ------------------------------------------------------------------ wave64 reductions
// Within a 16-lane row the partner values arrive as DPP operands of the VALU instruction (lane ^ 1, lane ^ 2, the mirrored
// lane of the 8-lane half, the mirrored lane of the row); the four row totals are then read as scalars.  No LDS round trips:
// the six dependent ds_bpermute of a __shfl_xor ladder cost ~0.4 us per reduction, which the one-row-per-wave LayerNorm
// kernels (two reductions each) are made of.  Every lane returns the same value; the order of the additions is fixed.
template <int CTRL>
__device__ __forceinline__ float dpp_partner(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_value(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_partner<0xB1>(v);   // quad_perm 1,0,3,2
  v += dpp_partner<0x4E>(v);   // quad_perm 2,3,0,1
  v += dpp_partner<0x141>(v);  // row_half_mirror
  v += dpp_partner<0x140>(v);  // row_mirror
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_partner<0xB1>(v));
  v = fmaxf(v, dpp_partner<0x4E>(v));
  v = fmaxf(v, dpp_partner<0x141>(v));
  v = fmaxf(v, dpp_partner<0x140>(v));
  return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

// ------------------------------------------------------------------ math
// exact-erf GELU (hidden_act="gelu").  erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the
// bf16 output rounding of 2^-9 relative): one v_exp + one v_rcp + 5 FMAs; exp(-x^2/2) is shared with the
// Gaussian pdf term of the derivative.  The epilogue that evaluates this is VALU-bound (about as many VALU cycles per
// tile as the main loop has MFMA cycles), so every operation counts.
// q = (1 - erf(|x| / sqrt 2)) / 2 = the Gaussian tail probability, e = exp(-x^2 / 2); the constants of 7.1.26 carry the
// 1/sqrt 2 of the argument and the 1/2 of the cdf, so the cdf is 1 - q (x >= 0) or q (x < 0) without further scaling
__device__ __forceinline__ void gauss_tail_terms(float x, float& q, float& e) {
#pragma clang fp contract(off)  // only the explicit fmaf below fuse: the value must not depend on what else the caller computes
  // v_rcp_f32 (1 ulp): __frcp_rn / 1.0f / x expand to the ten-instruction IEEE division sequence, per element, in a
  // VALU-bound epilogue; the approximation itself is good to 1.5e-7
  const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, fabsf(x), 1.0f));   // 0.3275911 / sqrt 2
  e = __builtin_amdgcn_exp2f(x * x * -0.72134752f);                // exp(-x^2 / 2) = 2^(-x^2 log2(e) / 2)
  float p = fmaf(0.5307027145f, t, -0.7265760135f);                // the 7.1.26 coefficients, halved
  p = fmaf(p, t, 0.7107068705f);
  p = fmaf(p, t, -0.142248368f);
  p = fmaf(p, t, 0.127414796f);
  q = p * t * e;
}
__device__ __forceinline__ float gauss_cdf(float x, float q) {
#pragma clang fp contract(off)
  return x >= 0.f ? 1.0f - q : q;
}
// (no implicit contraction: left to the compiler, x * (1 - q) becomes an fma in this function but not in gelu_erf_both, where
// the cdf has a second use - the inference and the training forward then differ by a bf16 ulp here and there)
__device__ __forceinline__ float gelu_erf(float x) {
#pragma clang fp contract(off)
  float q, e;
  gauss_tail_terms(x, q, e);
  return x * gauss_cdf(x, q);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  float q, e;
  gauss_tail_terms(x, q, e);
  return fmaf(x * 0.3989422804014327f, e, gauss_cdf(x, q));
}
// value and derivative from one erf / exp evaluation (the forward saves the derivative for the backward)
__device__ __forceinline__ void gelu_erf_both(float x, float& g, float& gp) {
#pragma clang fp contract(off)
  float q, e;
  gauss_tail_terms(x, q, e);
  const float cdf = gauss_cdf(x, q);
  g = x * cdf;
  gp = fmaf(x * 0.3989422804014327f, e, cdf);
}

// The same two functions for the 8 elements of an epilogue chunk, written stage by stage ACROSS the elements: element for element
// the operations and their order are those of gelu_erf / gelu_erf_both (bit-identical results), but the eight dependency chains
// (rcp, exp2, four fused multiply-adds, ...) sit side by side in the instruction stream instead of one after the other - with one
// wave per SIMD (gemm_a4.hip) nothing else fills the issue slots a dependent chain leaves empty.  GELU8_STAGE makes a stage's
// eight results opaque at once: hipcc must finish the stage for all elements before the next one starts (left alone - and even
// with sched_barrier, which does not order pure arithmetic - it re-serialises the chains to save registers).
// The arithmetic of a stage runs on PAIRS of elements (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two IEEE single-precision
// operations per instruction, the same roundings as the scalar forms - left to itself hipcc pairs some of the multiplications but
// keeps the polynomial as scalar v_fmaak_f32 with literal constants); v_rcp / v_exp and the |x| modifier have no packed form.
typedef float gelu_f2 __attribute__((ext_vector_type(2)));
#define GELU8_STAGE(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]))
__device__ __forceinline__ gelu_f2 gelu_splat(float c) { return gelu_f2{c, c}; }
__device__ __forceinline__ void gelu8_tail_terms(const float (&x)[8], gelu_f2 (&t)[4], gelu_f2 (&e)[4], gelu_f2 (&p)[4]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int j = 0; j < 4; ++j)
    t[j] = gelu_f2{__builtin_amdgcn_rcpf(fmaf(0.23164189f, fabsf(x[2 * j]), 1.0f)), __builtin_amdgcn_rcpf(fmaf(0.23164189f, fabsf(x[2 * j + 1]), 1.0f))};
  GELU8_STAGE(t);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const gelu_f2 x2 = {x[2 * j], x[2 * j + 1]};
    const gelu_f2 a = x2 * x2 * gelu_splat(-0.72134752f);
    e[j] = gelu_f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
  }
  GELU8_STAGE(e);
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = __builtin_elementwise_fma(gelu_splat(0.5307027145f), t[j], gelu_splat(-0.7265760135f));
  GELU8_STAGE(p);
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = __builtin_elementwise_fma(p[j], t[j], gelu_splat(0.7107068705f));
  GELU8_STAGE(p);
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = __builtin_elementwise_fma(p[j], t[j], gelu_splat(-0.142248368f));
  GELU8_STAGE(p);
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = __builtin_elementwise_fma(p[j], t[j], gelu_splat(0.127414796f));
  GELU8_STAGE(p);
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = p[j] * t[j] * e[j];   // q
  GELU8_STAGE(p);
}
__device__ __forceinline__ void gelu_erf_both8(const float (&x)[8], float (&g)[8], float (&gp)[8]) {
#pragma clang fp contract(off)
  gelu_f2 t[4], e[4], q[4];
  gelu8_tail_terms(x, t, e, q);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const gelu_f2 x2 = {x[2 * j], x[2 * j + 1]};
    const gelu_f2 up = gelu_splat(1.0f) - q[j];
    const gelu_f2 cdf = {x2.x >= 0.f ? up.x : q[j].x, x2.y >= 0.f ? up.y : q[j].y};
    const gelu_f2 g2 = x2 * cdf;
    const gelu_f2 d2 = __builtin_elementwise_fma(x2 * gelu_splat(0.3989422804014327f), e[j], cdf);
    g[2 * j] = g2.x; g[2 * j + 1] = g2.y;
    gp[2 * j] = d2.x; gp[2 * j + 1] = d2.y;
  }
}
__device__ __forceinline__ void gelu_erf8(float (&x)[8]) {   // in place; the value of gelu_erf (see the note on contraction there)
  gelu_f2 t[4], e[4], q[4];
  gelu8_tail_terms(x, t, e, q);
  {
#pragma clang fp contract(off)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const gelu_f2 x2 = {x[2 * j], x[2 * j + 1]};
      const gelu_f2 up = gelu_splat(1.0f) - q[j];
      const gelu_f2 cdf = {x2.x >= 0.f ? up.x : q[j].x, x2.y >= 0.f ? up.y : q[j].y};
      const gelu_f2 g2 = x2 * cdf;
      x[2 * j] = g2.x; x[2 * j + 1] = g2.y;
    }
  }
}
#undef GELU8_STAGE

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: a call site keeps one of these (static) and
// raises the limit once per device it launches on (a bit per device ordinal; a second thread at worst repeats the idempotent call)
#include <atomic>
struct cocodr_lds_once {
  std::atomic<unsigned long long> mask{0};
  static unsigned long long bit() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return 1ull << (dev & 63);
  }
  bool pending() const { return (mask.load(std::memory_order_acquire) & bit()) == 0; }
  void done() { mask.fetch_or(bit(), std::memory_order_release); }
};

// ------------------------------------------------------------------ dropout masks (cocodr_dropout_mask, include/cocodr.h)
// One 32-bit word per PAIR of adjacent elements of the site's tensor: w = lowbias32(pair ^ k0) ^ k1, element 2 pair takes
// the low 16 bits, element 2 pair + 1 the high 16 bits, keep iff the 16-bit value >= threshold (= round(p * 65536)).
// lowbias32 is a bijection of the 32-bit pair index, so within a site no two pairs share a word by construction.
__host__ __device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t drop_word(uint32_t pair, uint32_t k0, uint32_t k1) { return lowbias32(pair ^ k0) ^ k1; }
__host__ __device__ __forceinline__ bool drop_keep_lo(uint32_t w, uint32_t thr) { return (w & 0xffffu) >= thr; }
__host__ __device__ __forceinline__ bool drop_keep_hi(uint32_t w, uint32_t thr) { return (w >> 16) >= thr; }
// v[0..N) = consecutive elements starting at the EVEN flat index `first`: dropped ones -> 0, kept ones x scale
template <int N>
__device__ __forceinline__ void drop_apply(float (&v)[N], uint64_t first, const cocodr_dropout_mask& d) {
  static_assert(N % 2 == 0, "pairs");
  const uint32_t pair0 = (uint32_t)(first >> 1);
#pragma unroll
  for (int j = 0; j < N / 2; ++j) {
    const uint32_t w = drop_word(pair0 + j, d.k0, d.k1);
    v[2 * j] = drop_keep_lo(w, d.threshold) ? v[2 * j] * d.scale : 0.f;
    v[2 * j + 1] = drop_keep_hi(w, d.threshold) ? v[2 * j + 1] * d.scale : 0.f;
  }
}

// ------------------------------------------------------------------ LDS tile addressing
// [rows][64] bf16 tiles (128-B rows, 8 chunks of 16 B).  The chunk index is XOR-swizzled with a
// 3-bit function of the row chosen so that BOTH access patterns used on such tiles are
// bank-conflict free on gfx950:
//   * ds_read_b128 of MFMA 32x32x16 operand fragments (lane -> row, fixed chunk): the 16-lane
//     service groups see 16 distinct (row&1, f(row)) pairs;
//   * ds_read_b64_tr_b16 of 4 consecutive rows x 64 B: rows r and r+2 differ in bit 2 of f.
__host__ __device__ __forceinline__ int swz64(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
// byte offset of 16-B chunk `ch` (0..7) of row `row`
__host__ __device__ __forceinline__ int tile64_off(int row, int ch) { return row * 128 + ((ch ^ swz64(row)) << 4); }
// [rows][128] bf16 tiles (256-B rows, 16 chunks): only read through ds_read_b64_tr_b16
// (4 rows x 64 B per 32 lanes) -> rows r..r+3 are spread over the four 64-B windows.
__host__ __device__ __forceinline__ int tile128_off(int row, int ch) { return row * 256 + ((ch ^ ((row & 3) << 2)) << 4); }

__device__ __forceinline__ bf16x8 lds_read_b128(const char* lds, int off) {
  return *reinterpret_cast<const bf16x8*>(lds + off);
}
// Transposed 4x16 read: the 16 lanes of a group hand in the addresses of the sixteen 8-byte
// pieces of a [4 rows][16 cols] bf16 block (lane c -> row c/4, cols 4*(c%4)..+3); lane c gets
// back column c of the block, rows 0..3.
__device__ __forceinline__ s16x4 lds_read_tr16(const char* lds, int off) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + off));
}
__device__ __forceinline__ bf16x8 join_tr(s16x4 a, s16x4 b) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

// XCD-aware, bijective block remap (8 XCDs, block b is dispatched to XCD b % 8): every XCD gets a
// contiguous range of logical tiles so neighbouring tiles (sharing operand panels) share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// internal (csrc/rowops.hip), used by the GEMM's fused column sums and the encoder backward's deferred reductions:
// out_s[z * stride_out + n] = sum_p partial[((z * P + p) * nseg + s) * n_len + n],  s < nseg <= 3
struct cocodr_reduce_job {  // one cocodr_reduce_partials call
  const float* partial;
  float *o0, *o1, *o2;
  int P, nseg, n_len, batch;
  long long stride_out;
};
int cocodr_reduce_partials_multi(const cocodr_reduce_job* jobs, int njobs, hipStream_t st);
int cocodr_reduce_partials(const float* partial, float* o0, float* o1, float* o2, int P, int nseg, int n_len, int batch,
                           long long stride_out, hipStream_t st);
// LayerNorm backward without the final reduction: partial [ln_bwd_blocks(M)][nseg][H] (dgamma, dbeta[, dy column sums])
int cocodr_ln_bwd_blocks(int M);
// dy_drop / dm (optional): the dropout form of cocodr_ln_bwd_drop
int cocodr_ln_bwd_partials(const uint16_t* dout, const uint16_t* y, const float* gamma, const float* mean, const float* rstd,
                           uint16_t* dy, float* partial, int M, int H, int nseg, hipStream_t st, uint16_t* dy_drop = nullptr,
                           const cocodr_dropout_mask* dm = nullptr);
