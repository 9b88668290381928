// HBM-bound row kernels of the encoder: embedding gather + LayerNorm, LayerNorm forward/backward,
// bias-gradient column sums, fp32->bf16 weight shadow cast.
// One 64-lane wave owns one token row; a lane owns the 4-element (8 B bf16 / 16 B fp32) chunks
// lane, lane+64, ... so every wave-level access is a contiguous 512 B / 1 KiB burst; row statistics
// are wave shuffle reductions (no LDS, no atomics on the activation path).
#include <algorithm>

#include "common.h"

namespace {

constexpr int MAXC = 4;  // chunks per lane -> H <= 1024

template <int NC>
struct RowVecT {
  float v[NC][4];
};
using RowVec = RowVecT<MAXC>;

template <int NC>
__device__ __forceinline__ void load_bf16_row(const uint16_t* row, int nch, int lane, RowVecT<NC>& r) {
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) unpack4(*reinterpret_cast<const uint2*>(row + c * 4), r.v[i]);
    else r.v[i][0] = r.v[i][1] = r.v[i][2] = r.v[i][3] = 0.f;
  }
}
__device__ __forceinline__ void load_f32_row(const float* row, int nch, int lane, RowVec& r) {
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const float4 t = *reinterpret_cast<const float4*>(row + c * 4);
      r.v[i][0] = t.x; r.v[i][1] = t.y; r.v[i][2] = t.z; r.v[i][3] = t.w;
    } else r.v[i][0] = r.v[i][1] = r.v[i][2] = r.v[i][3] = 0.f;
  }
}
template <int NC>
__device__ __forceinline__ void store_bf16_row(uint16_t* row, int nch, int lane, const RowVecT<NC>& r) {
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) *reinterpret_cast<uint2*>(row + c * 4) = pack4(r.v[i]);
  }
}
template <int NC>
__device__ __forceinline__ float row_sum(const RowVecT<NC>& r) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) s += (r.v[i][0] + r.v[i][1]) + (r.v[i][2] + r.v[i][3]);
  return wave_sum(s);
}

// mean / rstd of a row held in registers (two-pass, biased variance; padded chunks hold zeros)
template <int NC>
__device__ __forceinline__ void row_stats(const RowVecT<NC>& r, int nch, int lane, int H, float eps, float& mean, float& rstd) {
  mean = row_sum(r) / (float)H;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i)
    if (lane + 64 * i < nch) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = r.v[i][e] - mean; s += d * d; }
    }
  rstd = rsqrtf(wave_sum(s) / (float)H + eps);
}

template <int NC>
__device__ __forceinline__ void ln_apply(RowVecT<NC>& r, const float* gamma, const float* beta, int nch, int lane, float mean, float rstd) {
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + c * 4);
      const float4 b = *reinterpret_cast<const float4*>(beta + c * 4);
      r.v[i][0] = (r.v[i][0] - mean) * rstd * g.x + b.x;
      r.v[i][1] = (r.v[i][1] - mean) * rstd * g.y + b.y;
      r.v[i][2] = (r.v[i][2] - mean) * rstd * g.z + b.z;
      r.v[i][3] = (r.v[i][3] - mean) * rstd * g.w + b.w;
    }
  }
}

// dropout mask of one token row held in registers (flat element index row * H + 4 c + e): dropped -> 0, kept -> x scale
template <int NC>
__device__ __forceinline__ void drop_row(RowVecT<NC>& r, int row, int H, int nch, int lane, const cocodr_dropout_mask& dm) {
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) drop_apply<4>(r.v[i], (uint64_t)row * H + c * 4, dm);
  }
}

// gather word[id] + pos[l] + type0 into registers
__device__ __forceinline__ void embed_gather(const float* word, const float* pos, const float* type0, int id, int l, int H,
                                             int nch, int lane, RowVec& r) {
  const float* w = word + (size_t)id * H;
  const float* p = pos + (size_t)l * H;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const float4 a = *reinterpret_cast<const float4*>(w + c * 4);
      const float4 b = *reinterpret_cast<const float4*>(p + c * 4);
      const float4 t = *reinterpret_cast<const float4*>(type0 + c * 4);
      r.v[i][0] = a.x + b.x + t.x; r.v[i][1] = a.y + b.y + t.y; r.v[i][2] = a.z + b.z + t.z; r.v[i][3] = a.w + b.w + t.w;
    } else r.v[i][0] = r.v[i][1] = r.v[i][2] = r.v[i][3] = 0.f;
  }
}

__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(const int32_t* __restrict__ ids, const float* __restrict__ word,
                                                           const float* __restrict__ pos, const float* __restrict__ type0,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           uint16_t* __restrict__ out, float* __restrict__ mean_o,
                                                           float* __restrict__ rstd_o, int M, int L, int H, int vocab, float eps,
                                                           const cocodr_dropout_mask dm, const int32_t* __restrict__ positions) {
  const int lane = threadIdx.x & 63, nch = H >> 2;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    RowVec r;
    embed_gather(word, pos, type0, id, positions ? positions[row] : row % L, H, nch, lane, r);  // packed batches carry their position ids
    float mean, rstd;
    row_stats(r, nch, lane, H, eps, mean, rstd);
    ln_apply(r, gamma, beta, nch, lane, mean, rstd);
    if (dm.threshold) drop_row(r, row, H, nch, lane, dm);  // hf BertEmbeddings: dropout(LayerNorm(..))
    store_bf16_row(out + (size_t)row * H, nch, lane, r);
    if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
  }
}

// NC chunks per lane (3 for H <= 768); FULL: H == 256 NC, the per-chunk guards fold away
template <int NC, bool FULL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const uint16_t* __restrict__ y, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, uint16_t* __restrict__ out,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                     float* __restrict__ cls_out, int cls_stride, int M, int H, float eps,
                                                     const int32_t* __restrict__ cls_slot) {
  const int lane = threadIdx.x & 63, nch = FULL ? 64 * NC : (H >> 2);
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    RowVecT<NC> r;
    load_bf16_row(y + (size_t)row * H, nch, lane, r);
    float mean, rstd;
    row_stats(r, nch, lane, H, eps, mean, rstd);
    ln_apply(r, gamma, beta, nch, lane, mean, rstd);
    store_bf16_row(out + (size_t)row * H, nch, lane, r);
    if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
    const int slot = !cls_out ? -1 : (cls_slot ? cls_slot[row] : (row % cls_stride == 0 ? row / cls_stride : -1));
    if (slot >= 0) {  // fp32 copy of a sequence's first row (packed batches name the rows: cls_slot[row] = sequence or -1)
      float* dst = cls_out + (size_t)slot * H;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) *reinterpret_cast<float4*>(dst + c * 4) = make_float4(r.v[i][0], r.v[i][1], r.v[i][2], r.v[i][3]);
      }
    }
  }
}

// LayerNorm backward core for one row held in registers:
//   in : d = upstream gradient, x = LN input (pre-normalisation)
//   out: d <- gradient w.r.t. the LN input; dg/db accumulate the gamma/beta gradients.
template <int NC>
__device__ __forceinline__ void ln_bwd_row(RowVecT<NC>& d, const RowVecT<NC>& x, const float* gamma, int nch, int lane, int H, float mean,
                                           float rstd, RowVecT<NC>& dg, RowVecT<NC>& db) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const float4 g4 = *reinterpret_cast<const float4*>(gamma + c * 4);
      const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (x.v[i][e] - mean) * rstd;
        dg.v[i][e] += d.v[i][e] * xh;
        db.v[i][e] += d.v[i][e];
        d.v[i][e] *= g[e];  // dxhat
        s1 += d.v[i][e];
        s2 += d.v[i][e] * xh;
      }
    }
  }
  s1 = wave_sum(s1) / (float)H;
  s2 = wave_sum(s2) / (float)H;
  // xhat is recomputed instead of kept: 4 NC fewer live registers across the two wave reductions (padded chunks hold
  // d = 0, x = 0 and are never stored)
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) d.v[i][e] = rstd * (d.v[i][e] - s1 - (x.v[i][e] - mean) * rstd * s2);
}

// reduce the NW waves' per-lane accumulators through LDS and write one partial row [H]
constexpr int NW = 16;          // waves per workgroup in the backward row kernels
constexpr int RB_THREADS = NW * 64;
__device__ __forceinline__ void block_reduce_store(const RowVec& acc, float* lds /* [NW][MAXC*256] */, float* dst, int nch, int tid) {
  const int lane = tid & 63, wid = tid >> 6;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
    *reinterpret_cast<float4*>(lds + wid * (MAXC * 256) + (lane + 64 * i) * 4) = make_float4(acc.v[i][0], acc.v[i][1], acc.v[i][2], acc.v[i][3]);
  __syncthreads();
  for (int c = tid; c < nch; c += RB_THREADS) {
    float4 s = *reinterpret_cast<const float4*>(lds + c * 4);
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const float4 t = *reinterpret_cast<const float4*>(lds + w * (MAXC * 256) + c * 4);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    *reinterpret_cast<float4*>(dst + c * 4) = s;
  }
}

template <int NC>
__device__ __forceinline__ void zero_row(RowVecT<NC>& r) {
#pragma unroll
  for (int i = 0; i < NC; ++i) r.v[i][0] = r.v[i][1] = r.v[i][2] = r.v[i][3] = 0.f;
}

// all `nseg` accumulator rows of the workgroup in as few LDS passes as the 160 KiB allow (three rows of H = 768 at once,
// two of H = 1024): every wave writes its rows, one barrier, then nseg x H/4 threads each add the NW waves' float4 in wave
// order (deterministic) and store the partial row.  red: [rows in pass][NW][H] floats.
template <int NC>
__device__ __forceinline__ void lds_put_row(const RowVecT<NC>& a, float* dst, int nch, int lane) {
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) *reinterpret_cast<float4*>(dst + c * 4) = make_float4(a.v[i][0], a.v[i][1], a.v[i][2], a.v[i][3]);
  }
}
template <int NC>
__device__ __forceinline__ void block_reduce_store_rows(const RowVecT<NC>& a0, const RowVecT<NC>& a1, const RowVecT<NC>& a2, int nseg, int per_pass,
                                                        float* red, float* prow, int H, int nch, int tid) {
  const int lane = tid & 63, wid = tid >> 6;
  for (int s0 = 0; s0 < nseg; s0 += per_pass) {
    const int ns = min(per_pass, nseg - s0);
    if (s0 > 0) __syncthreads();
    // the accumulators stay in registers: no indexing by a run-time row number
    if (0 >= s0 && 0 < s0 + ns) lds_put_row(a0, red + ((size_t)(0 - s0) * NW + wid) * H, nch, lane);
    if (1 >= s0 && 1 < s0 + ns) lds_put_row(a1, red + ((size_t)(1 - s0) * NW + wid) * H, nch, lane);
    if (2 >= s0 && 2 < s0 + ns && nseg > 2) lds_put_row(a2, red + ((size_t)(2 - s0) * NW + wid) * H, nch, lane);
    __syncthreads();
    for (int t = tid; t < ns * nch; t += RB_THREADS) {
      const int sgm = t / nch, c = t - sgm * nch;
      const float* src = red + (size_t)sgm * NW * H + c * 4;
      float4 acc = *reinterpret_cast<const float4*>(src);
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)w * H);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      *reinterpret_cast<float4*>(prow + (size_t)(s0 + sgm) * H + c * 4) = acc;
    }
  }
}

template <int NC>
struct RawRow {
  uint2 v[NC];
};
template <int NC>
__device__ __forceinline__ void load_raw_row(const uint16_t* row, int nch, int lane, RawRow<NC>& r) {
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    r.v[i] = c < nch ? *reinterpret_cast<const uint2*>(row + c * 4) : make_uint2(0u, 0u);
  }
}
template <int NC>
__device__ __forceinline__ void unpack_raw_row(const RawRow<NC>& r, RowVecT<NC>& o) {
#pragma unroll
  for (int i = 0; i < NC; ++i) unpack4(r.v[i], o.v[i]);
}

// NC = 16-B fp32 chunks per lane: 3 covers H <= 768 (a quarter fewer registers than the general 4: no spills at 4 waves / SIMD)
// FULL: H == 256 NC (768 / 1024) - every lane owns all NC chunks, so the per-chunk guards fold away
// DROP: the LayerNorm input was dropout(dense) + residual - dy keeps the gradient of the sum (the residual branch's), dy_drop
// gets it masked and scaled (the dense output's) and the column sums (that Linear's bias gradient) are taken of dy_drop
template <int NC, bool PREFETCH, bool FULL, bool DROP = false>
__global__ __launch_bounds__(RB_THREADS) void ln_bwd_kernel(const uint16_t* __restrict__ dout, const uint16_t* __restrict__ y,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean_i,
                                                     const float* __restrict__ rstd_i, uint16_t* __restrict__ dy,
                                                     float* __restrict__ partial, int M, int H, int nseg, int per_pass,
                                                     uint16_t* __restrict__ dy_drop, const cocodr_dropout_mask dm) {
  extern __shared__ __attribute__((aligned(16))) float red_dyn[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nch = FULL ? 64 * NC : (H >> 2);
  const int rows_per = (M + gridDim.x - 1) / gridDim.x;
  const int r_begin = blockIdx.x * rows_per, r_end = min(M, r_begin + rows_per);
  RowVecT<NC> dg, db, dxs;  // dxs: column sums of the input gradient = bias gradient of the Linear that feeds this LayerNorm
  zero_row(dg);
  zero_row(db);
  zero_row(dxs);
  // the next row's loads are in flight while this row is reduced and stored
  int row = r_begin + wid;
  RawRow<NC> nd, nx;
  float nmean = 0.f, nrstd = 0.f;
  if (row < r_end) {
    load_raw_row(dout + (size_t)row * H, nch, lane, nd);
    load_raw_row(y + (size_t)row * H, nch, lane, nx);
    nmean = mean_i[row]; nrstd = rstd_i[row];
  }
  for (; row < r_end; row += NW) {
    RowVecT<NC> d, x;
    unpack_raw_row(nd, d);
    unpack_raw_row(nx, x);
    const float mean = nmean, rstd = nrstd;
    const int nrow = row + NW;
    if (PREFETCH && nrow < r_end) {
      load_raw_row(dout + (size_t)nrow * H, nch, lane, nd);
      load_raw_row(y + (size_t)nrow * H, nch, lane, nx);
      nmean = mean_i[nrow]; nrstd = rstd_i[nrow];
    }
    ln_bwd_row(d, x, gamma, nch, lane, H, mean, rstd, dg, db);
    store_bf16_row(dy + (size_t)row * H, nch, lane, d);
    if constexpr (DROP) {
      drop_row(d, row, H, nch, lane, dm);
      store_bf16_row(dy_drop + (size_t)row * H, nch, lane, d);
    }
    if (nseg == 3) {
#pragma unroll
      for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) dxs.v[i][e] += d.v[i][e];
    }
    if (!PREFETCH && nrow < r_end) {  // H = 1024: the extra row in registers would spill at 4 waves / SIMD
      load_raw_row(dout + (size_t)nrow * H, nch, lane, nd);
      load_raw_row(y + (size_t)nrow * H, nch, lane, nx);
      nmean = mean_i[nrow]; nrstd = rstd_i[nrow];
    }
  }
  block_reduce_store_rows(dg, db, dxs, nseg, per_pass, red_dyn, partial + (size_t)blockIdx.x * nseg * H, H, nch, tid);
}

// grid.x = L (one workgroup per position): the position-embedding gradient row is a plain sum over
// the batch (no atomics); only the sparse word-embedding rows use fp32 atomics.  Padded batches (row = b L + l) and packed
// ones (row = seq_off[b] + l where sequence b is that long) alike - the packed form used to scatter the position rows by
// atomics as well, twice the atomic traffic of the padded kernel on fewer rows.
__global__ __launch_bounds__(RB_THREADS) void embed_ln_bwd_kernel(const uint16_t* __restrict__ dout, const int32_t* __restrict__ ids,
                                                           const float* __restrict__ word, const float* __restrict__ pos,
                                                           const float* __restrict__ type0, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                           float* __restrict__ dword, float* __restrict__ dpos,
                                                           float* __restrict__ partial, int B, int L, int H, int vocab,
                                                           const cocodr_dropout_mask dm, const int32_t* __restrict__ seq_off) {
  __shared__ __attribute__((aligned(16))) float red[NW * MAXC * 256];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nch = H >> 2;
  const int l = blockIdx.x;
  RowVec dg, db, dp;
  zero_row(dg);
  zero_row(db);
  zero_row(dp);
  const int b_per = (B + gridDim.y - 1) / gridDim.y;
  const int b_end = min(B, (int)(blockIdx.y + 1) * b_per);
  for (int b = blockIdx.y * b_per + wid; b < b_end; b += NW) {
    // packed batches (seq_off != NULL): sequence b owns rows [seq_off[b], seq_off[b + 1]); position l exists in it or not
    int row = b * L + l;
    if (seq_off != nullptr) {
      const int r0 = seq_off[b];
      if (l >= seq_off[b + 1] - r0) continue;
      row = r0 + l;
    }
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    RowVec d, x;
    load_bf16_row(dout + (size_t)row * H, nch, lane, d);
    embed_gather(word, pos, type0, id, l, H, nch, lane, x);
    // padded positions carry an exactly-zero upstream gradient (nothing attends to them): skipping them removes
    // the thousands-way same-address contention on the [PAD] row, which serialises at the memory-side atomic unit
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(d.v[i][e]));
    if (wave_max(amax) == 0.f) continue;
    if (dm.threshold) drop_row(d, row, H, nch, lane, dm);  // gradient of dropout(LayerNorm(..)) w.r.t. the LayerNorm output
    ln_bwd_row(d, x, gamma, nch, lane, H, mean_i[row], rstd_i[row], dg, db);
    float* wrow = dword + (size_t)id * H;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dp.v[i][e] += d.v[i][e];
          atomicAdd(wrow + c * 4 + e, d.v[i][e]);
        }
      }
    }
  }
  // partial rows: [split][l][3][H]; dpos / dtype0 / dgamma / dbeta are finished by reduce kernels
  float* prow = partial + ((size_t)blockIdx.y * gridDim.x + l) * 3 * H;
  block_reduce_store(dg, red, prow, nch, tid);
  block_reduce_store(db, red, prow + H, nch, tid);
  block_reduce_store(dp, red, prow + 2 * H, nch, tid);
}

// dpos[l][:] = sum over batch splits of the position partial rows
__global__ __launch_bounds__(256) void embed_dpos_kernel(const float* __restrict__ partial, float* __restrict__ dpos, int L, int H, int S) {
  const int l = blockIdx.x;
  for (int h = threadIdx.x; h < H; h += 256) {
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += partial[((size_t)s * L + l) * 3 * H + 2 * H + h];
    dpos[(size_t)l * H + h] = acc;
  }
}

// out_s[z*stride_out + n] = sum_p partial[((z*P + p)*nseg + s)*n_len + n], s < nseg
struct ReduceArgs {
  const float* partial;
  float* out[3];
  int P, nseg, n_len, batch;
  long long stride_out;
};
__global__ __launch_bounds__(256) void reduce_partials_kernel(ReduceArgs a) {
  // block = 64 columns x 4 partial-row groups; 8 independent loads in flight per thread
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  const int s = blockIdx.y, z = blockIdx.z;
  float acc = 0.f;
  if (n < a.n_len) {
    const size_t step = (size_t)a.nseg * a.n_len;
    const float* p = a.partial + ((size_t)z * a.P * a.nseg + s) * a.n_len + n;
    int i = rg;
    for (; i + 28 < a.P; i += 32) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = p[(size_t)(i + 4 * u) * step];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += t[u];
    }
    for (; i < a.P; i += 4) acc += p[(size_t)i * step];
  }
  red[rg][c] = acc;
  __syncthreads();
  if (rg == 0 && n < a.n_len) a.out[s][(size_t)z * a.stride_out + n] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// several independent reductions of that kind in ONE launch (the deferred reductions at the end of a backward range are
// five ~5 us kernels otherwise): blockIdx.z walks the concatenated batch items of the jobs
constexpr int REDUCE_MAX_JOBS = 6;
struct ReduceJobs {
  ReduceArgs job[REDUCE_MAX_JOBS];
  int z_end[REDUCE_MAX_JOBS];  // exclusive prefix of the batch counts
  int njobs;
};
__global__ __launch_bounds__(256) void reduce_partials_multi_kernel(ReduceJobs J) {
  __shared__ float red[4][64];
  int j = 0;
  while (j + 1 < J.njobs && (int)blockIdx.z >= J.z_end[j]) ++j;
  const ReduceArgs& a = J.job[j];
  const int z = blockIdx.z - (j > 0 ? J.z_end[j - 1] : 0);
  const int s = blockIdx.y;
  if (s >= a.nseg || (int)blockIdx.x * 64 >= a.n_len) return;  // workgroup-uniform
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  float acc = 0.f;
  if (n < a.n_len) {
    const size_t step = (size_t)a.nseg * a.n_len;
    const float* p = a.partial + ((size_t)z * a.P * a.nseg + s) * a.n_len + n;
    int i = rg;
    for (; i + 28 < a.P; i += 32) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = p[(size_t)(i + 4 * u) * step];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += t[u];
    }
    for (; i < a.P; i += 4) acc += p[(size_t)i * step];
  }
  red[rg][c] = acc;
  __syncthreads();
  if (rg == 0 && n < a.n_len) a.out[s][(size_t)z * a.stride_out + n] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// column sums: workgroup = 256-column strip x row range; a lane owns 4 adjacent columns
__global__ __launch_bounds__(256) void colsum_kernel(const uint16_t* __restrict__ X, float* __restrict__ partial, int M, int N,
                                                     int ldx, long long strideX, int S) {
  __shared__ __attribute__((aligned(16))) float red[4 * 256];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int z = blockIdx.z, s = blockIdx.y;
  const int col = blockIdx.x * 256 + lane * 4;
  const int rows_per = (M + S - 1) / S;
  const int r_begin = s * rows_per, r_end = min(M, r_begin + rows_per);
  const uint16_t* Xz = X + (size_t)z * strideX;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (col < N) {
    int row = r_begin + wid;
    for (; row + 12 < r_end; row += 16) {  // four independent 8-B loads per lane in flight
      uint2 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const uint2*>(Xz + (size_t)(row + 4 * k) * ldx + col);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t[4];
        unpack4(v[k], t);
        acc[0] += t[0]; acc[1] += t[1]; acc[2] += t[2]; acc[3] += t[3];
      }
    }
    for (; row < r_end; row += 4) {
      float t[4];
      unpack4(*reinterpret_cast<const uint2*>(Xz + (size_t)row * ldx + col), t);
      acc[0] += t[0]; acc[1] += t[1]; acc[2] += t[2]; acc[3] += t[3];
    }
  }
  *reinterpret_cast<float4*>(red + wid * 256 + lane * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  if (wid == 0 && col < N) {
    float4 t0 = *reinterpret_cast<const float4*>(red + lane * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 t = *reinterpret_cast<const float4*>(red + w * 256 + lane * 4);
      t0.x += t.x; t0.y += t.y; t0.z += t.z; t0.w += t.w;
    }
    *reinterpret_cast<float4*>(partial + ((size_t)z * S + s) * N + col) = t0;
  }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t n8, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const float4 a = *reinterpret_cast<const float4*>(src + i * 8);
    const float4 b = *reinterpret_cast<const float4*>(src + i * 8 + 4);
    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    *reinterpret_cast<uint4*>(dst + i * 8) = pack8(f);
  }
  if (blockIdx.x == 0) {
    for (size_t i = n8 * 8 + threadIdx.x; i < n; i += 256) dst[i] = f2bf(src[i]);
  }
}

__global__ __launch_bounds__(256) void scatter_cls_kernel(const float* __restrict__ dE, uint16_t* __restrict__ d_last, int M, int L, int H) {
  const int lane = threadIdx.x & 63, nch = H >> 2;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    RowVec r;
    if (row % L == 0) load_f32_row(dE + (size_t)(row / L) * H, nch, lane, r);
    else zero_row(r);
    store_bf16_row(d_last + (size_t)row * H, nch, lane, r);
  }
}

// row ranges per column strip: a [8192, 1024] matrix has only four 256-column strips, so the row axis has to fill the 256 CUs
int colsum_splits(int M) { return M >= 4096 ? 128 : (M >= 512 ? 16 : 1); }
int ln_bwd_blocks(int M) { return M >= 256 * NW ? 256 : (M + NW - 1) / NW; }

int launch_ln_bwd(const uint16_t* dout, const uint16_t* y, const float* gamma, const float* mean, const float* rstd, uint16_t* dy,
                  float* partial, int M, int H, int nseg, hipStream_t st, uint16_t* dy_drop = nullptr,
                  const cocodr_dropout_mask* dm = nullptr) {
  static cocodr_lds_once attr_done;
  if (attr_done.pending()) {
    hipFuncSetAttribute((const void*)&ln_bwd_kernel<3, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)&ln_bwd_kernel<3, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)&ln_bwd_kernel<MAXC, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)&ln_bwd_kernel<MAXC, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)&ln_bwd_kernel<3, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)&ln_bwd_kernel<3, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)&ln_bwd_kernel<MAXC, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)&ln_bwd_kernel<MAXC, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.done();
  }
  const size_t row_bytes = (size_t)NW * H * 4;  // one accumulator row of every wave
  const int per_pass = std::max(1, std::min(nseg, (int)((160 * 1024) / row_bytes)));
  const bool drop = dm != nullptr && dm->threshold != 0;
  const cocodr_dropout_mask dmv = drop ? *dm : cocodr_dropout_mask{0, 0, 0, 1.0f};
  // The guard-free (FULL) instantiations looked good in isolation but spill at four waves per SIMD (23 VGPRs at H = 1024, 73 with
  // the dropout hash in the loop; the guarded ones 0 / 2): inside the training steps the guarded kernels are 0.1-1 % faster
  // without dropout (profiles/archive/r02x) and 1.5-2.3x per call with it.  COCODR_LN_FULL=1: A/B switch back to the guard-free ones.
  static const bool full = getenv("COCODR_LN_FULL") != nullptr;
  auto kern = H <= 768 ? ln_bwd_kernel<3, true, false> : ln_bwd_kernel<MAXC, false, false>;
  if (drop) kern = H <= 768 ? ln_bwd_kernel<3, true, false, true> : ln_bwd_kernel<MAXC, false, false, true>;
  if (full && H == 768) kern = drop ? ln_bwd_kernel<3, true, true, true> : ln_bwd_kernel<3, true, true>;
  if (full && H == 1024) kern = drop ? ln_bwd_kernel<MAXC, false, true, true> : ln_bwd_kernel<MAXC, false, true>;
  hipLaunchKernelGGL(kern, dim3(ln_bwd_blocks(M)), dim3(RB_THREADS), per_pass * row_bytes, st, dout, y, gamma, mean, rstd, dy, partial, M, H,
                     nseg, per_pass, dy_drop, dmv);
  CK_LAUNCH("ln_bwd");
  return COCODR_OK;
}

int launch_reduce(const float* partial, float* o0, float* o1, float* o2, int P, int nseg, int n_len, int batch, long long stride_out,
                  hipStream_t st) {
  ReduceArgs a;
  a.partial = partial;
  a.out[0] = o0; a.out[1] = o1; a.out[2] = o2;
  a.P = P; a.nseg = nseg; a.n_len = n_len; a.batch = batch; a.stride_out = stride_out;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((n_len + 63) / 64, nseg, batch), dim3(256), 0, st, a);
  CK_LAUNCH("reduce_partials");
  return COCODR_OK;
}

bool row_shape_ok(int H) { return H % 4 == 0 && H >= 4 && H <= MAXC * 256; }
}  // namespace

int cocodr_reduce_partials(const float* partial, float* o0, float* o1, float* o2, int P, int nseg, int n_len, int batch,
                           long long stride_out, hipStream_t st) {
  return launch_reduce(partial, o0, o1, o2, P, nseg, n_len, batch, stride_out, st);
}
// up to REDUCE_MAX_JOBS reductions (same argument meaning as cocodr_reduce_partials, one array entry each) in one launch
int cocodr_reduce_partials_multi(const cocodr_reduce_job* jobs, int njobs, hipStream_t st) {
  if (njobs <= 0) return COCODR_OK;
  CK_ARG(jobs && njobs <= REDUCE_MAX_JOBS, "reduce_partials_multi: 1..%d jobs", REDUCE_MAX_JOBS);
  ReduceJobs J;
  int gx = 1, gy = 1, gz = 0;
  for (int j = 0; j < njobs; ++j) {
    const cocodr_reduce_job& q = jobs[j];
    ReduceArgs& a = J.job[j];
    a.partial = q.partial;
    a.out[0] = q.o0; a.out[1] = q.o1; a.out[2] = q.o2;
    a.P = q.P; a.nseg = q.nseg; a.n_len = q.n_len; a.batch = q.batch; a.stride_out = q.stride_out;
    gx = std::max(gx, (q.n_len + 63) / 64);
    gy = std::max(gy, q.nseg);
    gz += q.batch;
    J.z_end[j] = gz;
  }
  J.njobs = njobs;
  hipLaunchKernelGGL(reduce_partials_multi_kernel, dim3(gx, gy, gz), dim3(256), 0, st, J);
  CK_LAUNCH("reduce_partials_multi");
  return COCODR_OK;
}
int cocodr_ln_bwd_blocks(int M) { return ln_bwd_blocks(M); }
int cocodr_ln_bwd_partials(const uint16_t* dout, const uint16_t* y, const float* gamma, const float* mean, const float* rstd,
                           uint16_t* dy, float* partial, int M, int H, int nseg, hipStream_t st, uint16_t* dy_drop,
                           const cocodr_dropout_mask* dm) {
  return launch_ln_bwd(dout, y, gamma, mean, rstd, dy, partial, M, H, nseg, st, dy_drop, dm);
}

namespace {
int row_grid(int M) { return std::min((M + 3) / 4, 2048); }

}  // namespace

namespace {
const cocodr_dropout_mask kNoDrop = {0, 0, 0, 1.0f};
cocodr_dropout_mask drop_or_none(const cocodr_dropout_mask* d) { return d && d->threshold ? *d : kNoDrop; }
}  // namespace

extern "C" int cocodr_embed_ln_fwd_drop(const int32_t* ids, const float* word, const float* pos, const float* type0, const float* gamma,
                                        const float* beta, uint16_t* out, float* mean, float* rstd, int B, int L, int H, int vocab,
                                        float eps, const cocodr_dropout_mask* drop, cocodr_stream_t stream) {
  CK_ARG(ids && word && pos && type0 && gamma && beta && out && mean && rstd, "embed_ln_fwd: null pointer");
  CK_ARG(B > 0 && L > 0 && vocab > 0 && row_shape_ok(H), "embed_ln_fwd: bad shape B=%d L=%d H=%d", B, L, H);
  const int M = B * L;
  hipLaunchKernelGGL(embed_ln_fwd_kernel, dim3(row_grid(M)), dim3(256), 0, (hipStream_t)stream, ids, word, pos, type0, gamma, beta,
                     out, mean, rstd, M, L, H, vocab, eps, drop_or_none(drop), (const int32_t*)nullptr);
  CK_LAUNCH("embed_ln_fwd");
  return COCODR_OK;
}
extern "C" int cocodr_embed_ln_fwd_packed(const int32_t* ids, const int32_t* positions, const float* word, const float* pos,
                                          const float* type0, const float* gamma, const float* beta, uint16_t* out, float* mean,
                                          float* rstd, int T, int H, int vocab, float eps, const cocodr_dropout_mask* drop,
                                          cocodr_stream_t stream) {
  CK_ARG(ids && positions && word && pos && type0 && gamma && beta && out && mean && rstd, "embed_ln_fwd(packed): null pointer");
  CK_ARG(T > 0 && vocab > 0 && row_shape_ok(H), "embed_ln_fwd(packed): bad shape T=%d H=%d", T, H);
  hipLaunchKernelGGL(embed_ln_fwd_kernel, dim3(row_grid(T)), dim3(256), 0, (hipStream_t)stream, ids, word, pos, type0, gamma, beta,
                     out, mean, rstd, T, 1, H, vocab, eps, drop_or_none(drop), positions);
  CK_LAUNCH("embed_ln_fwd(packed)");
  return COCODR_OK;
}
extern "C" int cocodr_embed_ln_fwd(const int32_t* ids, const float* word, const float* pos, const float* type0, const float* gamma,
                                   const float* beta, uint16_t* out, float* mean, float* rstd, int B, int L, int H, int vocab,
                                   float eps, cocodr_stream_t stream) {
  return cocodr_embed_ln_fwd_drop(ids, word, pos, type0, gamma, beta, out, mean, rstd, B, L, H, vocab, eps, nullptr, stream);
}

int embed_bwd_splits(int B) { return B >= 64 ? 8 : (B >= 16 ? 2 : 1); }
extern "C" size_t cocodr_embed_bwd_partial_floats(int L, int H) { return (size_t)8 * L * 3 * H; }

extern "C" int cocodr_embed_ln_bwd(const uint16_t* dout, const int32_t* ids, const float* word, const float* pos, const float* type0,
                                   const float* gamma, const float* mean, const float* rstd, float* dword, float* dpos,
                                   float* dtype0, float* dgamma, float* dbeta, float* partial, int B, int L, int H, int vocab,
                                   cocodr_stream_t stream) {
  return cocodr_embed_ln_bwd_drop(dout, ids, word, pos, type0, gamma, mean, rstd, dword, dpos, dtype0, dgamma, dbeta, partial, B, L, H,
                                  vocab, nullptr, stream);
}
extern "C" int cocodr_embed_ln_bwd_drop(const uint16_t* dout, const int32_t* ids, const float* word, const float* pos, const float* type0,
                                        const float* gamma, const float* mean, const float* rstd, float* dword, float* dpos,
                                        float* dtype0, float* dgamma, float* dbeta, float* partial, int B, int L, int H, int vocab,
                                        const cocodr_dropout_mask* drop, cocodr_stream_t stream) {
  CK_ARG(dout && ids && word && pos && type0 && gamma && mean && rstd && dword && dpos && dtype0 && dgamma && dbeta && partial,
         "embed_ln_bwd: null pointer");
  CK_ARG(B > 0 && L > 0 && vocab > 0 && row_shape_ok(H), "embed_ln_bwd: bad shape B=%d L=%d H=%d", B, L, H);
  hipStream_t st = (hipStream_t)stream;
  const int S = embed_bwd_splits(B);
  hipLaunchKernelGGL(embed_ln_bwd_kernel, dim3(L, S), dim3(RB_THREADS), 0, st, dout, ids, word, pos, type0, gamma, mean, rstd, dword, dpos,
                     partial, B, L, H, vocab, drop_or_none(drop), (const int32_t*)nullptr);
  CK_LAUNCH("embed_ln_bwd");
  hipLaunchKernelGGL(embed_dpos_kernel, dim3(L), dim3(256), 0, st, partial, dpos, L, H, S);
  CK_LAUNCH("embed_dpos");
  return launch_reduce(partial, dgamma, dbeta, dtype0, L * S, 3, H, 1, 0, st);
}

extern "C" int cocodr_ln_fwd(const uint16_t* y, const float* gamma, const float* beta, uint16_t* out, float* mean, float* rstd,
                             float* cls_out, int cls_stride, int M, int H, float eps, cocodr_stream_t stream) {
  return cocodr_ln_fwd_slots(y, gamma, beta, out, mean, rstd, cls_out, cls_stride, nullptr, M, H, eps, stream);
}
extern "C" int cocodr_ln_fwd_slots(const uint16_t* y, const float* gamma, const float* beta, uint16_t* out, float* mean, float* rstd,
                                   float* cls_out, int cls_stride, const int32_t* cls_slot, int M, int H, float eps,
                                   cocodr_stream_t stream) {
  CK_ARG(y && gamma && beta && out && mean && rstd, "ln_fwd: null pointer");
  CK_ARG(M > 0 && row_shape_ok(H), "ln_fwd: bad shape M=%d H=%d", M, H);
  CK_ARG(!cls_out || cls_slot || cls_stride > 0, "ln_fwd: cls_stride must be positive");
  auto kern = H == 768 ? ln_fwd_kernel<3, true> : (H < 768 ? ln_fwd_kernel<3, false> : (H == 1024 ? ln_fwd_kernel<MAXC, true> : ln_fwd_kernel<MAXC, false>));
  hipLaunchKernelGGL(kern, dim3(row_grid(M)), dim3(256), 0, (hipStream_t)stream, y, gamma, beta, out, mean, rstd, cls_out,
                     cls_stride > 0 ? cls_stride : 1, M, H, eps, cls_slot);
  CK_LAUNCH("ln_fwd");
  return COCODR_OK;
}

extern "C" size_t cocodr_embed_bwd_packed_partial_floats(int T, int H) { return (size_t)8 * std::min(512, T) * 3 * H; }
extern "C" int cocodr_embed_ln_bwd_packed(const uint16_t* dout, const int32_t* ids, const int32_t* seq_off, const float* word,
                                          const float* pos, const float* type0, const float* gamma, const float* mean,
                                          const float* rstd, float* dword, float* dpos, float* dtype0, float* dgamma, float* dbeta,
                                          float* partial, int B, int T, int max_len, int H, int vocab, const cocodr_dropout_mask* drop,
                                          cocodr_stream_t stream) {
  CK_ARG(dout && ids && seq_off && word && pos && type0 && gamma && mean && rstd && dword && dpos && dtype0 && dgamma && dbeta && partial,
         "embed_ln_bwd(packed): null pointer");
  CK_ARG(B > 0 && T > 0 && max_len > 0 && max_len <= 512 && max_len <= T && vocab > 0 && row_shape_ok(H),
         "embed_ln_bwd(packed): bad shape B=%d T=%d max_len=%d H=%d", B, T, max_len, H);
  hipStream_t st = (hipStream_t)stream;
  const int S = embed_bwd_splits(B);
  hipLaunchKernelGGL(embed_ln_bwd_kernel, dim3(max_len, S), dim3(RB_THREADS), 0, st, dout, ids, word, pos, type0, gamma, mean, rstd, dword,
                     dpos, partial, B, max_len, H, vocab, drop_or_none(drop), seq_off);
  CK_LAUNCH("embed_ln_bwd(packed)");
  hipLaunchKernelGGL(embed_dpos_kernel, dim3(max_len), dim3(256), 0, st, partial, dpos, max_len, H, S);  // rows [0, max_len) rewritten, as on the padded path
  CK_LAUNCH("embed_dpos(packed)");
  return launch_reduce(partial, dgamma, dbeta, dtype0, max_len * S, 3, H, 1, 0, st);
}

extern "C" size_t cocodr_ln_bwd_partial_floats(int M, int H) { return (size_t)ln_bwd_blocks(M) * 3 * H; }

extern "C" int cocodr_ln_bwd_drop(const uint16_t* dout, const uint16_t* y, const float* gamma, const float* mean, const float* rstd,
                                  uint16_t* dy, uint16_t* dy_drop, float* dgamma, float* dbeta, float* dy_colsum, float* partial, int M,
                                  int H, const cocodr_dropout_mask* drop, cocodr_stream_t stream) {
  CK_ARG(dout && y && gamma && mean && rstd && dy && dgamma && dbeta && partial, "ln_bwd: null pointer");
  CK_ARG(M > 0 && row_shape_ok(H), "ln_bwd: bad shape M=%d H=%d", M, H);
  const bool dropping = drop != nullptr && drop->threshold != 0;
  CK_ARG(!dropping || dy_drop, "ln_bwd: dropout needs the dy_drop output");
  hipStream_t st = (hipStream_t)stream;
  const int P = ln_bwd_blocks(M);
  const int nseg = dy_colsum ? 3 : 2;
  if (int rc = launch_ln_bwd(dout, y, gamma, mean, rstd, dy, partial, M, H, nseg, st, dy_drop, drop)) return rc;
  return launch_reduce(partial, dgamma, dbeta, dy_colsum, P, nseg, H, 1, 0, st);
}
extern "C" int cocodr_ln_bwd(const uint16_t* dout, const uint16_t* y, const float* gamma, const float* mean, const float* rstd,
                             uint16_t* dy, float* dgamma, float* dbeta, float* dy_colsum, float* partial, int M, int H,
                             cocodr_stream_t stream) {
  return cocodr_ln_bwd_drop(dout, y, gamma, mean, rstd, dy, nullptr, dgamma, dbeta, dy_colsum, partial, M, H, nullptr, stream);
}

extern "C" size_t cocodr_colsum_partial_floats(int M, int N, int batch) { return (size_t)colsum_splits(M) * N * (batch > 0 ? batch : 1); }

extern "C" int cocodr_colsum(const uint16_t* X, float* out, float* partial, int M, int N, int ldx, int batch, long long strideX,
                             long long strideOut, cocodr_stream_t stream) {
  CK_ARG(X && out && partial, "colsum: null pointer");
  CK_ARG(M > 0 && N > 0 && N % 4 == 0 && ldx % 4 == 0 && ldx >= N && batch > 0, "colsum: bad shape M=%d N=%d ldx=%d", M, N, ldx);
  hipStream_t st = (hipStream_t)stream;
  const int S = colsum_splits(M);
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256, S, batch), dim3(256), 0, st, X, partial, M, N, ldx, strideX, S);
  CK_LAUNCH("colsum");
  return launch_reduce(partial, out, nullptr, nullptr, S, 1, N, batch, strideOut, st);
}

extern "C" int cocodr_cast_f32_bf16(const float* src, uint16_t* dst, size_t n, cocodr_stream_t stream) {
  CK_ARG(src && dst, "cast: null pointer");
  CK_ARG((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "cast: pointers must be 16-byte aligned");
  if (n == 0) return COCODR_OK;
  const size_t n8 = n / 8;
  const int grid = (int)std::min((size_t)0x7fffffff, (n8 + 255) / 256 + 1);  // (full grid: see cocodr_adamw_step)
  hipLaunchKernelGGL(cast_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, n8, n);
  CK_LAUNCH("cast_f32_bf16");
  return COCODR_OK;
}

// zero fill of a gradient block (the embedding tables' gradient, 94 - 125 MB per backward: rows are accumulated with atomics,
// so the block has to start from zero; torch's fill kernel measured 1.5 TB/s on it, this one streams 16-B stores from 2048
// workgroups)
__global__ __launch_bounds__(256) void zero_kernel(float4* __restrict__ dst, size_t n4, float* __restrict__ tail, int ntail) {
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.f;
}
extern "C" int cocodr_zero_f32(float* dst, size_t n, cocodr_stream_t stream) {
  CK_ARG(dst != nullptr || n == 0, "zero_f32: null pointer");
  CK_ARG(((uintptr_t)dst & 15) == 0, "zero_f32: pointer must be 16-byte aligned");
  if (n == 0) return COCODR_OK;
  const size_t n4 = n / 4;
  const int grid = (int)std::min((size_t)0x7fffffff, n4 / 256 + 1);
  hipLaunchKernelGGL(zero_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<float4*>(dst), n4, dst + 4 * n4, (int)(n - 4 * n4));
  CK_LAUNCH("zero_f32");
  return COCODR_OK;
}

// ------------------------------------------------------------------ row plumbing of the label-sparse MLM head
// The Condenser / MLM head runs on the ~15 % labelled rows only (COCO/modeling.py:87-93, 222-224 form the full [B L, V] logits and let
// the cross entropy ignore -100): gather those rows, scatter their gradients back, and the GELU' product of the transform's backward.
// One wave per row, 8-byte lanes (H % 4 == 0); idx int64 (what torch.nonzero returns).
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint16_t* __restrict__ src, const long long* __restrict__ idx, uint16_t* __restrict__ dst,
                                                          int n, int H) {
  const int lane = threadIdx.x & 63;
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += gridDim.x * 4) {
    const uint2* s = reinterpret_cast<const uint2*>(src + (size_t)idx[r] * H);
    uint2* d = reinterpret_cast<uint2*>(dst + (size_t)r * H);
    for (int c = lane; c < H / 4; c += 64) d[c] = s[c];
  }
}
// MODE 0: dst bf16 [M,H] row idx[r] = src row r (rows not named stay as they are: the caller zero-fills);
// MODE 1: dst fp32 [M,H] row idx[r] += src row r (idx unique: no atomics)
template <int MODE>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const uint16_t* __restrict__ src, const long long* __restrict__ idx, void* __restrict__ dst,
                                                           int n, int H) {
  const int lane = threadIdx.x & 63;
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += gridDim.x * 4) {
    const uint2* s = reinterpret_cast<const uint2*>(src + (size_t)r * H);
    if (MODE == 0) {
      uint2* d = reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(dst) + (size_t)idx[r] * H);
      for (int c = lane; c < H / 4; c += 64) d[c] = s[c];
    } else {
      float4* d = reinterpret_cast<float4*>(reinterpret_cast<float*>(dst) + (size_t)idx[r] * H);
      for (int c = lane; c < H / 4; c += 64) {
        float f[4];
        unpack4(s[c], f);
        float4 v = d[c];
        v.x += f[0]; v.y += f[1]; v.z += f[2]; v.w += f[3];
        d[c] = v;
      }
    }
  }
}
__global__ __launch_bounds__(256) void mul_bf16_kernel(const uint2* __restrict__ a, const uint2* __restrict__ b, uint2* __restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float x[4], y[4];
    unpack4(a[i], x);
    unpack4(b[i], y);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] *= y[e];
    out[i] = pack4(x);
  }
}
extern "C" int cocodr_gather_rows(const uint16_t* src, const long long* idx, uint16_t* dst, int n, int H, cocodr_stream_t stream) {
  CK_ARG(src && idx && dst, "gather_rows: null pointer");
  CK_ARG(n >= 0 && H > 0 && H % 4 == 0, "gather_rows: bad shape n=%d H=%d", n, H);
  if (n == 0) return COCODR_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(std::min(2048, (n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, src, idx, dst, n, H);
  CK_LAUNCH("gather_rows");
  return COCODR_OK;
}
extern "C" int cocodr_scatter_rows(const uint16_t* src, const long long* idx, void* dst, int n, int H, int add_f32, cocodr_stream_t stream) {
  CK_ARG(src && idx && dst, "scatter_rows: null pointer");
  CK_ARG(n >= 0 && H > 0 && H % 4 == 0, "scatter_rows: bad shape n=%d H=%d", n, H);
  if (n == 0) return COCODR_OK;
  const dim3 grid(std::min(2048, (n + 3) / 4));
  if (add_f32) hipLaunchKernelGGL(scatter_rows_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, src, idx, dst, n, H);
  else hipLaunchKernelGGL(scatter_rows_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, src, idx, dst, n, H);
  CK_LAUNCH("scatter_rows");
  return COCODR_OK;
}
extern "C" int cocodr_mul_bf16(const uint16_t* a, const uint16_t* b, uint16_t* out, size_t n, cocodr_stream_t stream) {
  CK_ARG(a && b && out, "mul_bf16: null pointer");
  CK_ARG(n % 4 == 0 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 7) == 0), "mul_bf16: n %% 4 == 0 and 8-byte aligned pointers");
  if (n == 0) return COCODR_OK;
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(mul_bf16_kernel, dim3((int)std::min((size_t)0x7fffffff, n4 / 256 + 1)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const uint2*>(a), reinterpret_cast<const uint2*>(b), reinterpret_cast<uint2*>(out), n4);
  CK_LAUNCH("mul_bf16");
  return COCODR_OK;
}

namespace {
__global__ void cls_rows_kernel(const int32_t* __restrict__ seq_off, int L, int B, long long* __restrict__ idx) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) idx[b] = seq_off ? (long long)seq_off[b] : (long long)b * L;
}
}  // namespace
extern "C" int cocodr_cls_rows(const int32_t* seq_off, int L, int B, long long* idx, cocodr_stream_t stream) {
  CK_ARG(idx && B > 0 && (seq_off || L > 0), "cls_rows: bad arguments");
  hipLaunchKernelGGL(cls_rows_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, seq_off, L, B, idx);
  CK_LAUNCH("cls_rows");
  return COCODR_OK;
}

// ------------------------------------------------------------------ packed-batch layout (include/cocodr.h "Packed batches")
namespace {
template <typename T>
__device__ __forceinline__ int as_int(const void* p, size_t i) { return (int)reinterpret_cast<const T*>(p)[i]; }
__device__ __forceinline__ int load_int(const void* p, size_t i, int bytes) {
  return bytes == 8 ? as_int<long long>(p, i) : (bytes == 4 ? as_int<int32_t>(p, i) : as_int<uint8_t>(p, i));
}
// one wave per sequence: length = number of set mask entries; ok = the mask is a prefix mask (1 .. 1 0 .. 0)
__global__ __launch_bounds__(256) void mask_lengths_kernel(const void* __restrict__ mask, int bytes, int B, int L, long long ld,
                                                           int32_t* __restrict__ lens, int32_t* __restrict__ ok) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  int n = 0, last = -1;  // set entries seen by this lane, the highest position among them
  for (int p = lane; p < L; p += 64)
    if (load_int(mask, (size_t)b * ld + p, bytes) != 0) { ++n; last = p; }
  for (int o = 32; o > 0; o >>= 1) {
    n += __shfl_xor(n, o, 64);
    last = max(last, __shfl_xor(last, o, 64));
  }
  if (lane == 0) { lens[b] = n; ok[b] = (last + 1 == n) ? 1 : 0; }
}
// one workgroup per sequence writes the rows of its extent
__global__ __launch_bounds__(128) void pack_index_kernel(const void* __restrict__ ids, int bytes, long long ld, const int32_t* __restrict__ lens,
                                                         const int32_t* __restrict__ seq_off, int L, int32_t* __restrict__ out_ids,
                                                         int32_t* __restrict__ positions, int32_t* __restrict__ mask,
                                                         int32_t* __restrict__ cls_slot, long long* __restrict__ src) {
  const int b = blockIdx.x;
  const int r0 = seq_off[b], ext = seq_off[b + 1] - r0, len = lens[b];
  for (int p = threadIdx.x; p < ext; p += 128) {
    const int m = p < len ? 1 : 0;
    out_ids[r0 + p] = m ? load_int(ids, (size_t)b * ld + p, bytes) : 0;
    positions[r0 + p] = p;
    mask[r0 + p] = m;
    cls_slot[r0 + p] = p == 0 ? b : -1;
    if (src) src[r0 + p] = (long long)b * L + p;
  }
}
// The host arithmetic of the packed layout (cocodr_amd.modeling.packed_extents) on the device, ONE workgroup: extents =
// max(len, 1), the rows that make T a multiple of 32 handed to the last sequences with room below cap, running offsets, the
// longest-first order (stable, = numpy argsort(-ext, kind="stable")) and {T, max extent rounded up to 32, all masks prefix masks}.
// plan = lens [B] | seq_off [B + 1] | seq_order [B]
constexpr int PLAN_THREADS = 1024, PLAN_MAX_B = 4096;
__global__ __launch_bounds__(PLAN_THREADS) void pack_plan_kernel(const int32_t* __restrict__ lens, const int32_t* __restrict__ ok, int B, int cap,
                                                                 int32_t* __restrict__ plan, int32_t* __restrict__ result) {
  __shared__ int ext[PLAN_MAX_B];
  __shared__ int wsum[PLAN_THREADS / 64], wmax[PLAN_THREADS / 64], wok[PLAN_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (B + PLAN_THREADS - 1) / PLAN_THREADS, b0 = tid * per, b1 = min(B, b0 + per);
  int s = 0, good = 1;
  for (int b = b0; b < b1; ++b) {
    const int n = lens[b], e = max(n, 1);
    ext[b] = e;
    plan[b] = n;
    s += e;
    good &= (ok[b] != 0 && n <= cap) ? 1 : 0;
  }
  // inclusive scan of the per-thread sums: lanes, then waves
  int incl = s;
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  good = __all(good) ? 1 : 0;
  if (lane == 63) wsum[wave] = incl;
  if (lane == 0) wok[wave] = good;
  __syncthreads();
  int base = 0, total = 0, all_ok = 1;
  for (int w = 0; w < PLAN_THREADS / 64; ++w) {
    if (w < wave) base += wsum[w];
    total += wsum[w];
    all_ok &= wok[w];
  }
  // the <= 31 rows that make T a multiple of 32 go to the last sequences with room (one thread: a handful of iterations)
  const int pad = (32 - (total & 31)) & 31;
  __syncthreads();
  if (tid == 0) {
    int left = pad;
    for (int b = B - 1; b >= 0 && left > 0; --b) {
      const int give = min(left, max(cap - ext[b], 0));
      ext[b] += give;
      left -= give;
    }
    wsum[0] = left;  // (0 unless a length exceeds cap: reported through `result`)
  }
  __syncthreads();
  const int left = wsum[0];
  __syncthreads();
  // offsets with the adjusted extents (the adjustment touched the tail only, but a rescan is as cheap as patching it)
  s = 0;
  int mx = 0;
  for (int b = b0; b < b1; ++b) { s += ext[b]; mx = max(mx, ext[b]); }
  incl = s;
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
  if (lane == 63) wsum[wave] = incl;
  if (lane == 0) wmax[wave] = mx;
  __syncthreads();
  base = 0;
  mx = 0;
  for (int w = 0; w < PLAN_THREADS / 64; ++w) {
    if (w < wave) base += wsum[w];
    mx = max(mx, wmax[w]);
  }
  int run = base + incl - s;  // exclusive prefix of this thread's first sequence
  for (int b = b0; b < b1; ++b) {
    plan[B + b] = run;
    run += ext[b];
  }
  if (tid == 0) plan[2 * B] = total + pad - left;
  // longest first, ties in batch order: the rank of b counts the longer extents and the equal ones in front of it
  for (int b = tid; b < B; b += PLAN_THREADS) {
    const int e = ext[b];
    int rank = 0;
    for (int j = 0; j < B; ++j) {
      const int ej = ext[j];
      rank += (ej > e || (ej == e && j < b)) ? 1 : 0;
    }
    plan[2 * B + 1 + rank] = b;
  }
  if (tid == 0) {
    result[0] = total + pad - left;
    result[1] = (mx + 31) / 32 * 32;
    result[2] = (all_ok && left == 0) ? 1 : 0;
    result[3] = B;
  }
}
}  // namespace

extern "C" int cocodr_pack_plan(const int32_t* lens, const int32_t* prefix_ok, int B, int cap, int32_t* plan, int32_t* result,
                                cocodr_stream_t stream) {
  CK_ARG(lens && prefix_ok && plan && result, "pack_plan: null pointer");
  CK_ARG(B > 0 && B <= PLAN_MAX_B && cap > 0 && cap % 32 == 0, "pack_plan: bad arguments (B <= 4096, cap a multiple of 32)");
  hipLaunchKernelGGL(pack_plan_kernel, dim3(1), dim3(PLAN_THREADS), 0, (hipStream_t)stream, lens, prefix_ok, B, cap, plan, result);
  CK_LAUNCH("pack_plan");
  return COCODR_OK;
}

extern "C" int cocodr_mask_lengths(const void* mask, int elem_bytes, int B, int L, long long row_stride, int32_t* lens, int32_t* prefix_ok,
                                   cocodr_stream_t stream) {
  CK_ARG(mask && lens && prefix_ok, "mask_lengths: null pointer");
  CK_ARG(B > 0 && L > 0 && row_stride >= L && (elem_bytes == 1 || elem_bytes == 4 || elem_bytes == 8), "mask_lengths: bad arguments");
  hipLaunchKernelGGL(mask_lengths_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, mask, elem_bytes, B, L, row_stride, lens, prefix_ok);
  CK_LAUNCH("mask_lengths");
  return COCODR_OK;
}

extern "C" int cocodr_pack_index(const void* ids, int elem_bytes, long long row_stride, const int32_t* lens, const int32_t* seq_off, int B,
                                 int L, int32_t* out_ids, int32_t* positions, int32_t* mask, int32_t* cls_slot, long long* src,
                                 cocodr_stream_t stream) {
  CK_ARG(ids && lens && seq_off && out_ids && positions && mask && cls_slot, "pack_index: null pointer");
  CK_ARG(B > 0 && L > 0 && L % 32 == 0 && (elem_bytes == 4 || elem_bytes == 8) && row_stride > 0, "pack_index: bad arguments (L must be a multiple of 32)");
  hipLaunchKernelGGL(pack_index_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, ids, elem_bytes, row_stride, lens, seq_off, L, out_ids,
                     positions, mask, cls_slot, src);
  CK_LAUNCH("pack_index");
  return COCODR_OK;
}

extern "C" int cocodr_scatter_cls_grad(const float* dE, uint16_t* d_last, int B, int L, int H, cocodr_stream_t stream) {
  CK_ARG(dE && d_last, "scatter_cls_grad: null pointer");
  CK_ARG(B > 0 && L > 0 && row_shape_ok(H), "scatter_cls_grad: bad shape");
  const int M = B * L;
  hipLaunchKernelGGL(scatter_cls_kernel, dim3(row_grid(M)), dim3(256), 0, (hipStream_t)stream, dE, d_last, M, L, H);
  CK_LAUNCH("scatter_cls_grad");
  return COCODR_OK;
}

// ------------------------------------------------------------------ fused AdamW over a flat parameter
// torch.optim.AdamW semantics (decoupled weight decay, bias correction, eps outside the sqrt of the corrected
// second moment), one pass over p / g / m / v (16 B-per-lane vectors) that also refreshes the bf16 shadow of
// the weight matrices, so the separate fp32->bf16 cast pass disappears from the step.
namespace {
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, uint16_t* __restrict__ shadow, size_t shadow_begin,
                                                    size_t n4, float lr, float beta1, float beta2, float eps, float wd,
                                                    float bc1, float bc2_sqrt, float grad_scale,
                                                    const float* __restrict__ grad_scale_dev) {
  if (grad_scale_dev) grad_scale *= *grad_scale_dev;  // e.g. the clip coefficient cocodr_grad_norm_clip left on the device
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    // one pass over 30 B / parameter that nothing re-reads soon: non-temporal loads and stores (kernel -10 %)
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v pq = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p) + i), gq = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(g) + i);
    const f4v mq = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(m) + i), vq = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(v) + i);
    float4 pv = make_float4(pq.x, pq.y, pq.z, pq.w); const float4 gv = make_float4(gq.x, gq.y, gq.z, gq.w);
    float4 mv = make_float4(mq.x, mq.y, mq.z, mq.w); float4 vv = make_float4(vq.x, vq.y, vq.z, vq.w);
    float pa[4] = {pv.x, pv.y, pv.z, pv.w};
    const float ga[4] = {gv.x * grad_scale, gv.y * grad_scale, gv.z * grad_scale, gv.w * grad_scale};
    float ma[4] = {mv.x, mv.y, mv.z, mv.w};
    float va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pa[e] *= (1.0f - lr * wd);
      ma[e] = beta1 * ma[e] + (1.0f - beta1) * ga[e];
      va[e] = beta2 * va[e] + (1.0f - beta2) * ga[e] * ga[e];
      const float denom = sqrtf(va[e]) / bc2_sqrt + eps;
      pa[e] -= (lr / bc1) * (ma[e] / denom);
    }
    { f4v t = {pa[0], pa[1], pa[2], pa[3]}; __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(p) + i); }
    { f4v t = {ma[0], ma[1], ma[2], ma[3]}; __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(m) + i); }
    { f4v t = {va[0], va[1], va[2], va[3]}; __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(v) + i); }
    if (shadow && i * 4 >= shadow_begin) *reinterpret_cast<uint2*>(shadow + (i * 4 - shadow_begin)) = pack4(pa);
  }
}

// ---- gradient norm / clip coefficient (torch.nn.utils.clip_grad_norm_, ANCE/drivers/run_ann.py:347-352), no host sync
constexpr int GN_BLOCKS = 1024;
template <int THREADS, bool NT>
__global__ __launch_bounds__(THREADS) void sumsq_kernel(const float* __restrict__ x, size_t n, float* __restrict__ partial) {
  __shared__ float red[THREADS / 64];
  float acc = 0.f;
  const size_t n4 = n / 4;
  const size_t chunk = (n4 + gridDim.x - 1) / gridDim.x;  // a contiguous run per block (a grid stride put its iterations MBs apart)
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n4 ? lo + chunk : n4;
  typedef float f4v __attribute__((ext_vector_type(4)));
  for (size_t i = lo + threadIdx.x; i < hi; i += THREADS) {
    const f4v v = NT ? __builtin_nontemporal_load(reinterpret_cast<const f4v*>(x) + i) : reinterpret_cast<const f4v*>(x)[i];
    acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float t = x[n4 * 4 + threadIdx.x]; acc += t * t; }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) s += red[w];
    partial[blockIdx.x] = s;
  }
}
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partial, int np, float max_norm, float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    out[0] = norm;
    out[1] = fminf(1.0f, max_norm / (norm + 1e-6f));
  }
}

// ---- LAMB (ANCE/utils/lamb.py:61-121): Adam moments without bias correction, per-tensor trust ratio
// clamp(||w||, 0, 10) / ||m / (sqrt(v) + eps) + wd * w||.  The flat parameter is cut into chunks that never straddle a
// tensor (host-built plan): pass 1 updates m, v and leaves per-chunk (sum w^2, sum u^2), a small kernel turns them into
// one trust ratio per tensor (chunks added in a fixed order: deterministic), pass 2 re-forms u and applies the step.
__device__ __forceinline__ float lamb_u(float w, float m, float v, float eps, float wd) { return m / (sqrtf(v) + eps) + wd * w; }

__global__ __launch_bounds__(256) void lamb_moments_kernel(const float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, const long long* __restrict__ chunk_start,
                                                           const int* __restrict__ chunk_len, float beta1, float beta2, float eps,
                                                           float wd, float grad_scale, const float* __restrict__ grad_scale_dev,
                                                           float* __restrict__ chunk_sums) {
  __shared__ float red[2][4];
  if (grad_scale_dev) grad_scale *= *grad_scale_dev;
  const size_t base = (size_t)chunk_start[blockIdx.x] / 4;
  const int len4 = chunk_len[blockIdx.x] / 4;
  float sw = 0.f, su = 0.f;
  for (int i = threadIdx.x; i < len4; i += 256) {
    typedef float f4v __attribute__((ext_vector_type(4)));  // (42 B / parameter over the two passes that nothing re-reads soon)
    const f4v pv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p) + base + i);
    const f4v gv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(g) + base + i);
    const f4v mv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(m) + base + i);
    const f4v vv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(v) + base + i);
    const float pa[4] = {pv.x, pv.y, pv.z, pv.w};
    const float ga[4] = {gv.x * grad_scale, gv.y * grad_scale, gv.z * grad_scale, gv.w * grad_scale};
    float ma[4] = {mv.x, mv.y, mv.z, mv.w};
    float va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ma[e] = beta1 * ma[e] + (1.0f - beta1) * ga[e];
      va[e] = beta2 * va[e] + (1.0f - beta2) * ga[e] * ga[e];
      const float u = lamb_u(pa[e], ma[e], va[e], eps, wd);
      sw += pa[e] * pa[e];
      su += u * u;
    }
    { f4v t = {ma[0], ma[1], ma[2], ma[3]}; __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(m) + base + i); }
    { f4v t = {va[0], va[1], va[2], va[3]}; __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(v) + base + i); }
  }
  sw = wave_sum(sw);
  su = wave_sum(su);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sw; red[1][threadIdx.x >> 6] = su; }
  __syncthreads();
  if (threadIdx.x == 0) {
    chunk_sums[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    chunk_sums[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}
// one wave per tensor: trust[s] = clamp(||w||, 0, 10) / ||u||  (1 when either norm is 0); stats [s] = (||w|| clamped, ||u||)
__global__ __launch_bounds__(64) void lamb_trust_kernel(const float* __restrict__ chunk_sums, const int* __restrict__ seg_chunk_begin,
                                                        float* __restrict__ trust, float* __restrict__ stats) {
  const int s = blockIdx.x, lane = threadIdx.x;
  if (seg_chunk_begin[s] == seg_chunk_begin[s + 1]) return;  // a tensor the one-pass kernel owns (cocodr_lamb_step_fused): no chunks here
  float sw = 0.f, su = 0.f;
  for (int c = seg_chunk_begin[s] + lane; c < seg_chunk_begin[s + 1]; c += 64) { sw += chunk_sums[2 * c]; su += chunk_sums[2 * c + 1]; }
  sw = wave_sum(sw);
  su = wave_sum(su);
  if (lane == 0) {
    const float wn = fminf(sqrtf(sw), 10.0f), un = sqrtf(su);
    trust[s] = (wn == 0.f || un == 0.f) ? 1.0f : wn / un;
    if (stats) { stats[2 * s] = wn; stats[2 * s + 1] = un; }
  }
}
__global__ __launch_bounds__(256) void lamb_apply_kernel(float* __restrict__ p, const float* __restrict__ m, const float* __restrict__ v,
                                                         uint16_t* __restrict__ shadow, size_t shadow_begin,
                                                         const long long* __restrict__ chunk_start, const int* __restrict__ chunk_len,
                                                         const int* __restrict__ chunk_seg, const float* __restrict__ trust, float lr,
                                                         float eps, float wd) {
  const size_t base = (size_t)chunk_start[blockIdx.x] / 4;
  const int len4 = chunk_len[blockIdx.x] / 4;
  const float step = lr * trust[chunk_seg[blockIdx.x]];
  for (int i = threadIdx.x; i < len4; i += 256) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v pv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p) + base + i);
    const f4v mv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(m) + base + i);
    const f4v vv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(v) + base + i);
    float pa[4] = {pv.x, pv.y, pv.z, pv.w};
    const float ma[4] = {mv.x, mv.y, mv.z, mv.w};
    const float va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) pa[e] -= step * lamb_u(pa[e], ma[e], va[e], eps, wd);
    { f4v t = {pa[0], pa[1], pa[2], pa[3]}; __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(p) + base + i); }
    const size_t el = (base + i) * 4;
    if (shadow && el >= shadow_begin) *reinterpret_cast<uint2*>(shadow + (el - shadow_begin)) = pack4(pa);
  }
}

// ---- LAMB in ONE pass over a tensor (round 5).  The two-pass form above moves 42 B / parameter because the trust ratio of a tensor
// needs ||u|| of the WHOLE tensor before any element can be updated, so p, m, v are read a second time.  A weight matrix of the
// encoder (<= 4 M elements) fits the chip's register files: here a persistent grid of G co-resident workgroups spreads every tensor
// over all of them; pass 1 (m, v, u; w and u stay on chip - registers, then LDS) leaves ONE 16-byte granule (sum w^2, sum u^2, each next to the
// launch's epoch tag) per workgroup, and one tensor later every workgroup reads the G granules, adds them in the same fixed order
// (deterministic, the same ratio everywhere) and applies the step from its registers: 30 B / parameter (AdamW's traffic).
// The element stream is software-pipelined across tensor boundaries (the loads of the next float4 quadruple are in flight while the
// current one is worked on), and nobody waits for anybody in the common case: a granule is read a whole tensor after it was written.
// Inter-workgroup visibility without fences: granules are written with ONE write-through (sc1) 16-byte store and read with sc1
// loads; each 8-byte half carries its own tag, so a torn read is recognised and repeated (MI355X_MICROARCH.md "Workgroup dispatch
// ... visibility": data-tagged granules).  A release fence here would write back the XCD's whole L2 - tens of MB of freshly
// streamed m / v lines - per workgroup and tensor (measured: 6.5x slower than the two-pass kernels).  The spin is bounded: a grid
// that is not co-resident (it is sized from the occupancy query, which knows nothing of CU masks, partitions or a second process)
// raises *err; a workgroup that sees the flag applies NO step to its tensor (trust ratio 0) - a skipped update and a loud flag
// instead of a hung queue or wrongly scaled weights.  FlatLamb.step reads the flag on its first step and every 64th and falls
// back to the two-pass kernels for good when it is set.
constexpr int LF_THREADS = 1024, LF_V = 4, LF_PER_CU = 1;  // ONE 1024-thread workgroup per CU (G = #CUs: a gather reads G granules - with 256-thread
                                                            // workgroups, 4 per CU, the G^2 granule reads were half of the tensors' own traffic); capacity G * 1024 * 16 floats
typedef float lf4 __attribute__((ext_vector_type(4)));
typedef uint32_t lu4 __attribute__((ext_vector_type(4)));
typedef uint32_t lu2 __attribute__((ext_vector_type(2)));
struct LambFusedArgs {
  float* p; const float* g; float* m; float* v; uint16_t* shadow; size_t shadow_begin;
  const long long* seg_start; const int* seg_len; const int* seg_index; const int* wg_begin; const int* wg_count; const int* round_first;
  int nfused, nrounds;
  float beta1, beta2, eps, wd, grad_scale, lr; const float* grad_scale_dev;
  float* part; uint32_t epoch; int* err; float* trust; float* stats;
};
__global__ __launch_bounds__(LF_THREADS, LF_PER_CU * LF_THREADS / 256) void lamb_fused_kernel(const LambFusedArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ float red[2][LF_THREADS / 64];
  __shared__ lf4 keep[LF_V][2][LF_THREADS];  // w and u of the PREVIOUS round (128 KB: thread-private slots, no barriers), see below
  const int tid = threadIdx.x, bid = blockIdx.x, G = gridDim.x;
  const float gs = a.grad_scale * (a.grad_scale_dev ? *a.grad_scale_dev : 1.0f);
  auto block_sum2 = [&](float& x, float& y) {  // both sums over the workgroup, the same value in every thread (fixed order)
    x = wave_sum(x);
    y = wave_sum(y);
    __syncthreads();  // (red is reused)
    if ((tid & 63) == 0) { red[0][tid >> 6] = x; red[1][tid >> 6] = y; }
    __syncthreads();
    x = y = 0.f;
#pragma unroll
    for (int w = 0; w < LF_THREADS / 64; ++w) { x += red[0][w]; y += red[1][w]; }
  };
  // A tensor is addressed through buffer descriptors that END with it: a lane whose float4 lies behind the tensor loads zeros
  // (u = 0: no contribution to either norm) and its stores are dropped by the bounds check - no branches in the element stream, and
  // 32-bit offsets.  aux 2 = non-temporal (streamed once), aux 16 = sc1 (write-through / past the L1).
  auto uni = [](int x) { return __builtin_amdgcn_readfirstlane(x); };  // (tell hipcc the value is wave-uniform: a descriptor it believes
  auto uni64 = [](unsigned long long x) {                               //  divergent puts every buffer instruction into a waterfall loop)
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return ((unsigned long long)hi << 32) | lo;
  };
  auto rsrc = [&](const void* base, size_t elem0, size_t n_elem, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + elem0 * bytes), 0, (uint32_t)(n_elem * bytes), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rgran = rsrc(a.part, 0, (size_t)a.nfused * G * 4, 4);
  // A ROUND holds one or more tensors, each spread over its own range of workgroups (the host sizes the ranges by tensor length:
  // BERT's 1 M-element Wq / Wk / Wv / Wo share a round, a 4 M-element FFN matrix has one to itself), so the per-round tail - granule,
  // gather, two workgroup sums - is paid once per ~4 M elements and a gather reads only the granules of the workgroup's own tensor.
  // This workgroup's tensor of round r: entry e (-1: none, the workgroup idles through the round), first element, length, its index
  // among the tensor's workgroups and their number.
  struct Slot { int e; size_t e0, n; int local, cnt; };
  auto slot = [&](int r) {
    Slot s{-1, 0, 0, 0, 1};
    for (int e = uni(a.round_first[r]); e < uni(a.round_first[r + 1]); ++e) {
      const int b0 = uni(a.wg_begin[e]), c = uni(a.wg_count[e]);
      if (bid >= b0 && bid < b0 + c) {
        s.e = e;
        s.e0 = (size_t)uni64((unsigned long long)a.seg_start[e]);
        s.n = (size_t)uni64((unsigned long long)a.seg_len[e]);
        s.local = bid - b0;
        s.cnt = c;
      }
    }
    return s;
  };
  struct Quad { lf4 p, g, m, v; };
  struct Desc { __amdgpu_buffer_rsrc_t p, g, m, v; uint32_t off0, offj; };  // byte offset of this thread's float4 j: off0 + j offj
  auto desc = [&](const Slot& s) {
    return Desc{rsrc(a.p, s.e0, s.n, 4), rsrc(a.g, s.e0, s.n, 4), rsrc(a.m, s.e0, s.n, 4), rsrc(a.v, s.e0, s.n, 4),
                (uint32_t)(s.local * LF_THREADS + tid) * 16u, (uint32_t)s.cnt * LF_THREADS * 16u};
  };
  auto load = [&](const Desc& d, int j) {
    const uint32_t o = d.off0 + (uint32_t)j * d.offj;
    Quad q;
    q.p = __builtin_bit_cast(lf4, __builtin_amdgcn_raw_buffer_load_b128(d.p, o, 0, 2));
    q.g = __builtin_bit_cast(lf4, __builtin_amdgcn_raw_buffer_load_b128(d.g, o, 0, 2));
    q.m = __builtin_bit_cast(lf4, __builtin_amdgcn_raw_buffer_load_b128(d.m, o, 0, 2));
    q.v = __builtin_bit_cast(lf4, __builtin_amdgcn_raw_buffer_load_b128(d.v, o, 0, 2));
    return q;
  };
  // one float4 of pass 1: m, v updated in memory; P = the weights, U = the update direction stay in registers
  auto work = [&](const Desc& d, int j, Quad q, lf4& P, lf4& U, float& sw, float& su) {
    const uint32_t o = d.off0 + (uint32_t)j * d.offj;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = q.g[e] * gs;
      q.m[e] = a.beta1 * q.m[e] + (1.0f - a.beta1) * ge;
      q.v[e] = a.beta2 * q.v[e] + (1.0f - a.beta2) * ge * ge;
      const float u = lamb_u(q.p[e], q.m[e], q.v[e], a.eps, a.wd);
      U[e] = u;
      sw += q.p[e] * q.p[e];
      su += u * u;
    }
    P = q.p;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lu4, q.m), d.m, o, 0, 2);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lu4, q.v), d.v, o, 0, 2);
  };
  auto publish = [&](const Slot& s, float sw, float su) {
    block_sum2(sw, su);
    if (tid == 0 && s.e >= 0) {
      const lu4 gr = {__float_as_uint(sw), a.epoch, __float_as_uint(su), a.epoch};
      __builtin_amdgcn_raw_buffer_store_b128(gr, rgran, (uint32_t)(((size_t)s.e * G + s.local) * 16), 0, 16);
    }
  };
  // the trust ratio of this workgroup's tensor from its granules (fixed order: thread t adds granules t, t + 1024, ...; then the block sum)
  auto gather = [&](const Slot& s) {
    float sw = 0.f, su = 0.f;
    for (int w = tid; w < s.cnt && s.e >= 0; w += LF_THREADS) {
      const uint32_t o = (uint32_t)(((size_t)s.e * G + w) * 16);
      lu4 gr = __builtin_amdgcn_raw_buffer_load_b128(rgran, o, 0, 16);
      int spins = 0;
      while (gr[1] != a.epoch || gr[3] != a.epoch) {  // not there yet (or torn): rare - it was written a whole round ago
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1 << 24)) { *a.err = 1; break; }  // seconds: the grid is not co-resident
        gr = __builtin_amdgcn_raw_buffer_load_b128(rgran, o, 0, 16);
      }
      sw += __uint_as_float(gr[0]);
      su += __uint_as_float(gr[2]);
    }
    block_sum2(sw, su);
    const float wn = fminf(sqrtf(sw), 10.0f), un = sqrtf(su);
    // a gather that gave up (this workgroup's or, as far as it is visible here, anybody's) holds partial norms: ratio 0 - the
    // tensor keeps its weights (P - lr * 0 * U) instead of taking a wrongly scaled step; the host reads the flag (optim.py)
    const bool failed = __builtin_nontemporal_load(a.err) != 0;
    const float tr = failed ? 0.0f : ((wn == 0.f || un == 0.f) ? 1.0f : wn / un);
    if (s.e >= 0 && s.local == 0 && tid == 0) {
      const int si = a.seg_index[s.e];
      a.trust[si] = tr;
      if (a.stats) { a.stats[2 * si] = wn; a.stats[2 * si + 1] = un; }
    }
    return tr;
  };
  auto apply = [&](const Slot& s, float tr) {  // w, u of the slot's tensor come back from this thread's LDS slots
    const float step = a.lr * tr;
    const __amdgpu_buffer_rsrc_t rp = rsrc(a.p, s.e0, s.n, 4);
    const bool shadowed = a.shadow != nullptr && s.e0 >= a.shadow_begin;  // (workgroup-uniform)
    const __amdgpu_buffer_rsrc_t rs = rsrc(a.shadow, shadowed ? s.e0 - a.shadow_begin : 0, shadowed ? s.n : 0, 2);
    const uint32_t off0 = (uint32_t)(s.local * LF_THREADS + tid) * 16u, offj = (uint32_t)s.cnt * LF_THREADS * 16u;
#pragma unroll
    for (int j = 0; j < LF_V; ++j) {
      const uint32_t o = off0 + (uint32_t)j * offj;
      const lf4 P = keep[j][0][tid], U = keep[j][1][tid];
      float pa[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) pa[e] = P[e] - step * U[e];
      const lf4 t = {pa[0], pa[1], pa[2], pa[3]};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lu4, t), rp, o, 0, 2);
      const uint2 h = pack4(pa);
      const lu2 hv = {h.x, h.y};
      __builtin_amdgcn_raw_buffer_store_b64(hv, rs, o >> 1, 0, 0);  // (num_records 0 when the tensor has no shadow: dropped)
    }
  };
  // The stream: a round's quadruples with TWO requests ahead of the one being worked on (across the round boundary too: the tail of
  // a round - its granule, the previous round's ratio and update - runs with the next round's first two quadruples in flight).  The
  // current round's w, u are in registers; behind its tail they move to this thread's LDS slots, where the update of the previous
  // round has just been read from.
  auto unit = [&](int u) {  // request quadruple u of the flattened (round, j) sequence; past the end: a repeat of the last (not used)
    const int total = a.nrounds * LF_V;
    const int uu = u < total ? u : total - 1;
    const int r_ = uu / LF_V;
    return load(desc(slot(r_)), uu - r_ * LF_V);
  };
  Quad q0 = unit(0), q1 = unit(1);
  Slot prev{-1, 0, 0, 0, 1};
#pragma unroll 1
  for (int r = 0; r < a.nrounds; ++r) {
    lf4 P[LF_V], U[LF_V];
    float sw = 0.f, su = 0.f;
    const Slot cur = slot(r);
    const Desc d = desc(cur);
#pragma unroll
    for (int j = 0; j < LF_V; ++j) {
      const Quad q2 = unit(r * LF_V + j + 2);
      work(d, j, q0, P[j], U[j], sw, su);
      q0 = q1;
      q1 = q2;
      __builtin_amdgcn_sched_barrier(0);  // (keep exactly two quadruples of requests ahead)
    }
    publish(cur, sw, su);
    if (r > 0) apply(prev, gather(prev));
#pragma unroll
    for (int j = 0; j < LF_V; ++j) { keep[j][0][tid] = P[j]; keep[j][1][tid] = U[j]; }
    prev = cur;
  }
  apply(prev, gather(prev));
#endif
}
// workgroups of the persistent grid: what is certainly co-resident per the occupancy query, at most LF_PER_CU per CU; 0 = the kernel
// cannot run here
int lamb_fused_grid() {
  static int grid = -1;
  if (grid < 0) {
    int dev = 0, n_cu = 0, occ = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, lamb_fused_kernel, LF_THREADS, 0) != hipSuccess) {
      (void)hipGetLastError();
      grid = 0;
    } else {
      // (the query is one block per CU high only where it answers 7 or 8 for 256-thread blocks - MI355X_MICROARCH.md "Residency")
      const int safe = occ >= 7 ? occ - 1 : occ;
      const int per_cu = safe < LF_PER_CU ? safe : LF_PER_CU;
      grid = per_cu > 0 && n_cu > 0 ? per_cu * n_cu : 0;
    }
  }
  return grid;
}
}  // namespace

extern "C" size_t cocodr_lamb_fused_capacity(void) { return (size_t)lamb_fused_grid() * LF_THREADS * LF_V * 4; }
extern "C" int cocodr_lamb_fused_workgroups(void) { return lamb_fused_grid(); }
extern "C" size_t cocodr_lamb_fused_workgroup_elements(void) { return (size_t)LF_THREADS * LF_V * 4; }
extern "C" size_t cocodr_lamb_fused_workspace_floats(int nfused) {
  return nfused > 0 ? (size_t)nfused * lamb_fused_grid() * 4 + 4 : 0;
}
extern "C" size_t cocodr_lamb_fused_error_index(int nfused) { return (size_t)nfused * lamb_fused_grid() * 4; }
extern "C" int cocodr_lamb_step_fused(float* p, const float* g, float* m, float* v, uint16_t* shadow, size_t shadow_begin,
                                      const cocodr_lamb_fused_plan* plan, float lr, float beta1, float beta2, float eps, float weight_decay,
                                      float grad_scale, const float* grad_scale_dev, float* workspace, float* trust, float* stats,
                                      cocodr_stream_t stream) {
  CK_ARG(p && g && m && v && plan && workspace && trust, "lamb_step_fused: null pointer");
  CK_ARG(plan->seg_start && plan->seg_len && plan->seg_index && plan->wg_begin && plan->wg_count && plan->round_first && plan->nfused > 0 &&
             plan->nrounds > 0 && plan->nrounds <= plan->nfused, "lamb_step_fused: incomplete plan");
  CK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)workspace) & 15) == 0 && (((uintptr_t)shadow) & 7) == 0 &&
             shadow_begin % 4 == 0, "lamb_step_fused: pointers must be 16-byte aligned");
  CK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "lamb_step_fused: bad hyper-parameters");
  const int G = lamb_fused_grid();
  CK_ARG(G > 0, "lamb_step_fused: no co-resident grid on this device (cocodr_lamb_fused_capacity() == 0): use cocodr_lamb_step");
  CK_ARG((size_t)plan->nfused * G * 16 < (1ull << 32), "lamb_step_fused: too many tensors for one call");
  // the tag of this call's granules: any value the previous calls on this workspace did not use (the workspace starts zeroed)
  static std::atomic<uint32_t> epoch_counter{0};
  uint32_t epoch = ++epoch_counter;
  if (epoch == 0) epoch = ++epoch_counter;
  LambFusedArgs a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.shadow = shadow; a.shadow_begin = shadow_begin;
  a.seg_start = plan->seg_start; a.seg_len = plan->seg_len; a.seg_index = plan->seg_index; a.nfused = plan->nfused;
  a.wg_begin = plan->wg_begin; a.wg_count = plan->wg_count; a.round_first = plan->round_first; a.nrounds = plan->nrounds;
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay; a.grad_scale = grad_scale; a.lr = lr; a.grad_scale_dev = grad_scale_dev;
  a.part = workspace;
  a.epoch = epoch;
  a.err = reinterpret_cast<int*>(workspace + (size_t)plan->nfused * G * 4);
  a.trust = trust; a.stats = stats;
  hipLaunchKernelGGL(lamb_fused_kernel, dim3(G), dim3(LF_THREADS), 0, (hipStream_t)stream, a);
  CK_LAUNCH("lamb_step_fused");
  return COCODR_OK;
}

extern "C" int cocodr_grad_norm_clip(const float* const* grads, const size_t* numels, int count, float max_norm, float* partial,
                                     float* out, cocodr_stream_t stream) {
  CK_ARG(grads && numels && partial && out && count > 0 && count <= 8, "grad_norm_clip: need 1..8 tensors, workspace and output");
  hipStream_t st = (hipStream_t)stream;
  for (int t = 0; t < count; ++t) {
    CK_ARG(grads[t] && (((uintptr_t)grads[t]) & 15) == 0, "grad_norm_clip: tensor %d must be a 16-byte aligned device pointer", t);
    // 1024 threads per block and non-temporal loads: 73 us against 82 for 256-thread blocks on 110 M gradients (tools/optim_time.py)
    hipLaunchKernelGGL((sumsq_kernel<1024, true>), dim3(GN_BLOCKS), dim3(1024), 0, st, grads[t], numels[t], partial + (size_t)t * GN_BLOCKS);
    CK_LAUNCH("grad_norm_clip(sumsq)");
  }
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, st, partial, count * GN_BLOCKS, max_norm, out);
  CK_LAUNCH("grad_norm_clip");
  return COCODR_OK;
}

extern "C" int cocodr_lamb_step(float* p, const float* g, float* m, float* v, uint16_t* shadow, size_t shadow_begin, size_t n,
                                const cocodr_lamb_plan* plan, float lr, float beta1, float beta2, float eps, float weight_decay,
                                float grad_scale, const float* grad_scale_dev, float* workspace, float* stats,
                                cocodr_stream_t stream) {
  CK_ARG(p && g && m && v && plan && workspace, "lamb_step: null pointer");
  CK_ARG(plan->chunk_start && plan->chunk_len && plan->chunk_seg && plan->seg_chunk_begin && plan->nchunk > 0 && plan->nseg > 0,
         "lamb_step: incomplete plan");
  CK_ARG(n % 4 == 0 && shadow_begin % 4 == 0, "lamb_step: n and shadow_begin must be multiples of 4");
  CK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && (((uintptr_t)shadow) & 7) == 0,
         "lamb_step: pointers must be 16-byte aligned");
  CK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "lamb_step: bad hyper-parameters");
  hipStream_t st = (hipStream_t)stream;
  float* chunk_sums = workspace;                       // [nchunk][2]
  float* trust = workspace + (size_t)2 * plan->nchunk;  // [nseg]
  hipLaunchKernelGGL(lamb_moments_kernel, dim3(plan->nchunk), dim3(256), 0, st, p, g, m, v, plan->chunk_start, plan->chunk_len, beta1,
                     beta2, eps, weight_decay, grad_scale, grad_scale_dev, chunk_sums);
  CK_LAUNCH("lamb_step(moments)");
  hipLaunchKernelGGL(lamb_trust_kernel, dim3(plan->nseg), dim3(64), 0, st, chunk_sums, plan->seg_chunk_begin, trust, stats);
  CK_LAUNCH("lamb_step(trust)");
  hipLaunchKernelGGL(lamb_apply_kernel, dim3(plan->nchunk), dim3(256), 0, st, p, m, v, shadow, shadow_begin, plan->chunk_start,
                     plan->chunk_len, plan->chunk_seg, trust, lr, eps, weight_decay);
  CK_LAUNCH("lamb_step(apply)");
  return COCODR_OK;
}

extern "C" int cocodr_adamw_step(float* p, const float* g, float* m, float* v, uint16_t* shadow, size_t shadow_begin, size_t n,
                                 float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                                 const float* grad_scale_dev, cocodr_stream_t stream) {
  CK_ARG(p && g && m && v, "adamw_step: null pointer");
  CK_ARG(n % 4 == 0 && shadow_begin % 4 == 0, "adamw_step: n and shadow_begin must be multiples of 4");
  CK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && (((uintptr_t)shadow) & 7) == 0,
         "adamw_step: pointers must be 16-byte aligned");
  CK_ARG(step >= 1 && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "adamw_step: bad hyper-parameters");
  if (n == 0) return COCODR_OK;
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2 = sqrtf(1.0f - powf(beta2, (float)step));
  const size_t n4 = n / 4;
  // one block per 256 float4 (no grid stride): 4096 grid-striding blocks streamed 5.36 TB/s, this 6.5 (110 M parameters: 615 -> 505 us;
  // tools/adamw_time.py) - a block's successive iterations were 4 MB apart
  static const size_t grid_cap = getenv("COCODR_ADAMW_GRID") ? (size_t)atoll(getenv("COCODR_ADAMW_GRID")) : (size_t)0x7fffffff;  // (A/B knob)
  const int grid = (int)std::min(grid_cap, (n4 + 255) / 256);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, shadow, shadow_begin, n4, lr, beta1,
                     beta2, eps, weight_decay, bc1, bc2, grad_scale, grad_scale_dev);
  CK_LAUNCH("adamw_step");
  return COCODR_OK;
}
