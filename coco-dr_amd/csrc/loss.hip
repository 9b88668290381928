// Training losses of the COCO-DR hot path, fp32 end to end (the logits are raw un-normalised
// dot products of LayerNorm outputs - O(100) in magnitude - so the similarity matrix is formed with
// the exact-fp32 MFMA, v_mfma_f32_32x32x2_f32, not in bf16).
//
//  * simce  : COCO/modeling.py:244-248 compute_contrastive_loss (+ co_target :172-177, mean :229)
//             and the gradient that reaches the LOCAL rows of the gathered [CLS] matrix
//             (COCO/modeling.py:182-186).  With S = E E^T symmetric and t(i) = i ^ 1 an involution,
//                 dL/dE_i = (W/M) * sum_j Gs[i][j] E_j,
//                 Gs[i][j] = exp(S_ij - lse_i) + exp(S_ij - lse_j) - 2 [j == t(i)],  Gs[i][i] = 0
//             so a rank only needs its own row block of S - never G^T - and no backward collective.
//  * triplet: ANCE/model/models.py:97-106 (logits, -log_softmax[:,0]) and :260-261 ((loss*w).mean()).
#include <algorithm>

#include "common.h"

namespace {

constexpr int TS = 64;   // output tile
constexpr int TK = 32;   // contraction chunk
constexpr int TLD = 33;  // padded LDS leading dim (floats)

// C tile [64 x 64] += A[64 x 32] * B^T, A and B staged as [row][k] fp32 tiles (ld 33);
// wave (wm, wn) owns the 32x32 sub-tile.  MFMA 32x32x2 f32: lane l feeds A[i = l&31][k = l>>5].
__device__ __forceinline__ void mma_chunk_nt(const float* As, const float* Bs, int wm, int wn, int lane, f32x16& acc) {
#pragma unroll
  for (int kk = 0; kk < TK; kk += 2) {
    const float a = As[(wm * 32 + (lane & 31)) * TLD + kk + (lane >> 5)];
    const float b = Bs[(wn * 32 + (lane & 31)) * TLD + kk + (lane >> 5)];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
}

// S = E E^T with -inf diagonal.  grid (ceil(M/64), ceil(M/64)), 1024 threads = 4 output sub-tiles x 4 K-splits.
// E is small and L2-resident and the fp32 MFMA chain of one 32x32 tile is latency-bound (64 clocks per dependent
// 32x32x2 step), so the contraction is split over four waves per sub-tile; operands go global -> registers as float4
// (lane half h feeds k = 8g + 4h + e: any k assignment shared by both operands is a valid contraction order) and the
// four partial tiles are added in a fixed order through LDS (deterministic).
__global__ __launch_bounds__(1024) void simce_scores_kernel(const float* __restrict__ E, float* __restrict__ S, int M, int H) {
  __shared__ float part[3 * 4 * 64 * 16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int sub = wid & 3, ks = wid >> 2, wm = sub >> 1, wn = sub & 1, half = lane >> 5;
  const int i0 = blockIdx.y * TS, j0 = blockIdx.x * TS;
  const int ri = i0 + wm * 32 + (lane & 31), rj = j0 + wn * 32 + (lane & 31);
  const float* __restrict__ ea = E + (size_t)min(ri, M - 1) * H;
  const float* __restrict__ eb = E + (size_t)min(rj, M - 1) * H;
  const bool va = ri < M, vb = rj < M;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int ng = (H + 7) / 8;
#pragma unroll 4
  for (int g = ks; g < ng; g += 4) {
    const int k = 8 * g + 4 * half;
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
    if (k + 3 < H && (H & 3) == 0) {
      const float4 a4 = *reinterpret_cast<const float4*>(ea + k), b4 = *reinterpret_cast<const float4*>(eb + k);
      a[0] = a4.x; a[1] = a4.y; a[2] = a4.z; a[3] = a4.w;
      b[0] = b4.x; b[1] = b4.y; b[2] = b4.z; b[3] = b4.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k + e < H) { a[e] = ea[k + e]; b[e] = eb[k + e]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va ? a[e] : 0.f, vb ? b[e] : 0.f, acc, 0, 0, 0);
  }
  if (ks > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) part[(((ks - 1) * 4 + sub) * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (ks == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r];
#pragma unroll
      for (int q = 0; q < 3; ++q) v += part[((q * 4 + sub) * 16 + r) * 64 + lane];
      const int i = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int j = j0 + wn * 32 + (lane & 31);
      if (i < M && j < M) S[(size_t)i * M + j] = (i == j) ? -INFINITY : v;
    }
  }
}

// one wave per row: lse_i and loss_i = (lse_i - S[i][i^1]) * W
__global__ __launch_bounds__(256) void simce_rowstats_kernel(const float* __restrict__ S, float* __restrict__ lse,
                                                             float* __restrict__ loss_rows, int M, float world) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= M) return;
  const float* row = S + (size_t)i * M;
  float mx = -INFINITY;
  for (int j = lane; j < M; j += 64) mx = fmaxf(mx, row[j]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int j = lane; j < M; j += 64) s += __expf(row[j] - mx);
  s = wave_sum(s);
  const float l = mx + __logf(s);
  if (lane == 0) {
    lse[i] = l;
    loss_rows[i] = (l - row[i ^ 1]) * world;
  }
}

// out[0] = scale * sum_i x[i] * (w ? w[i] : 1)   (single workgroup, fixed order -> deterministic)
__global__ __launch_bounds__(256) void weighted_mean_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ out, int n, float scale) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i] * (w ? w[i] : 1.f);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) * scale;
}

// dE_local[r][h] = coef * sum_j Gs[row0 + r][j] * E[j][h].  grid (ceil(H/64), ceil(m_local/64)).
__global__ __launch_bounds__(256) void simce_grad_kernel(const float* __restrict__ E, const float* __restrict__ S,
                                                         const float* __restrict__ lse, float* __restrict__ dE, int M, int H,
                                                         int row0, int m_local, float coef) {
  __shared__ float Gs[TS * TLD];  // [r][j-chunk]
  __shared__ float Bs[TS * TLD];  // [h][j-chunk]  (E chunk stored transposed so both operands are [row][k])
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
  const int r0 = blockIdx.y * TS, h0 = blockIdx.x * TS;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int j0 = 0; j0 < M; j0 += TK) {
    for (int q = tid; q < TS * TK; q += 256) {
      const int r = q >> 5, k = q & 31;
      const int i = row0 + r0 + r, j = j0 + k;
      float g = 0.f;
      if (r0 + r < m_local && j < M && j != i) {
        const float s = S[(size_t)i * M + j];
        g = __expf(s - lse[i]) + __expf(s - lse[j]) - ((j == (i ^ 1)) ? 2.f : 0.f);
      }
      Gs[r * TLD + k] = g;
    }
    for (int q = tid; q < TS * TK; q += 256) {
      const int k = q >> 6, hh = q & 63;  // coalesced along h
      Bs[hh * TLD + k] = (j0 + k < M && h0 + hh < H) ? E[(size_t)(j0 + k) * H + h0 + hh] : 0.f;
    }
    __syncthreads();
    mma_chunk_nt(Gs, Bs, wm, wn, lane, acc);
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rr = r0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int hh = h0 + wn * 32 + (lane & 31);
    if (rr < m_local && hh < H) dE[(size_t)rr * H + hh] = acc[r] * coef;
  }
}

// one wave per triplet row
__global__ __launch_bounds__(256) void triplet_kernel(const float* __restrict__ q, const float* __restrict__ a,
                                                      const float* __restrict__ b, const float* __restrict__ w,
                                                      float* __restrict__ loss_rows, float* __restrict__ logits,
                                                      float* __restrict__ dq, float* __restrict__ da, float* __restrict__ db, int B,
                                                      int H) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B) return;
  const float *qi = q + (size_t)i * H, *ai = a + (size_t)i * H, *bi = b + (size_t)i * H;
  float s0 = 0.f, s1 = 0.f;
  for (int h = lane; h < H; h += 64) {
    s0 += qi[h] * ai[h];
    s1 += qi[h] * bi[h];
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  const float mx = fmaxf(s0, s1);
  const float lse = mx + __logf(__expf(s0 - mx) + __expf(s1 - mx));
  const float p0 = __expf(s0 - lse), p1 = __expf(s1 - lse);
  const float wi = (w ? w[i] : 1.f) / (float)B;
  const float c0 = (p0 - 1.f) * wi, c1 = p1 * wi;
  if (lane == 0) {
    loss_rows[i] = lse - s0;
    logits[2 * i] = s0;
    logits[2 * i + 1] = s1;
  }
  for (int h = lane; h < H; h += 64) {
    const float qv = qi[h];
    dq[(size_t)i * H + h] = c0 * ai[h] + c1 * bi[h];
    da[(size_t)i * H + h] = c0 * qv;
    db[(size_t)i * H + h] = c1 * qv;
  }
}

// one 256-thread workgroup per row: V ~ 30 k logits, three passes (max, sum-exp, gradient) from L2
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                 const float* __restrict__ row_scale, float* __restrict__ loss_rows,
                                                 uint16_t* __restrict__ dlogits, int V, int ld) {
  __shared__ float red[4];
  const int i = blockIdx.x, tid = threadIdx.x;
  const float* row = logits + (size_t)i * ld;
  float mx = -INFINITY;
  for (int j = tid * 4; j < V; j += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(row + j);
    mx = fmaxf(mx, v.x);
    if (j + 1 < V) mx = fmaxf(mx, v.y);
    if (j + 2 < V) mx = fmaxf(mx, v.z);
    if (j + 3 < V) mx = fmaxf(mx, v.w);
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int j = tid * 4; j < V; j += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(row + j);
    s += __expf(v.x - mx);
    if (j + 1 < V) s += __expf(v.y - mx);
    if (j + 2 < V) s += __expf(v.z - mx);
    if (j + 3 < V) s += __expf(v.w - mx);
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  s = (red[0] + red[1]) + (red[2] + red[3]);
  const float lse = mx + __logf(s);
  const int lab = labels[i];
  const float sc = row_scale[i];
  if (tid == 0) loss_rows[i] = lse - row[lab];
  uint16_t* drow = dlogits + (size_t)i * ld;
  for (int j = tid * 4; j < ld; j += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(row + j);
    const float x[4] = {v.x, v.y, v.z, v.w};
    float g[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = j + e;
      g[e] = col < V ? sc * (__expf(x[e] - lse) - (col == lab ? 1.f : 0.f)) : 0.f;
    }
    *reinterpret_cast<uint2*>(drow + j) = pack4(g);
  }
}

// the same with the row held in registers: 1024 threads x 8 float4 cover ld <= 32768 - one read of the fp32 logits instead of three
// (149 MB per 1216 x 30 592 rows: the kernel is bound by that traffic)
constexpr int CE1_THREADS = 1024, CE1_NV = 8;
__global__ __launch_bounds__(CE1_THREADS) void ce_row_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                             const float* __restrict__ row_scale, float* __restrict__ loss_rows,
                                                             uint16_t* __restrict__ dlogits, int V, int ld) {
  __shared__ float red[CE1_THREADS / 64];
  const int i = blockIdx.x, tid = threadIdx.x;
  const float* row = logits + (size_t)i * ld;
  float4 v[CE1_NV];
#pragma unroll
  for (int k = 0; k < CE1_NV; ++k) {
    const int j = (tid + k * CE1_THREADS) * 4;
    v[k] = j < ld ? *reinterpret_cast<const float4*>(row + j) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  }
  const int lab = labels[i];
  const float sc = row_scale[i];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < CE1_NV; ++k) {
    const int j = (tid + k * CE1_THREADS) * 4;
    if (j + 3 < V) mx = fmaxf(mx, fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(v[k].z, v[k].w)));
    else {
      if (j < V) mx = fmaxf(mx, v[k].x);
      if (j + 1 < V) mx = fmaxf(mx, v[k].y);
      if (j + 2 < V) mx = fmaxf(mx, v[k].z);
    }
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int w = 1; w < CE1_THREADS / 64; ++w) mx = fmaxf(mx, red[w]);
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < CE1_NV; ++k) {
    const int j = (tid + k * CE1_THREADS) * 4;
    if (j < V) s += __expf(v[k].x - mx);
    if (j + 1 < V) s += __expf(v[k].y - mx);
    if (j + 2 < V) s += __expf(v[k].z - mx);
    if (j + 3 < V) s += __expf(v[k].w - mx);
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  s = 0.f;
#pragma unroll
  for (int w = 0; w < CE1_THREADS / 64; ++w) s += red[w];
  const float lse = mx + __logf(s);
  uint16_t* drow = dlogits + (size_t)i * ld;
#pragma unroll
  for (int k = 0; k < CE1_NV; ++k) {
    const int j = (tid + k * CE1_THREADS) * 4;
    if (j >= ld) continue;
    const float x[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
    float g[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = j + e;
      g[e] = col < V ? sc * (__expf(x[e] - lse) - (col == lab ? 1.f : 0.f)) : 0.f;
      if (col == lab) loss_rows[i] = lse - x[e];
    }
    *reinterpret_cast<uint2*>(drow + j) = pack4(g);
  }
}

}  // namespace

extern "C" int cocodr_ce_fwd_bwd(const float* logits, const int32_t* labels, const float* row_scale, int n, int V, int ld,
                                 float* loss_rows, uint16_t* dlogits, cocodr_stream_t stream) {
  CK_ARG(logits && labels && row_scale && loss_rows && dlogits, "ce: null pointer");
  CK_ARG(n > 0 && V > 0 && ld >= V && ld % 4 == 0, "ce: bad shape n=%d V=%d ld=%d (ld %% 4 == 0)", n, V, ld);
  static const bool three_pass = getenv("COCODR_CE_THREE_PASS") != nullptr;  // A/B switch
  if (ld <= CE1_THREADS * CE1_NV * 4 && !three_pass)
    hipLaunchKernelGGL(ce_row_kernel, dim3(n), dim3(CE1_THREADS), 0, (hipStream_t)stream, logits, labels, row_scale, loss_rows, dlogits, V, ld);
  else
    hipLaunchKernelGGL(ce_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, logits, labels, row_scale, loss_rows, dlogits, V, ld);
  CK_LAUNCH("ce");
  return COCODR_OK;
}


extern "C" size_t cocodr_simce_workspace_floats(int M) { return (size_t)M * M + (size_t)M; }

extern "C" int cocodr_simce_fwd_bwd(const float* E, int M, int H, int world, int row0, int m_local, float* loss_rows, float* loss,
                                    float* dE_local, float* workspace, cocodr_stream_t stream) {
  CK_ARG(E && loss_rows && loss && dE_local && workspace, "simce: null pointer");
  CK_ARG(M >= 2 && M % 2 == 0 && H > 0 && world >= 1, "simce: bad shape M=%d H=%d world=%d (M must be even: span pairs)", M, H, world);
  CK_ARG(row0 >= 0 && m_local > 0 && row0 + m_local <= M, "simce: local rows [%d,%d) outside [0,%d)", row0, row0 + m_local, M);
  hipStream_t st = (hipStream_t)stream;
  float* S = workspace;
  float* lse = workspace + (size_t)M * M;
  const int nt = (M + TS - 1) / TS;
  hipLaunchKernelGGL(simce_scores_kernel, dim3(nt, nt), dim3(1024), 0, st, E, S, M, H);
  CK_LAUNCH("simce_scores");
  hipLaunchKernelGGL(simce_rowstats_kernel, dim3((M + 3) / 4), dim3(256), 0, st, S, lse, loss_rows, M, (float)world);
  CK_LAUNCH("simce_rowstats");
  hipLaunchKernelGGL(weighted_mean_kernel, dim3(1), dim3(256), 0, st, loss_rows, (const float*)nullptr, loss, M, 1.0f / (float)M);
  CK_LAUNCH("simce_mean");
  hipLaunchKernelGGL(simce_grad_kernel, dim3((H + TS - 1) / TS, (m_local + TS - 1) / TS), dim3(256), 0, st, E, S, lse, dE_local, M, H,
                     row0, m_local, (float)world / (float)M);
  CK_LAUNCH("simce_grad");
  return COCODR_OK;
}

extern "C" int cocodr_triplet_nll_fwd_bwd(const float* q, const float* a, const float* b, const float* weights, int B, int H,
                                          float* loss_rows, float* logits, float* loss, float* dq, float* da, float* db,
                                          cocodr_stream_t stream) {
  CK_ARG(q && a && b && loss_rows && logits && loss && dq && da && db, "triplet: null pointer");
  CK_ARG(B > 0 && H > 0, "triplet: bad shape");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(triplet_kernel, dim3((B + 3) / 4), dim3(256), 0, st, q, a, b, weights, loss_rows, logits, dq, da, db, B, H);
  CK_LAUNCH("triplet");
  hipLaunchKernelGGL(weighted_mean_kernel, dim3(1), dim3(256), 0, st, loss_rows, weights, loss, B, 1.0f / (float)B);
  CK_LAUNCH("triplet_mean");
  return COCODR_OK;
}

// ------------------------------------------------------------------ gram of a short, very wide fp32 matrix
// out = A A^T for A [G, D], G <= 64 rows, D ~ 1e7 columns: the per-group gradient matrix of iDRO
// (ANCE/model/dro_loss.py:236-238 `all_grads @ all_grads.T`).  One pass over HBM: 8 x 8 blocks of the output, the
// workgroups of a block pair split the columns, every thread keeps the 64 products of its columns in registers; per-workgroup
// partials are summed by a second kernel in a fixed order (deterministic, no atomics).  A block pair (i, j) streams rows 8i..
// and 8j.. once: G <= 8 reads A once, G = 16 twice - byte stream work, nowhere near a GEMM.
namespace {
constexpr int GRAM_B = 8, GRAM_WGS = 512;
__global__ __launch_bounds__(256) void gram_partial_kernel(const float* __restrict__ A, long long lda, int G, long long D,
                                                           float* __restrict__ partial, int nblk) {
  // blockIdx.y = block pair p -> (bi, bj), bi <= bj, enumerated row-major over the upper triangle
  int bi = 0, p = blockIdx.y;
  while (p >= nblk - bi) { p -= nblk - bi; ++bi; }
  const int bj = bi + p;
  float acc[GRAM_B][GRAM_B];
#pragma unroll
  for (int i = 0; i < GRAM_B; ++i)
#pragma unroll
    for (int j = 0; j < GRAM_B; ++j) acc[i][j] = 0.f;
  const float* Ai = A + (long long)bi * GRAM_B * lda;
  const float* Aj = A + (long long)bj * GRAM_B * lda;
  const int ri = min(GRAM_B, G - bi * GRAM_B), rj = min(GRAM_B, G - bj * GRAM_B);
  for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < D; c += (long long)gridDim.x * 256) {
    float x[GRAM_B], y[GRAM_B];
#pragma unroll
    for (int i = 0; i < GRAM_B; ++i) x[i] = i < ri ? Ai[i * lda + c] : 0.f;
    if (bi == bj) {
#pragma unroll
      for (int j = 0; j < GRAM_B; ++j) y[j] = x[j];
    } else {
#pragma unroll
      for (int j = 0; j < GRAM_B; ++j) y[j] = j < rj ? Aj[j * lda + c] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < GRAM_B; ++i)
#pragma unroll
      for (int j = 0; j < GRAM_B; ++j) acc[i][j] = fmaf(x[i], y[j], acc[i][j]);
  }
  __shared__ float red[4][GRAM_B * GRAM_B];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < GRAM_B; ++i)
#pragma unroll
    for (int j = 0; j < GRAM_B; ++j) {
      const float s = wave_sum(acc[i][j]);
      if (lane == 0) red[wid][i * GRAM_B + j] = s;
    }
  __syncthreads();
  if (threadIdx.x < GRAM_B * GRAM_B)
    partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (GRAM_B * GRAM_B) + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(64) void gram_finish_kernel(const float* __restrict__ partial, int nwg, int G, int nblk,
                                                         float* __restrict__ out) {
  int bi = 0, p = blockIdx.x;
  while (p >= nblk - bi) { p -= nblk - bi; ++bi; }
  const int bj = bi + p;
  const int e = threadIdx.x, i = bi * GRAM_B + e / GRAM_B, j = bj * GRAM_B + e % GRAM_B;
  float s = 0.f;
  for (int w = 0; w < nwg; ++w) s += partial[((size_t)blockIdx.x * nwg + w) * (GRAM_B * GRAM_B) + e];
  if (i < G && j < G) {
    out[(size_t)i * G + j] = s;
    if (bi != bj) out[(size_t)j * G + i] = s;
  }
}
// G in (16, 64]: one workgroup owns the WHOLE G x G partial of its column slabs.  A slab = 64 rows x 128 columns in LDS (rows
// >= G zero, row stride 132 floats: the 16 consecutive rows one 16-B read instruction touches fall into distinct bank groups);
// thread (ty, tx) of the 16 x 16 grid accumulates the 4 x 4 outputs of rows ty + 16 i x tx + 16 j over the slab's columns from 8
// float4 reads per 4 columns.  A is read exactly once.
constexpr int GS_COLS = 128, GS_LD = 132;
__global__ __launch_bounds__(256) void gram_slab_kernel(const float* __restrict__ A, long long lda, int G, long long D,
                                                        float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float slab[64 * GS_LD];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int tx = lane & 7, ty = lane >> 3;
  // every wave keeps its own 64 x 64 partial (8 x 8 outputs per lane: rows ty + 8 i x tx + 8 j) over a quarter of the slab's
  // column groups: 16 LDS reads per 128 packed FMAs.  Two partial sums per output (even / odd columns) so that the products run as
  // v_pk_fma_f32 (two FMAs per lane and instruction); with 4 x 4 outputs per lane the loop was bound by the LDS reads
  cocodr_f32x2 acc2[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc2[i][j] = cocodr_f32x2{0.f, 0.f};
  const long long nslab = (D + GS_COLS - 1) / GS_COLS;
  const bool vec = (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);  // 16-B loads when every row starts aligned
  constexpr int NV = 64 * GS_COLS / 4 / 256;  // float4 chunks per thread and slab
  // (fetching the next slab into registers under this slab's products needs 32 more VGPRs and drops the kernel to one wave per
  // SIMD: 5.0 -> 8.2 ms at 50 x 37.8 M; two resident workgroups per CU overlap each other's loads instead)
  for (long long sl = blockIdx.x; sl < nslab; sl += gridDim.x) {
    const long long c0 = sl * GS_COLS;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int e = threadIdx.x + k * 256, r = e / (GS_COLS / 4), c = (e % (GS_COLS / 4)) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < G) {
        const float* src = A + (long long)r * lda + c0 + c;
        if (vec && c0 + c + 3 < D) v = *reinterpret_cast<const float4*>(src);
        else {
          if (c0 + c < D) v.x = src[0];
          if (c0 + c + 1 < D) v.y = src[1];
          if (c0 + c + 2 < D) v.z = src[2];
          if (c0 + c + 3 < D) v.w = src[3];
        }
      }
      *reinterpret_cast<float4*>(&slab[r * GS_LD + c]) = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int c = wid * 4; c < GS_COLS; c += 16) {
      float4 x[8], y[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = *reinterpret_cast<const float4*>(&slab[(ty + 8 * i) * GS_LD + c]);
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = *reinterpret_cast<const float4*>(&slab[(tx + 8 * j) * GS_LD + c]);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc2[i][j] = __builtin_elementwise_fma(cocodr_f32x2{x[i].x, x[i].y}, cocodr_f32x2{y[j].x, y[j].y}, acc2[i][j]);
          acc2[i][j] = __builtin_elementwise_fma(cocodr_f32x2{x[i].z, x[i].w}, cocodr_f32x2{y[j].z, y[j].w}, acc2[i][j]);
        }
    }
  }
  // the four waves' partials, summed through LDS in wave order (deterministic)
  float* tile = slab;  // [64][64]
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wid == w) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int o = (ty + 8 * i) * 64 + tx + 8 * j;
          const float v = acc2[i][j][0] + acc2[i][j][1];
          tile[o] = w == 0 ? v : tile[o] + v;
        }
    }
  }
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * 64 * 64;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) out[e] = tile[e];
}
__global__ __launch_bounds__(256) void gram_slab_finish_kernel(const float* __restrict__ partial, int nwg, int G, float* __restrict__ out) {
  const int e = blockIdx.x * 256 + threadIdx.x;  // element of the 64 x 64 partial
  const int i = e >> 6, j = e & 63;
  if (i >= G || j >= G) return;
  float s = 0.f;
  for (int w = 0; w < nwg; ++w) s += partial[(size_t)w * 4096 + e];
  out[(size_t)i * G + j] = s;
}
int gram_slab_wgs(long long D) { return (int)std::max<long long>(1, std::min<long long>(1024, (D + GS_COLS * 4 - 1) / (GS_COLS * 4))); }
int gram_wgs(long long D) { return (int)std::max<long long>(1, std::min<long long>(GRAM_WGS, (D + 256 * 8 - 1) / (256 * 8))); }
}  // namespace

extern "C" size_t cocodr_gram_f32_workspace_floats(int G, long long D) {
  if (G <= 0 || D <= 0) return 0;
  if (G > 16) return (size_t)gram_slab_wgs(D) * 64 * 64;
  const int nblk = (G + GRAM_B - 1) / GRAM_B;
  return (size_t)(nblk * (nblk + 1) / 2) * gram_wgs(D) * GRAM_B * GRAM_B;
}
extern "C" int cocodr_gram_f32(const float* A, long long lda, int G, long long D, float* out, float* workspace, cocodr_stream_t stream) {
  CK_ARG(A && out && workspace, "gram: null pointer");
  CK_ARG(G > 0 && G <= 64 && D > 0 && lda >= D, "gram: bad shape G=%d D=%lld lda=%lld (G <= 64)", G, D, lda);
  hipStream_t st = (hipStream_t)stream;
  if (G > 16) {  // the slab form reads A once whatever G is; the register form below would re-read it per 8-row block pair
    const int nwg = gram_slab_wgs(D);
    hipLaunchKernelGGL(gram_slab_kernel, dim3(nwg), dim3(256), 0, st, A, lda, G, D, workspace);
    CK_LAUNCH("gram_slab");
    hipLaunchKernelGGL(gram_slab_finish_kernel, dim3(16), dim3(256), 0, st, workspace, nwg, G, out);
    CK_LAUNCH("gram_slab_finish");
    return COCODR_OK;
  }
  const int nblk = (G + GRAM_B - 1) / GRAM_B, pairs = nblk * (nblk + 1) / 2, nwg = gram_wgs(D);
  hipLaunchKernelGGL(gram_partial_kernel, dim3(nwg, pairs), dim3(256), 0, st, A, lda, G, D, workspace, nblk);
  CK_LAUNCH("gram_partial");
  hipLaunchKernelGGL(gram_finish_kernel, dim3(pairs), dim3(64), 0, st, workspace, nwg, G, nblk, out);
  CK_LAUNCH("gram_finish");
  return COCODR_OK;
}
