// Brute-force inner-product search  D, I = topk_k(Q P^T)  - the faiss.IndexFlatIP(dim).search of
// evaluate/evaluation/evaluate_beir.py:220-224 and ANCE/drivers/run_ann_data_gen.py:310-317,390.
//
// Scores are exact fp32 (v_mfma_f32_32x32x2_f32 is bit-identical to an fmaf chain), so top-k ids
// match an fp32 CPU search whenever the score gap exceeds fp32 round-off.  Selection is an exact,
// deterministic radix select per query row (order: score descending, position ascending on ties)
// followed by a bitonic sort of the k survivors in LDS.
//
// Round-1 structure: a chunk of query rows is scored into a [QC, Np] fp32 slab in the caller's
// workspace, then selected.  (The slab round trip is the next thing to remove - see DESIGN.md.)
#include <algorithm>

#include "common.h"
#include "prof.h"

namespace {

// ------------------------------------------------------------------ fp32 score GEMM  S = Q P^T
constexpr int SB = 128;   // tile
constexpr int SK = 16;    // contraction chunk
constexpr int SLD = 17;   // padded LDS leading dim

__global__ __launch_bounds__(256, 2) void score_gemm_kernel(const float* __restrict__ Q, const float* __restrict__ P,
                                                            float* __restrict__ S, int nq, int np, int H, long long lds_s) {
  __shared__ float Qs[2][SB * SLD], Ps[2][SB * SLD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
  const int ntn = (np + SB - 1) / SB;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int q0 = (tile / ntn) * SB, p0 = (tile % ntn) * SB;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // each thread stages 2 float4 per operand tile: row = (tid >> 2) + 64*i, k = (tid & 3) * 4
  float4 rq[2], rp[2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (tid >> 2) + 64 * i, k = k0 + ((tid & 3) << 2);
      rq[i] = (q0 + row < nq && k < H) ? *reinterpret_cast<const float4*>(Q + (size_t)(q0 + row) * H + k) : make_float4(0, 0, 0, 0);
      rp[i] = (p0 + row < np && k < H) ? *reinterpret_cast<const float4*>(P + (size_t)(p0 + row) * H + k) : make_float4(0, 0, 0, 0);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (tid >> 2) + 64 * i, k = (tid & 3) << 2;
      float* dq = &Qs[buf][row * SLD + k];
      float* dp = &Ps[buf][row * SLD + k];
      dq[0] = rq[i].x; dq[1] = rq[i].y; dq[2] = rq[i].z; dq[3] = rq[i].w;
      dp[0] = rp[i].x; dp[1] = rp[i].y; dp[2] = rp[i].z; dp[3] = rp[i].w;
    }
  };
  const int nt = (H + SK - 1) / SK;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) gload((t + 1) * SK);
#pragma unroll
    for (int kk = 0; kk < SK; kk += 2) {
      float fa[2], fb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = Qs[buf][(wm * 64 + a * 32 + (lane & 31)) * SLD + kk + (lane >> 5)];
#pragma unroll
      for (int b = 0; b < 2; ++b) fb[b] = Ps[buf][(wn * 64 + b * 32 + (lane & 31)) * SLD + kk + (lane >> 5)];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    if (t + 1 < nt) sstore(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = q0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int pj = p0 + wn * 64 + b * 32 + (lane & 31);
        if (qi < nq && pj < np) S[(size_t)qi * lds_s + pj] = acc[a][b][r];
      }
}

// ------------------------------------------------------------------ exact top-k per row
constexpr int TOPK_THREADS = 512;
constexpr int NREP = 4;        // replicated LDS histograms (lane & 3) to thin same-address atomics
constexpr int NBIN = 2048;
constexpr int KMAX = 2048;
constexpr int NCAND = 2048;    // LDS candidate list of the compacted path

// order-preserving map: smaller key <=> larger float  (NaN sorts last)
__device__ __forceinline__ uint32_t desc_key(float f) {
  if (f != f) return 0xffffffffu;  // every NaN behind -inf (key_to_float gives a NaN back)
  uint32_t u = __float_as_uint(f);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending in f
  return ~u;
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
  uint32_t u = ~k;
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __uint_as_float(u);
}

struct SelState {
  uint32_t prefix;  // selected high bits so far
  int need;         // how many still to take among the elements matching the prefix
  int count_eq;
};

// One radix pass over the elements `scan` enumerates: histogram digit(v) of the values val(key, idx, v) accepts whose
// high bits match the prefix, pick the smallest digit whose cumulative count reaches st.need.
template <typename Scan, typename Val>
__device__ __forceinline__ void radix_pass(Scan scan, Val val, int shift, int bits, uint32_t himask, SelState& st, uint32_t* hist,
                                           int* sh) {
  const int tid = threadIdx.x;
  const int nb = 1 << bits;
  for (int i = tid; i < NREP * NBIN; i += TOPK_THREADS) hist[i] = 0;
  __syncthreads();
  uint32_t* my = hist + (tid & (NREP - 1)) * NBIN;
  const uint32_t prefix = st.prefix;
  scan([&](uint32_t key, uint32_t idx) {
    uint32_t v;
    if (val(key, idx, v) && (v & himask) == prefix) atomicAdd(&my[(v >> shift) & (nb - 1)], 1u);
  });
  __syncthreads();
  for (int b = tid; b < nb; b += TOPK_THREADS) {
    uint32_t t = 0;
#pragma unroll
    for (int r = 0; r < NREP; ++r) t += hist[r * NBIN + b];
    hist[b] = t;
  }
  __syncthreads();
  if (tid == 0) {
    int cum = 0, d = 0;
    for (; d < nb - 1; ++d) {
      if (cum + (int)hist[d] >= st.need) break;
      cum += hist[d];
    }
    sh[0] = d;
    sh[1] = st.need - cum;
    sh[2] = hist[d];
  }
  __syncthreads();
  st.prefix |= ((uint32_t)sh[0]) << shift;
  st.need = sh[1];
  st.count_eq = sh[2];
  __syncthreads();
}

// Exact selection of the k best of one score row.  Pass 1 histograms the top 11 key bits of the whole row (16-byte
// loads).  When the bin that holds the k-th score is small (the usual case: k << Np and scores spread over a few
// exponents), ONE more pass over the row moves the sure winners to the output buffer and that bin's members to an LDS
// candidate list, and the remaining radix / tie-break passes run on the list: 2 trips over the row instead of 4.
// Otherwise all passes stream the row, as before.  Result and order (score descending, position ascending on ties) are
// identical on both paths.
__global__ __launch_bounds__(TOPK_THREADS) void topk_kernel(const float* __restrict__ S, long long lds_s, int np, int k,
                                                            long long id_offset, float* __restrict__ D, long long* __restrict__ I) {
  __shared__ uint32_t hist[NREP * NBIN];
  __shared__ unsigned long long buf[KMAX];   // (key << 32) | idx : ascending = score desc, idx asc
  __shared__ unsigned long long cand[NCAND];
  __shared__ int sh[6];
  const int tid = threadIdx.x;
  const float* row = S + (size_t)blockIdx.x * lds_s;
  float* Drow = D + (size_t)blockIdx.x * k;
  long long* Irow = I + (size_t)blockIdx.x * k;
  const int kk = min(k, np);
  int kpad = 1;
  while (kpad < kk) kpad <<= 1;
  for (int i = tid; i < kpad; i += TOPK_THREADS) buf[i] = ~0ull;
  if (tid == 0) { sh[3] = 0; sh[4] = 0; }

  auto scan_row = [&](auto body) {  // rows start 16-byte aligned (workspace leading dimension is a multiple of 4)
    const int n4 = np >> 2;
    const float4* r4 = reinterpret_cast<const float4*>(row);
    for (int i = tid; i < n4; i += TOPK_THREADS) {
      const float4 v = r4[i];
      body(desc_key(v.x), (uint32_t)(4 * i));
      body(desc_key(v.y), (uint32_t)(4 * i + 1));
      body(desc_key(v.z), (uint32_t)(4 * i + 2));
      body(desc_key(v.w), (uint32_t)(4 * i + 3));
    }
    for (int i = (n4 << 2) + tid; i < np; i += TOPK_THREADS) body(desc_key(row[i]), (uint32_t)i);
  };
  auto by_key = [](uint32_t key, uint32_t, uint32_t& v) { v = key; return true; };

  uint32_t thr = 0xffffffffu, ithr = 0xffffffffu;
  bool compacted = false;
  int ncand = 0;
  if (kk < np) {
    SelState st{0u, kk, 0};
    radix_pass(scan_row, by_key, 21, 11, 0x00000000u, st, hist, sh);
    compacted = st.count_eq <= NCAND;  // workgroup-uniform (read from LDS)
    if (compacted) {
      const uint32_t bin = st.prefix >> 21;
      scan_row([&](uint32_t key, uint32_t idx) {
        const uint32_t top = key >> 21;
        if (top < bin) buf[atomicAdd(&sh[3], 1)] = ((unsigned long long)key << 32) | idx;
        else if (top == bin) cand[atomicAdd(&sh[4], 1)] = ((unsigned long long)key << 32) | idx;
      });
      __syncthreads();
      ncand = sh[4];
    }
    auto scan_cand = [&](auto body) {
      for (int c = tid; c < ncand; c += TOPK_THREADS) {
        const unsigned long long e = cand[c];
        body((uint32_t)(e >> 32), (uint32_t)e);
      }
    };
    auto rest = [&](auto scan) {
      radix_pass(scan, by_key, 10, 11, 0xffe00000u, st, hist, sh);
      radix_pass(scan, by_key, 0, 10, 0xfffffc00u, st, hist, sh);
      thr = st.prefix;
      if (st.count_eq > st.need) {  // exact score ties straddle the cut: keep the lowest positions
        SelState si{0u, st.need, 0};
        const uint32_t t = thr;
        auto by_idx = [t](uint32_t key, uint32_t idx, uint32_t& v) { v = idx; return key == t; };
        radix_pass(scan, by_idx, 21, 11, 0x00000000u, si, hist, sh);
        radix_pass(scan, by_idx, 10, 11, 0xffe00000u, si, hist, sh);
        radix_pass(scan, by_idx, 0, 10, 0xfffffc00u, si, hist, sh);
        ithr = si.prefix;
      }
    };
    if (compacted) rest(scan_cand);
    else rest(scan_row);
  }
  __syncthreads();
  {
    auto take = [&](uint32_t key, uint32_t idx) {
      if (key < thr || (key == thr && idx <= ithr)) {
        const int pos = atomicAdd(&sh[3], 1);
        if (pos < kpad) buf[pos] = ((unsigned long long)key << 32) | idx;
      }
    };
    if (compacted) {
      for (int c = tid; c < ncand; c += TOPK_THREADS) {
        const unsigned long long e = cand[c];
        take((uint32_t)(e >> 32), (uint32_t)e);
      }
    } else {
      scan_row(take);
    }
  }
  __syncthreads();
  // bitonic sort ascending over kpad entries
  for (int size = 2; size <= kpad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (kpad >> 1); t += TOPK_THREADS) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const unsigned long long a = buf[lo], b = buf[hi];
        if ((a > b) == up) { buf[lo] = b; buf[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < k; i += TOPK_THREADS) {
    if (i < kk) {
      const unsigned long long e = buf[i];
      Drow[i] = key_to_float((uint32_t)(e >> 32));
      Irow[i] = (long long)(uint32_t)e + id_offset;
    } else {
      Drow[i] = -INFINITY;
      Irow[i] = -1;
    }
  }
}

int query_chunk(int nq, int np) {
  long long qc = (1ll << 29) / std::max(np, 1);  // <= 2 GiB slab
  qc = std::max(128ll, std::min(qc, 8192ll));
  qc = (qc / 128) * 128;
  return (int)std::min<long long>(qc, ((long long)nq + 127) / 128 * 128);
}

}  // namespace

extern "C" size_t cocodr_score_topk_workspace_bytes(int Nq, int Np, int k) {
  (void)k;
  if (Nq <= 0 || Np <= 0) return 0;
  const size_t ld = ((size_t)Np + 3) / 4 * 4;
  return (size_t)query_chunk(Nq, Np) * ld * sizeof(float);
}

extern "C" int cocodr_score_topk(const float* Q, const float* P, int Nq, int Np, int H, int k, long long id_offset, float* D,
                                 long long* I, void* workspace, size_t workspace_bytes, cocodr_stream_t stream) {
  CK_ARG(Q && P && D && I && workspace, "score_topk: null pointer");
  CK_ARG(Nq > 0 && Np > 0 && H > 0 && H % 4 == 0, "score_topk: bad shape Nq=%d Np=%d H=%d (H %% 4 == 0)", Nq, Np, H);
  CK_ARG(k > 0 && k <= KMAX, "score_topk: k=%d must be in [1,%d]", k, KMAX);
  CK_ARG((((uintptr_t)Q | (uintptr_t)P) & 15) == 0, "score_topk: Q and P must be 16-byte aligned");
  if (workspace_bytes < cocodr_score_topk_workspace_bytes(Nq, Np, k)) {
    cocodr_set_error("score_topk: workspace %zu B < required %zu B", workspace_bytes, cocodr_score_topk_workspace_bytes(Nq, Np, k));
    return COCODR_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int QC = query_chunk(Nq, Np);
  const long long ld = ((long long)Np + 3) / 4 * 4;
  float* S = reinterpret_cast<float*>(workspace);
  for (int q0 = 0; q0 < Nq; q0 += QC) {
    const int nq = std::min(QC, Nq - q0);
    const int ntm = (nq + SB - 1) / SB, ntn = (Np + SB - 1) / SB;
    {
      ProfScope prof(PROF_SCORE, st, 2.0 * nq * (double)Np * H);
      hipLaunchKernelGGL(score_gemm_kernel, dim3(ntm * ntn), dim3(256), 0, st, Q + (size_t)q0 * H, P, S, nq, Np, H, ld);
    }
    CK_LAUNCH("score_gemm");
    hipLaunchKernelGGL(topk_kernel, dim3(nq), dim3(TOPK_THREADS), 0, st, S, ld, Np, k, id_offset, D + (size_t)q0 * k, I + (size_t)q0 * k);
    CK_LAUNCH("topk");
  }
  return COCODR_OK;
}
