// Brute-force inner-product search  D, I = topk_k(Q P^T)  - the faiss.IndexFlatIP(dim).search of
// evaluate/evaluation/evaluate_beir.py:220-224 and ANCE/drivers/run_ann_data_gen.py:310-317,390.
//
// Scores: fp32-accurate on the 16-bit matrix pipe (split precision, the default; further down) or exact fp32
// (v_mfma_f32_32x32x2_f32 is bit-identical to an fmaf chain; mode 1), so top-k ids match an fp32 CPU search whenever the score gap
// exceeds fp32 round-off.  Selection is exact and deterministic (order: score descending, position ascending on ties).
//
// Two routes to the k best of a row:
//  * exhaustive (small searches, mode 1, and the rows the other route hands back): a chunk of query rows is scored into a
//    [QC, Np] fp32 slab in the caller's workspace, then a radix select per row + bitonic sort of the k survivors.  The workspace
//    holds TWO slabs: the selection of chunk c runs on a side stream while the score GEMM of chunk c + 1 runs on the caller's
//    (measured in round 5: ~1 % - a CU holds a GEMM workgroup or selection workgroups, not both).
//  * filtered (round 5; >= 100 Mi scores over >= 32 768 passages; score_filter.h, filtered_search below): per-row thresholds from
//    a strided passage sample, the score GEMM's epilogue keeps only the scores at or above them, selection from those candidates.
//    No slab.  Same D and I, tie for tie.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>

#include "common.h"
#include "prof.h"
#include "score_filter.h"

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
    // the operands of the next-but-one k-pair are requested in front of the MFMAs of this one (two register sets): left
    // to itself hipcc issues them one MFMA (64 cycles) ahead of their use, less than the LDS latency
    float fa[2][2][2], fb[2][2][2];  // [set][k-pair within the set][fragment]
    auto frags = [&](int set, int kk) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int a = 0; a < 2; ++a) fa[set][j][a] = Qs[buf][(wm * 64 + a * 32 + (lane & 31)) * SLD + kk + 2 * j + (lane >> 5)];
#pragma unroll
        for (int b = 0; b < 2; ++b) fb[set][j][b] = Ps[buf][(wn * 64 + b * 32 + (lane & 31)) * SLD + kk + 2 * j + (lane >> 5)];
      }
    };
    frags(0, 0);
#pragma unroll
    for (int kk = 0; kk < SK; kk += 4) {
      const int set = (kk >> 2) & 1;
      if (kk + 4 < SK) frags(set ^ 1, kk + 4);
      __builtin_amdgcn_sched_barrier(0);  // keep the requests in front of this set's MFMAs
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][j][a], fb[set][j][b], acc[a][b], 0, 0, 0);
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
constexpr int NCAND = 3072;    // LDS candidate list of the compacted paths (72 KiB of LDS per workgroup in all: two per CU)

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
// THREADS: the workgroup's size; REP: histogram replicas (hist holds REP * NBIN words)
template <int THREADS = TOPK_THREADS, int REP = NREP, typename Scan, typename Val>
__device__ __forceinline__ void radix_pass(Scan scan, Val val, int shift, int bits, uint32_t himask, SelState& st, uint32_t* hist,
                                           int* sh) {
  constexpr int TOPK_THREADS = THREADS, NREP = REP;  // (shadow the file-level defaults inside this function)
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
  // the first digit whose cumulative count reaches st.need, by a block-wide prefix sum over the bins (each thread owns
  // nb / TOPK_THREADS consecutive bins).  One thread walking the 2048 bins - a chain of dependent LDS reads, ~55 us - was
  // most of the selection's time: every row pays three to six of these passes.
  {
    constexpr int CMAX = NBIN / TOPK_THREADS;
    const int C = nb / TOPK_THREADS;  // nb is 2048 or 1024
    int c[CMAX], local = 0;
#pragma unroll
    for (int j = 0; j < CMAX; ++j) {
      c[j] = j < C ? (int)hist[tid * C + j] : 0;
      local += c[j];
    }
    int incl = local;
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    int* wsum = sh + 8;  // TOPK_THREADS / 64 wave totals
    if (lane == 63) wsum[wv] = incl;
    if (tid == 0) sh[0] = -1;
    __syncthreads();
    int before = incl - local, total = 0;
#pragma unroll
    for (int w = 0; w < TOPK_THREADS / 64; ++w) {
      const int t = wsum[w];
      if (w < wv) before += t;
      total += t;
    }
    if (before < st.need && st.need <= before + local) {  // exactly one thread when total >= need
#pragma unroll
      for (int j = 0; j < CMAX; ++j) {
        if (j < C && before < st.need && st.need <= before + c[j]) { sh[0] = tid * C + j; sh[1] = st.need - before; sh[2] = c[j]; }
        before += c[j];
      }
    }
    if (total < st.need && tid == TOPK_THREADS - 1) {  // (cannot happen for a consistent need; mirrors the serial walk's last bin)
      const int last = (int)hist[nb - 1];
      sh[0] = nb - 1; sh[1] = st.need - (total - last); sh[2] = last;
    }
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
// unscale (or NULL): two ints, the binary exponents the split-precision path scaled Q and P by - D is multiplied by
// 2^-(unscale[0] + unscale[1]) on the way out (exact; ordering and ties are those of the scaled scores)
// row_map / n_dev (or NULL): slab row b holds the scores of output row row_map[b + n_base], and only while b + n_base < *n_dev
// (the exhaustive pass over the rows the filtered search hands back; the count exists on the device alone)
__global__ __launch_bounds__(TOPK_THREADS) void topk_kernel(const float* __restrict__ S, long long lds_s, int np, int k,
                                                            long long id_offset, float* __restrict__ D, long long* __restrict__ I,
                                                            const int* __restrict__ unscale, const int* __restrict__ row_map = nullptr,
                                                            const int* __restrict__ n_dev = nullptr, int n_base = 0) {
  __shared__ uint32_t hist[NREP * NBIN];
  __shared__ unsigned long long buf[KMAX];   // (key << 32) | idx : ascending = score desc, idx asc
  __shared__ unsigned long long cand[NCAND];
  __shared__ int sh[8 + TOPK_THREADS / 64];  // [0..2] radix-pass result, [3], [4] list counters, [8..] wave totals of its prefix sum
  const int tid = threadIdx.x;
  if (n_dev != nullptr && (int)blockIdx.x + n_base >= *n_dev) return;  // workgroup-uniform
  const size_t orow = row_map ? (size_t)row_map[blockIdx.x + n_base] : (size_t)blockIdx.x;
  const float* row = S + (size_t)blockIdx.x * lds_s;
  float* Drow = D + orow * k;
  long long* Irow = I + orow * k;
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
  bool compacted = false, sampled = false;
  int ncand = 0;
  // Sampled cut (k << Np): the r-th best of a 1/16 sample of the row, r a few standard deviations past the sample's share
  // of the top k, is with near certainty a score BELOW the k-th best of the row.  ONE pass over the row then moves
  // everything at or above it to the LDS candidate list; if the list holds at least k entries (so the guess was indeed
  // below the cut) and did not overflow, the exact selection runs on the list alone: ~1.1 trips over the row instead
  // of 2.  Any other outcome - an adversarial order, many ties - falls through to the full path below; the result is
  // the same either way.
  if (kk < np && np >= 32 * kk && np >= 4096) {
    const int blk = (np / 256) & ~3, stride = (np / 16) & ~3;  // 16 blocks of blk elements, 16-byte aligned
    auto scan_sample = [&](auto body) {
      const int b4 = blk >> 2;
      for (int i = tid; i < 16 * b4; i += TOPK_THREADS) {
        const int b = i / b4, j = i - b * b4;
        const int e0 = b * stride + 4 * j;
        const float4 v = *reinterpret_cast<const float4*>(row + e0);
        body(desc_key(v.x), (uint32_t)e0);
        body(desc_key(v.y), (uint32_t)(e0 + 1));
        body(desc_key(v.z), (uint32_t)(e0 + 2));
        body(desc_key(v.w), (uint32_t)(e0 + 3));
      }
    };
    const float mu = (float)kk * (16.0f * blk) / (float)np;
    const int r = min(16 * blk, (int)(mu + 4.0f * sqrtf(mu) + 8.0f));
    SelState ss{0u, r, 0};
    radix_pass(scan_sample, by_key, 21, 11, 0x00000000u, ss, hist, sh);
    radix_pass(scan_sample, by_key, 10, 11, 0xffe00000u, ss, hist, sh);
    radix_pass(scan_sample, by_key, 0, 10, 0xfffffc00u, ss, hist, sh);
    const uint32_t cut = ss.prefix;  // key of the r-th best sample score
    scan_row([&](uint32_t key, uint32_t idx) {
      if (key <= cut) {
        const int pos = atomicAdd(&sh[4], 1);
        if (pos < NCAND) cand[pos] = ((unsigned long long)key << 32) | idx;
      }
    });
    __syncthreads();
    const int total = sh[4];
    __syncthreads();
    if (total >= kk && total <= NCAND) {
      sampled = compacted = true;
      ncand = total;
    } else if (tid == 0) {
      sh[4] = 0;
    }
    __syncthreads();
  }
  if (kk < np) {
    SelState st{0u, kk, 0};
    if (!sampled) {
    radix_pass(scan_row, by_key, 21, 11, 0x00000000u, st, hist, sh);
    compacted = st.count_eq <= NCAND;  // workgroup-uniform (read from LDS)
    }
    if (compacted && !sampled) {
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
    if (sampled) radix_pass(scan_cand, by_key, 21, 11, 0x00000000u, st, hist, sh);  // the list holds the whole top k
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
  const float out_mul = unscale ? ldexpf(1.0f, -(unscale[0] + unscale[1])) : 1.0f;
  for (int i = tid; i < k; i += TOPK_THREADS) {
    if (i < kk) {
      const unsigned long long e = buf[i];
      Drow[i] = key_to_float((uint32_t)(e >> 32)) * out_mul;
      Irow[i] = (long long)(uint32_t)e + id_offset;
    } else {
      Drow[i] = -INFINITY;
      Irow[i] = -1;
    }
  }
}

// Bitonic sort (ascending) of a[0 .. n), n a power of two, in LDS by THREADS threads; ends behind a barrier.  A round trip through
// LDS carries up to THREE consecutive strides of a merge: a thread loads the 8 elements that differ in those three index bits, runs
// the three compare-exchange stages on registers and stores them back - 22 round trips instead of 55 single stages for n = 1024
// (the plain network is bound by its LDS traffic: every stage reads and writes every key).
template <int NST>
__device__ __forceinline__ void bitonic_round(unsigned long long* a, int g, int low, int size) {
  constexpr int E = 1 << NST;
  const int lo_bits = g & (low - 1);
  const int base = ((g - lo_bits) << NST) | lo_bits;
  const bool up = (base & size) == 0;
  unsigned long long v[E];
#pragma unroll
  for (int j = 0; j < E; ++j) v[j] = a[base + j * low];
#pragma unroll
  for (int st = NST - 1; st >= 0; --st)
#pragma unroll
    for (int j = 0; j < E; ++j)
      if ((j & (1 << st)) == 0) {
        const unsigned long long x = v[j], y = v[j | (1 << st)];
        const bool sw = (x > y) == up;
        v[j] = sw ? y : x;
        v[j | (1 << st)] = sw ? x : y;
      }
#pragma unroll
  for (int j = 0; j < E; ++j) a[base + j * low] = v[j];
}
template <int THREADS>
__device__ __forceinline__ void bitonic_sort_lds(unsigned long long* a, int n) {
  const int tid = threadIdx.x;
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0;) {
      const int nst = stride >= 4 ? 3 : (stride == 2 ? 2 : 1);
      const int low = stride >> (nst - 1);
      for (int g = tid; g < (n >> nst); g += THREADS) {
        if (nst == 3) bitonic_round<3>(a, g, low, size);
        else if (nst == 2) bitonic_round<2>(a, g, low, size);
        else bitonic_round<1>(a, g, low, size);
      }
      __syncthreads();
      stride >>= nst;
    }
  }
}

// ------------------------------------------------------------------ filtered search: selection from the GEMM's candidate blocks
// (score_filter.h).  One workgroup per query row: the row's hits of all column tiles are gathered into LDS as (key << 32) | column,
// sorted (bitonic, ascending = score descending, position ascending on ties - the order of topk_kernel) and the first k leave.
// A row whose blocks hold fewer than k hits (threshold too high), more than FCAP, or an overflowed block is not answered here:
// its index goes to fb_rows[atomicAdd(fb_count)] and the exhaustive pass (plain scores + topk_kernel) answers it.
// LDS: the list + ONE histogram (8 KiB) - 39 KiB for the 3 968-entry build, so four rows are in flight per CU: the kernel is a chain
// of short barrier-separated steps (three radix passes, 55 bitonic stages over 1 024 winners), bound by their latency, not by work.
constexpr int FCAP = 8192, FCAP_SMALL = 3968;  // candidates of a row the selection can hold (the plan picks the build)
constexpr int FSEL_THREADS = 256;
template <int LCAP>
__global__ __launch_bounds__(FSEL_THREADS) void filter_select_kernel(const uint2* __restrict__ cand, int ntn, int capt, int k,
                                                                     long long id_offset, float* __restrict__ D, long long* __restrict__ I,
                                                                     const int* __restrict__ unscale, int* __restrict__ fb_count,
                                                                     int* __restrict__ fb_rows) {
  __shared__ uint32_t hist[NBIN];
  __shared__ unsigned long long list[LCAP];   // (key << 32) | column : ascending = score descending, position ascending
  __shared__ int sh[8 + FSEL_THREADS / 64];   // [0..2] radix-pass result, [3] winners taken, [6] overflow, [7] hits (gather cursor)
  const int tid = threadIdx.x, row = blockIdx.x;
  const uint2* blocks = cand + (size_t)row * ntn * capt;
  if (tid == 0) { sh[3] = 0; sh[6] = 0; sh[7] = 0; }
  __syncthreads();
  {
    // ONE trip over the row's blocks: a block's first 64-byte line is its header and its first seven entries (the usual block holds
    // fewer), fetched as four independent 16-byte loads; entries go to the list at once (dropped, not written, past its end - the
    // row is handed back in that case anyway)
    int over = 0;
#pragma unroll 2
    for (int t = tid; t < ntn; t += FSEL_THREADS) {
      const uint4* b4 = reinterpret_cast<const uint4*>(blocks + (size_t)t * capt);  // (capt % 8 == 0: blocks are 64-byte aligned)
      const uint4 q0 = b4[0], q1 = b4[1], q2 = b4[2], q3 = b4[3];
      const int c = (int)q0.x;
      if (c > 0) {
        over |= c > capt - 1;
        const int at = atomicAdd(&sh[7], c);
        const uint32_t ev[7] = {q0.z, q1.x, q1.z, q2.x, q2.z, q3.x, q3.z}, ei[7] = {q0.w, q1.y, q1.w, q2.y, q2.w, q3.y, q3.w};
#pragma unroll
        for (int i = 0; i < 7; ++i)
          if (i < c && at + i < LCAP) list[at + i] = ((unsigned long long)desc_key(__uint_as_float(ev[i])) << 32) | ei[i];
        const uint2* b = blocks + (size_t)t * capt;
        for (int i = 7; i < min(c, capt - 1); ++i) {
          const uint2 e = b[1 + i];
          if (at + i < LCAP) list[at + i] = ((unsigned long long)desc_key(__uint_as_float(e.x)) << 32) | e.y;
        }
      }
    }
    if (over) sh[6] = 1;
  }
  __syncthreads();
  const int n = sh[7];
  if (sh[6] != 0 || n < k || n > LCAP) {  // workgroup-uniform: not answerable from the candidates
    if (tid == 0) fb_rows[atomicAdd(fb_count, 1)] = row;
    return;
  }
  auto scan = [&](auto body) {
    for (int c = tid; c < n; c += FSEL_THREADS) {
      const unsigned long long e = list[c];
      body((uint32_t)(e >> 32), (uint32_t)e);
    }
  };
  uint32_t thr = 0xffffffffu, ithr = 0xffffffffu;
  if (k < n) {  // the exact cut of topk_kernel, on the list
    auto by_key = [](uint32_t key, uint32_t, uint32_t& v) { v = key; return true; };
    SelState st{0u, k, 0};
    radix_pass<FSEL_THREADS, 1>(scan, by_key, 21, 11, 0x00000000u, st, hist, sh);
    radix_pass<FSEL_THREADS, 1>(scan, by_key, 10, 11, 0xffe00000u, st, hist, sh);
    radix_pass<FSEL_THREADS, 1>(scan, by_key, 0, 10, 0xfffffc00u, st, hist, sh);
    thr = st.prefix;
    if (st.count_eq > st.need) {  // exact score ties straddle the cut: keep the lowest positions
      SelState si{0u, st.need, 0};
      const uint32_t t = thr;
      auto by_idx = [t](uint32_t key, uint32_t idx, uint32_t& v) { v = idx; return key == t; };
      radix_pass<FSEL_THREADS, 1>(scan, by_idx, 21, 11, 0x00000000u, si, hist, sh);
      radix_pass<FSEL_THREADS, 1>(scan, by_idx, 10, 11, 0xffe00000u, si, hist, sh);
      radix_pass<FSEL_THREADS, 1>(scan, by_idx, 0, 10, 0xfffffc00u, si, hist, sh);
      ithr = si.prefix;
    }
  }
  int kpad = 1;
  while (kpad < k) kpad <<= 1;
  // the k winners move to the front of the list: every thread holds its entries in registers across the barrier, so no slot is
  // overwritten before it was read
  constexpr int PER = (LCAP + FSEL_THREADS - 1) / FSEL_THREADS;
  unsigned long long mine[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = tid + i * FSEL_THREADS;
    mine[i] = c < n ? list[c] : ~0ull;  // (n <= LCAP)
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const uint32_t key = (uint32_t)(mine[i] >> 32), idx = (uint32_t)mine[i];
    if (tid + i * FSEL_THREADS < n && (key < thr || (key == thr && idx <= ithr))) list[atomicAdd(&sh[3], 1)] = mine[i];
  }
  __syncthreads();
  for (int i = k + tid; i < kpad; i += FSEL_THREADS) list[i] = ~0ull;  // (kpad <= KMAX <= LCAP)
  __syncthreads();
  // (measured and not kept: the network on registers + wave shuffles - element r * 256 + tid in register r, only strides 64 / 128
  //  through the list - 374 us for 10 000 rows against 245 us for the plain LDS network: ds_bpermute costs what a 64-bit LDS access
  //  costs, and four rows per CU hide the barriers already)
  bitonic_sort_lds<FSEL_THREADS>(list, kpad);
  const float out_mul = unscale ? ldexpf(1.0f, -(unscale[0] + unscale[1])) : 1.0f;
  float* Drow = D + (size_t)row * k;
  long long* Irow = I + (size_t)row * k;
  for (int i = tid; i < k; i += FSEL_THREADS) {
    const unsigned long long e = list[i];
    Drow[i] = key_to_float((uint32_t)(e >> 32)) * out_mul;
    Irow[i] = (long long)(uint32_t)e + id_offset;
  }
}

// thr[row] = the j-th best of the row's n scores (the filter's threshold): three radix passes over the row, no list, no sort
__global__ __launch_bounds__(TOPK_THREADS) void row_rank_kernel(const float* __restrict__ S, long long lds_s, int n, int j,
                                                                float* __restrict__ thr) {
  __shared__ uint32_t hist[NREP * NBIN];
  __shared__ int sh[8 + TOPK_THREADS / 64];
  const int tid = threadIdx.x;
  const float* row = S + (size_t)blockIdx.x * lds_s;
  auto scan = [&](auto body) {  // (rows start 16-byte aligned, n % 4 == 0)
    const float4* r4 = reinterpret_cast<const float4*>(row);
    for (int i = tid; i < (n >> 2); i += TOPK_THREADS) {
      const float4 v = r4[i];
      body(desc_key(v.x), 0u); body(desc_key(v.y), 0u); body(desc_key(v.z), 0u); body(desc_key(v.w), 0u);
    }
  };
  auto by_key = [](uint32_t key, uint32_t, uint32_t& v) { v = key; return true; };
  SelState st{0u, j, 0};
  radix_pass(scan, by_key, 21, 11, 0x00000000u, st, hist, sh);
  radix_pass(scan, by_key, 10, 11, 0xffe00000u, st, hist, sh);
  radix_pass(scan, by_key, 0, 10, 0xfffffc00u, st, hist, sh);
  if (tid == 0) thr[blockIdx.x] = key_to_float(st.prefix);
}

// rows i * stride of X (row_halfs 16-bit elements each, a multiple of 8) -> row i of Y: the strided passage sample the
// thresholds come from.  idx != NULL: row idx[i + i_base] instead, for i + i_base < *n_dev (the fallback rows' operand copy).
__global__ __launch_bounds__(256) void copy_rows_kernel(const _Float16* __restrict__ X, _Float16* __restrict__ Y, int row_halfs,
                                                        long long stride, const int* __restrict__ idx, const int* __restrict__ n_dev,
                                                        int i_base) {
  const int i = blockIdx.x;
  long long src = (long long)i * stride;
  if (idx != nullptr) {
    if (i + i_base >= *n_dev) return;
    src = idx[i + i_base];
  }
  const uint4* x = reinterpret_cast<const uint4*>(X + (size_t)src * row_halfs);
  uint4* y = reinterpret_cast<uint4*>(Y + (size_t)i * row_halfs);
  for (int c = threadIdx.x; c < row_halfs / 8; c += 256) y[c] = x[c];
}

// ------------------------------------------------------------------ split-precision scores on the 16-bit matrix pipe
// x 2^s = xh + xl + r with xh, xl IEEE half (11 significant bits each) and |r| <= 2^-22 |x|: the three partial products
// ql.ph + qh.pl + qh.ph, every one exact in fp32 and accumulated in fp32 by v_mfma_f32_32x32x16_f16 in that order (small
// terms first), reproduce q.p to better than a sequential fp32 dot product does - measured max error 0.5e-6 of the largest
// score against 1.1e-6 for the fp32-MFMA / fmaf chain and 1.2e-6 for an fp32 BLAS product (profiles/archive/r02_score_split_*) -
// at 3/16 of the fp32 matrix pipe's time per score.  The power-of-two scale (per tensor, from its largest magnitude) keeps
// xl out of the half subnormals for every element that matters; it is exact and undone exactly.
__global__ __launch_bounds__(256) void maxabs_kernel(const float* __restrict__ X, size_t n, uint32_t* __restrict__ out) {
  uint32_t m = 0;
  auto one = [&](float x) {
    const uint32_t b = __float_as_uint(x) & 0x7fffffffu;
    if (b < 0x7f800000u) m = max(m, b);  // finite magnitudes only (bit patterns of non-negative floats order like the floats)
  };
  const float4* X4 = reinterpret_cast<const float4*>(X);  // (16-byte aligned, n % 4 == 0: checked by the caller)
  const size_t n4 = n / 4, step = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * step < n4; i += 4 * step) {  // four loads in flight per thread
    const float4 a = X4[i], b = X4[i + step], c = X4[i + 2 * step], d = X4[i + 3 * step];
    one(a.x); one(a.y); one(a.z); one(a.w); one(b.x); one(b.y); one(b.z); one(b.w);
    one(c.x); one(c.y); one(c.z); one(c.w); one(d.x); one(d.y); one(d.z); one(d.w);
  }
  for (; i < n4; i += step) {
    const float4 v = X4[i];
    one(v.x); one(v.y); one(v.z); one(v.w);
  }
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
// exps[which] = s with 2^14 <= max|x| 2^s < 2^15 (0 for an all-zero tensor)
__global__ void scale_exp_kernel(const uint32_t* __restrict__ maxbits, int* __restrict__ exps) {
  const int which = threadIdx.x;
  if (which < 2) {
    const uint32_t b = maxbits[which];
    int e = 0;
    if (b) {
      int ex;
      frexpf(__uint_as_float(b), &ex);  // max = f 2^ex, f in [0.5, 1)
      e = 15 - ex;
    }
    exps[which] = e;
  }
}
// Xc [rows_pad, parts Hp] half, zero padding in rows and columns.  parts = 3: Q rows as [xl | xh | xh], P rows as [xh | xl | xh]
// (is_p).  parts = 1 (mode 2, opt-in): the high halves alone - scores of IEEE-half operands with fp32 accumulation.
__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ X, _Float16* __restrict__ Xc, int rows, int rows_pad, int H,
                                                        int Hp, const int* __restrict__ exps, int is_p, int parts) {
  typedef _Float16 half4 __attribute__((ext_vector_type(4)));
  const float scale = ldexpf(1.0f, exps[is_p]);
  const int q = Hp / 4;  // four columns per thread (H % 4 == 0: a group of four is all data or all padding)
  const size_t total = (size_t)rows_pad * q;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / q), c = (int)(i - (size_t)r * q) * 4;
    half4 hi = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f}, lo = hi;
    if (r < rows && c < H) {
      const float4 x4 = *reinterpret_cast<const float4*>(X + (size_t)r * H + c);
      const float x[4] = {x4.x * scale, x4.y * scale, x4.z * scale, x4.w * scale};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = (_Float16)x[e];
        lo[e] = (_Float16)(x[e] - (float)hi[e]);
      }
    }
    _Float16* row = Xc + (size_t)r * parts * Hp + c;
    if (parts == 1) { *reinterpret_cast<half4*>(row) = hi; }
    else if (is_p) { *reinterpret_cast<half4*>(row) = hi; *reinterpret_cast<half4*>(row + Hp) = lo; *reinterpret_cast<half4*>(row + 2 * Hp) = hi; }
    else { *reinterpret_cast<half4*>(row) = lo; *reinterpret_cast<half4*>(row + Hp) = hi; *reinterpret_cast<half4*>(row + 2 * Hp) = hi; }
  }
}

// 0 = auto (split precision - fp32-accurate scores on the 16-bit pipe - when the workspace allows), 1 = exact fp32 MFMA always,
// 2 = half-precision scores (opt-in: ONE product of the operands rounded to IEEE half, fp32 accumulation - what a faiss fp16 flat
//     index computes; a third of mode 0's matrix work, score error ~1e-5 of |q||p| instead of ~1e-7)
int g_score_mode = -1;
int score_mode() {
  if (g_score_mode < 0) {
    const char* e = getenv("COCODR_SCORE_EXACT");
    g_score_mode = (e && atoi(e) != 0) ? 1 : 0;
  }
  return g_score_mode;
}
int score_parts() { return score_mode() == 2 ? 1 : 3; }
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
struct SplitPlan {
  int Hp, np_pad, QC, parts;
  size_t off_q, off_p, off_misc, off_slab, total;
};
SplitPlan split_plan(int Nq, int Np, int H) {
  SplitPlan s;
  s.parts = score_parts();
  s.Hp = (H + 63) / 64 * 64;
  s.np_pad = (Np + 255) / 256 * 256;
  long long qc = (1ll << 28) / s.np_pad;  // two slabs of <= 1 GiB
  qc = std::max(128ll, std::min(qc, 8192ll));
  qc = (qc / 128) * 128;
  s.QC = (int)std::min<long long>(qc, ((long long)Nq + 127) / 128 * 128);
  size_t o = 0;
  s.off_q = o; o = align256(o + (size_t)Nq * s.parts * s.Hp * 2);
  s.off_p = o; o = align256(o + (size_t)s.np_pad * s.parts * s.Hp * 2);
  s.off_misc = o; o = align256(o + 64);
  s.off_slab = o; o = align256(o + 2 * (size_t)s.QC * s.np_pad * 4);
  s.total = o;
  return s;
}

// The filtered search (score_filter.h): thresholds from a strided sample of ns passages, candidate blocks instead of a score slab.
// Everything it needs beyond the legacy layout sits behind sp.total; the slab area is reused, one after the other, for the sample's
// scores, the candidate blocks and the exhaustive pass over the rows handed back.
struct FilterPlan {
  bool on = false;
  int ns = 0, j = 0, capt = 0, ntn = 0, QF = 0, FBR = 0, lcap = 0;
  long long stride = 1;
  size_t off_ps = 0, off_qfb = 0, off_dt = 0, off_fb = 0, area = 0, total = 0;
};
int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
FilterPlan filter_plan(int Nq, int Np, int k, const SplitPlan& sp) {
  FilterPlan f;
  f.total = sp.total;
  if (score_mode() == 1 || env_int("COCODR_SCORE_NOFILTER", 0) != 0) return f;
  // where it pays (tools/search_crossover.py, H = 768): the extra launches (sample product, thresholds, gated exhaustive pass: ~60 us)
  // cost 2-22 % at <= 256 queries x <= 125 000 passages, level at 256 x 125 000 / 16 x 500 000, +6 ... +33 % from 2 000 queries up
  const bool force = env_int("COCODR_SCORE_FILTER_FORCE", 0) != 0;  // (test hook: small searches through this path)
  if (!force && (Np < 32768 || (long long)Nq * Np < (100ll << 20))) return f;
  // (measured up to ~1 M passages per call, tools/search_crossover.py; beyond ~2 M the fixed candidate blocks per row - Np / 256 of
  //  them - outgrow what the selection reads comfortably: larger shards take the exhaustive route until that regime is measured)
  if (!force && Np > (2 << 20)) return f;
  if (k > KMAX || (long long)k * 16 > Np) return f;
  // sample: every stride-th passage, ~Np / 32 of them (the sample's product is that fraction of extra matrix work)
  int ns = (Np / 32 + 255) / 256 * 256;
  ns = std::max(std::min(ns, 32768), std::min(4096, Np / 256 * 256));
  if (ns < 256) return f;
  f.ns = ns;
  f.stride = Np / ns;
  const double rho = (double)Np / ns;
  // the sample holds ~mu of the row's k best; its j-th best, j a few deviations past mu, is with near certainty BELOW the k-th best
  // of the row (the same cut topk_kernel makes inside a row) - then the scores >= it number ~j rho +- sqrt(j) rho
  const double mu = (double)k / rho;
  int j = (int)(mu + 4.0 * sqrt(mu) + 8.0);
  j = env_int("COCODR_SCORE_FILTER_J", j);  // (test hook: a rank too small hands every row back)
  j = std::max(1, std::min(j, ns));
  f.j = j;
  const double expect = j * rho, dev = sqrt((double)j) * rho;
  if (expect + 6.0 * dev > FCAP) return f;
  f.lcap = expect + 6.0 * dev <= FCAP_SMALL ? FCAP_SMALL : FCAP;
  f.ntn = sp.np_pad / 256;
  const double per_tile = expect / f.ntn;
  int capt = ((int)(2.0 * per_tile + 6.0 * sqrt(per_tile) + 16.0) + 1 + 7) / 8 * 8;  // entries + the header, whole 64-byte lines
  capt = env_int("COCODR_SCORE_FILTER_CAPT", capt);  // (test hook: tiny blocks overflow)
  f.capt = std::max(8, std::min(capt, 512)) / 8 * 8;
  f.area = 2 * (size_t)sp.QC * sp.np_pad * 4;
  long long qf = std::min<long long>({((long long)Nq + 255) / 256 * 256, (long long)(f.area / ((size_t)f.ntn * f.capt * 8)),
                                      (long long)(f.area / ((size_t)ns * 4)), 16384ll});
  qf = qf / 256 * 256;
  if (qf < 256) return f;
  f.QF = (int)qf;
  f.FBR = (int)std::min<long long>(f.QF, (long long)(f.area / ((size_t)sp.np_pad * 4)) / 128 * 128);
  if (f.FBR < 128) return f;
  const size_t row_bytes = (size_t)sp.parts * sp.Hp * 2;
  size_t o = sp.total;
  f.off_ps = o; o = align256(o + (size_t)ns * row_bytes);
  f.off_qfb = o; o = align256(o + (size_t)f.QF * row_bytes);
  f.off_dt = o; o = align256(o + (size_t)f.QF * 4);
  f.off_fb = o; o = align256(o + ((size_t)f.QF + 64) * 4);
  f.total = o;
  f.on = true;
  return f;
}

int query_chunk(int nq, int np) {
  long long qc = (1ll << 28) / std::max(np, 1);  // two slabs of <= 1 GiB
  qc = std::max(128ll, std::min(qc, 8192ll));
  qc = (qc / 128) * 128;
  return (int)std::min<long long>(qc, ((long long)nq + 127) / 128 * 128);
}

// side stream + events of the GEMM / selection pipeline (created once per process; the library still owns no threads
// and allocates no device memory)
struct Pipe {
  hipStream_t side = nullptr;
  hipEvent_t scored[2] = {nullptr, nullptr}, selected[2] = {nullptr, nullptr};
  bool ok = false;
};
Pipe& pipe() {
  static Pipe p;
  if (!p.ok) {
    bool good = hipStreamCreateWithFlags(&p.side, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 2 && good; ++i)
      good = hipEventCreateWithFlags(&p.scored[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&p.selected[i], hipEventDisableTiming) == hipSuccess;
    p.ok = good;
  }
  return p;
}

}  // namespace

namespace {
size_t exact_bytes(int Nq, int Np) {
  const size_t ld = ((size_t)Np + 3) / 4 * 4;
  return 2 * (size_t)query_chunk(Nq, Np) * ld * sizeof(float);
}
}  // namespace

extern "C" size_t cocodr_score_topk_workspace_bytes_dim(int Nq, int Np, int H, int k) {
  if (Nq <= 0 || Np <= 0 || H <= 0) return 0;
  if (score_mode() == 1) return exact_bytes(Nq, Np);
  const SplitPlan sp = split_plan(Nq, Np, H);
  return std::max(exact_bytes(Nq, Np), filter_plan(Nq, Np, k, sp).total);  // (mode 2 needs less than mode 0)
}
extern "C" int cocodr_score_filter_plan(int Nq, int Np, int H, int k, long long out[8]) {
  CK_ARG(out != nullptr && Nq > 0 && Np > 0 && H > 0 && k > 0, "score_filter_plan: bad argument");
  const SplitPlan sp = split_plan(Nq, Np, H);
  const FilterPlan f = filter_plan(Nq, Np, k, sp);
  out[0] = f.on ? 1 : 0; out[1] = f.ns; out[2] = f.stride; out[3] = f.j; out[4] = f.capt; out[5] = f.QF; out[6] = f.FBR;
  out[7] = (long long)f.off_fb;
  return COCODR_OK;
}
extern "C" size_t cocodr_score_topk_workspace_bytes(int Nq, int Np, int k) { return cocodr_score_topk_workspace_bytes_dim(Nq, Np, 1024, k); }
extern "C" int cocodr_score_set_mode(int mode) {
  CK_ARG(mode >= 0 && mode <= 2, "score_set_mode: 0 = auto (split precision), 1 = exact fp32 MFMA, 2 = half-precision scores");
  g_score_mode = mode;
  return COCODR_OK;
}

namespace {
// passages per launch of the split-precision product: the GEMM addresses an operand with 32-bit byte offsets, so they go in column
// blocks of < 4 GiB of half operands
int passage_block(const SplitPlan& sp) {
  int pblk = (int)std::min<long long>(sp.np_pad, ((1ll << 32) / ((long long)sp.parts * sp.Hp * 2) - 1) / 256 * 256);
  if (const char* e = getenv("COCODR_SCORE_PBLK")) pblk = std::max(256, std::min(pblk, atoi(e) / 256 * 256));  // test hook: small column blocks
  return pblk;
}

// The filtered search over the half operands Qc / Pc (split_f16_kernel's): per chunk of QF query rows
//   scores of the passage sample -> per-row threshold (topk_kernel on the sample, its j-th best) -> the full product with the filter
//   epilogue -> selection from the candidate blocks -> the exhaustive pass (device-gated: it costs three empty launches when, as
//   usual, no row was handed back).  Everything on the caller's stream.  Exact: a row is answered from its candidates only when they
//   provably contain its k best (>= k scores at or above the threshold, nothing overflowed); the order is topk_kernel's.
int filtered_search(const _Float16* Qc, const _Float16* Pc, const int* exps, int Nq, int Np, int H, int k, long long id_offset, float* D,
                    long long* I, char* wsb, const SplitPlan& sp, const FilterPlan& fp, hipStream_t st, bool p_resident = false) {
  const int K = sp.parts * sp.Hp;
  _Float16* Ps = reinterpret_cast<_Float16*>(wsb + fp.off_ps);
  _Float16* Qfb = reinterpret_cast<_Float16*>(wsb + fp.off_qfb);
  float* Dt = reinterpret_cast<float*>(wsb + fp.off_dt);
  int* fb_count = reinterpret_cast<int*>(wsb + fp.off_fb);
  int* fb_rows = fb_count + 64;
  float* area = reinterpret_cast<float*>(wsb + sp.off_slab);
  const int pblk = passage_block(sp);
  if (!p_resident) {  // (a resident index keeps its passage sample next to its half operands)
    hipLaunchKernelGGL(copy_rows_kernel, dim3(fp.ns), dim3(256), 0, st, Pc, Ps, K, fp.stride, (const int*)nullptr, (const int*)nullptr, 0);
    CK_LAUNCH("score_sample");
  }
  auto product = [&](const _Float16* A, int M, const _Float16* B, int N, float* C, long long ldc) {
    cocodr_gemm_args g = {};
    g.A = reinterpret_cast<const uint16_t*>(A);
    g.B = reinterpret_cast<const uint16_t*>(B);
    g.C = C;
    g.M = M; g.N = N; g.K = K;
    g.lda = g.ldb = K; g.ldc = (int)ldc;
    g.out_f32 = 1; g.batch = 1; g.ab_f16 = 1;
    return g;
  };
  for (int q0 = 0; q0 < Nq; q0 += fp.QF) {
    const int nq = std::min(fp.QF, Nq - q0);
    const _Float16* Aq = Qc + (size_t)q0 * K;
    {
      ProfScope prof(PROF_SCORE, st, 2.0 * nq * (double)Np * H);  // (the sample's product is overhead: its time counts, its FLOPs do not)
      // thresholds need no fp32 accuracy (they only steer the filter): the sample is scored by the product of the HIGH halves alone -
      // columns [Hp, 2 Hp) of a query row [xl | xh | xh], [0, Hp) of a passage row [xh | xl | xh] - a third of the contraction
      cocodr_gemm_args g = product(Aq + (sp.parts == 3 ? sp.Hp : 0), nq, Ps, fp.ns, area, fp.ns);
      g.K = sp.Hp;
      const int rc = cocodr_gemm(&g, (cocodr_stream_t)st);
      if (rc != COCODR_OK) return rc;
    }
    hipLaunchKernelGGL(row_rank_kernel, dim3(nq), dim3(TOPK_THREADS), 0, st, area, (long long)fp.ns, fp.ns, fp.j, Dt);
    CK_LAUNCH("score_thresholds");
    if (hipMemsetAsync(fb_count, 0, 4, st) != hipSuccess) { cocodr_set_error("score_topk: memset failed"); return COCODR_ERR_LAUNCH; }
    {
      ProfScope prof(PROF_SCORE, st, 0.0);
      for (int p0 = 0; p0 < sp.np_pad; p0 += pblk) {
        const cocodr_gemm_args g = product(Aq, nq, Pc + (size_t)p0 * K, std::min(pblk, sp.np_pad - p0), nullptr, 0);
        cocodr_score_filter f = {};
        f.mode = 0;
        f.thr = Dt; f.thr_stride = 1;
        f.cand = reinterpret_cast<uint2*>(area);
        f.capt = fp.capt; f.ntn_total = fp.ntn; f.tile0 = p0 / 256;
        f.col0 = p0; f.n_valid = Np;
        cocodr_gemm_pp_launch_filter(g, f, st);
      }
      CK_LAUNCH("score_filter_gemm");
    }
    if (fp.lcap <= FCAP_SMALL)
      hipLaunchKernelGGL(filter_select_kernel<FCAP_SMALL>, dim3(nq), dim3(FSEL_THREADS), 0, st, reinterpret_cast<const uint2*>(area), fp.ntn, fp.capt, k,
                         id_offset, D + (size_t)q0 * k, I + (size_t)q0 * k, exps, fb_count, fb_rows);
    else
      hipLaunchKernelGGL(filter_select_kernel<FCAP>, dim3(nq), dim3(FSEL_THREADS), 0, st, reinterpret_cast<const uint2*>(area), fp.ntn, fp.capt, k,
                         id_offset, D + (size_t)q0 * k, I + (size_t)q0 * k, exps, fb_count, fb_rows);
    CK_LAUNCH("score_filter_select");
    for (int base = 0; base < nq; base += fp.FBR) {  // the rows handed back, FBR at a time (gated by the device-side count)
      const int rows = std::min(fp.FBR, nq - base);
      hipLaunchKernelGGL(copy_rows_kernel, dim3(rows), dim3(256), 0, st, Aq, Qfb, K, 0ll, (const int*)fb_rows, (const int*)fb_count, base);
      for (int p0 = 0; p0 < sp.np_pad; p0 += pblk) {
        const cocodr_gemm_args g = product(Qfb, rows, Pc + (size_t)p0 * K, std::min(pblk, sp.np_pad - p0), area + p0, sp.np_pad);
        cocodr_score_filter f = {};
        f.mode = 1;
        f.m_dev = fb_count; f.m_base = base;
        cocodr_gemm_pp_launch_filter(g, f, st);
      }
      hipLaunchKernelGGL(topk_kernel, dim3(rows), dim3(TOPK_THREADS), 0, st, area, (long long)sp.np_pad, Np, k, id_offset, D + (size_t)q0 * k,
                         I + (size_t)q0 * k, exps, (const int*)fb_rows, (const int*)fb_count, base);
      CK_LAUNCH("score_filter_fallback");
    }
  }
  return COCODR_OK;
}
}  // namespace

static int score_topk_impl(const float* Q, const float* P, int Nq, int Np, int H, int k, long long id_offset, float* D, long long* I,
                           void* workspace, size_t workspace_bytes, bool p_resident, cocodr_stream_t stream);

extern "C" int cocodr_score_topk(const float* Q, const float* P, int Nq, int Np, int H, int k, long long id_offset, float* D,
                                 long long* I, void* workspace, size_t workspace_bytes, cocodr_stream_t stream) {
  return score_topk_impl(Q, P, Nq, Np, H, k, id_offset, D, I, workspace, workspace_bytes, false, stream);
}

// faiss.IndexFlatIP: add(P) once, search(Q, k) many times (ANCE/drivers/run_ann_data_gen.py:310-317,390; evaluate_beir.py:220-224).
// p_resident != 0: the caller vouches that `workspace` still holds what the previous cocodr_score_topk[_resident] call with the SAME
// (P, Np, H, Nq, k, score mode) left there - the passages' scale, their half-precision split image and the filter's passage sample -
// so none of it is rebuilt (3 of the search's launches and a pass over P).  The exact-fp32 mode has no passage image: flag ignored.
extern "C" int cocodr_score_topk_resident(const float* Q, const float* P, int Nq, int Np, int H, int k, long long id_offset, float* D,
                                          long long* I, void* workspace, size_t workspace_bytes, int p_resident, cocodr_stream_t stream) {
  return score_topk_impl(Q, P, Nq, Np, H, k, id_offset, D, I, workspace, workspace_bytes, p_resident != 0, stream);
}

static int score_topk_impl(const float* Q, const float* P, int Nq, int Np, int H, int k, long long id_offset, float* D, long long* I,
                           void* workspace, size_t workspace_bytes, bool p_resident, cocodr_stream_t stream) {
  CK_ARG(Q && P && D && I && workspace, "score_topk: null pointer");
  CK_ARG(Nq > 0 && Np > 0 && H > 0 && H % 4 == 0, "score_topk: bad shape Nq=%d Np=%d H=%d (H %% 4 == 0)", Nq, Np, H);
  CK_ARG(k > 0 && k <= KMAX, "score_topk: k=%d must be in [1,%d]", k, KMAX);
  CK_ARG((((uintptr_t)Q | (uintptr_t)P) & 15) == 0, "score_topk: Q and P must be 16-byte aligned");
  if (workspace_bytes < exact_bytes(Nq, Np)) {
    cocodr_set_error("score_topk: workspace %zu B < required %zu B", workspace_bytes, cocodr_score_topk_workspace_bytes_dim(Nq, Np, H, k));
    return COCODR_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  CK_ARG(((uintptr_t)workspace & 255) == 0, "score_topk: workspace must be 256-byte aligned");
  // split precision (16-bit matrix pipe) when the workspace has room for the half operands; exact fp32 MFMA otherwise
  const SplitPlan sp = split_plan(Nq, Np, H);
  const bool split = score_mode() != 1 && workspace_bytes >= sp.total;
  char* wsb = reinterpret_cast<char*>(workspace);
  _Float16* Qc = reinterpret_cast<_Float16*>(wsb + sp.off_q);
  _Float16* Pc = reinterpret_cast<_Float16*>(wsb + sp.off_p);
  uint32_t* maxbits = reinterpret_cast<uint32_t*>(wsb + sp.off_misc);
  int* exps = reinterpret_cast<int*>(wsb + sp.off_misc + 16);
  if (split) {
    // (resident passages: their maximum - maxbits[1] - and so their scale survive from the call that built the image)
    if (hipMemsetAsync(maxbits, 0, p_resident ? 4 : 32, st) != hipSuccess) { cocodr_set_error("score_topk: memset failed"); return COCODR_ERR_LAUNCH; }
    hipLaunchKernelGGL(maxabs_kernel, dim3(1024), dim3(256), 0, st, Q, (size_t)Nq * H, maxbits);
    if (!p_resident) hipLaunchKernelGGL(maxabs_kernel, dim3(2048), dim3(256), 0, st, P, (size_t)Np * H, maxbits + 1);
    hipLaunchKernelGGL(scale_exp_kernel, dim3(1), dim3(64), 0, st, maxbits, exps);
    hipLaunchKernelGGL(split_f16_kernel, dim3(2048), dim3(256), 0, st, Q, Qc, Nq, Nq, H, sp.Hp, exps, 0, sp.parts);
    if (!p_resident) hipLaunchKernelGGL(split_f16_kernel, dim3(4096), dim3(256), 0, st, P, Pc, Np, sp.np_pad, H, sp.Hp, exps, 1, sp.parts);
    CK_LAUNCH("score_split");
    const FilterPlan fp = filter_plan(Nq, Np, k, sp);
    if (fp.on && workspace_bytes >= fp.total) return filtered_search(Qc, Pc, exps, Nq, Np, H, k, id_offset, D, I, wsb, sp, fp, st, p_resident);
  }
  const int QC = split ? sp.QC : query_chunk(Nq, Np);
  const long long ld = split ? sp.np_pad : ((long long)Np + 3) / 4 * 4;
  float* slab0 = split ? reinterpret_cast<float*>(wsb + sp.off_slab) : reinterpret_cast<float*>(workspace);
  float* slab[2] = {slab0, slab0 + (size_t)QC * ld};
  static const bool serial = getenv("COCODR_SCORE_SERIAL") != nullptr;  // A/B switch: selection on the caller's stream
  static std::mutex pipe_mutex;  // the side stream and its events are shared by all callers: enqueue one search at a time
  std::lock_guard<std::mutex> guard(pipe_mutex);
  Pipe& pp = pipe();
  const bool piped = pp.ok && !serial && Nq > QC;
  int c = 0;
  for (int q0 = 0; q0 < Nq; q0 += QC, ++c) {
    const int nq = std::min(QC, Nq - q0);
    const int ntm = (nq + SB - 1) / SB, ntn = (Np + SB - 1) / SB;
    float* S = slab[c & 1];
    if (piped && c >= 2 && hipStreamWaitEvent(st, pp.selected[c & 1], 0) != hipSuccess) {  // the slab is free again
      cocodr_set_error("score_topk: stream wait failed");
      return COCODR_ERR_LAUNCH;
    }
    if (split) {
      ProfScope prof(PROF_SCORE, st, 2.0 * nq * (double)Np * H);  // algorithmic FLOPs of the scores, whatever pipe produces them
      const int pblk = passage_block(sp);
      for (int p0 = 0; p0 < sp.np_pad; p0 += pblk) {
        cocodr_gemm_args g = {};
        g.A = reinterpret_cast<const uint16_t*>(Qc + (size_t)q0 * sp.parts * sp.Hp);
        g.B = reinterpret_cast<const uint16_t*>(Pc + (size_t)p0 * sp.parts * sp.Hp);
        g.C = S + p0;
        g.M = nq; g.N = std::min(pblk, sp.np_pad - p0); g.K = sp.parts * sp.Hp;
        g.lda = g.ldb = sp.parts * sp.Hp; g.ldc = (int)ld;
        g.out_f32 = 1; g.batch = 1; g.ab_f16 = 1;
        const int rc = cocodr_gemm(&g, stream);
        if (rc != COCODR_OK) return rc;
      }
    } else {
      ProfScope prof(PROF_SCORE, st, 2.0 * nq * (double)Np * H);
      hipLaunchKernelGGL(score_gemm_kernel, dim3(ntm * ntn), dim3(256), 0, st, Q + (size_t)q0 * H, P, S, nq, Np, H, ld);
      CK_LAUNCH("score_gemm");
    }
    hipStream_t sel = st;
    if (piped) {
      sel = pp.side;
      if (hipEventRecord(pp.scored[c & 1], st) != hipSuccess || hipStreamWaitEvent(sel, pp.scored[c & 1], 0) != hipSuccess) {
        cocodr_set_error("score_topk: event hand-off failed");
        return COCODR_ERR_LAUNCH;
      }
    }
    hipLaunchKernelGGL(topk_kernel, dim3(nq), dim3(TOPK_THREADS), 0, sel, S, ld, Np, k, id_offset, D + (size_t)q0 * k, I + (size_t)q0 * k,
                       split ? exps : (const int*)nullptr);
    CK_LAUNCH("topk");
    if (piped && hipEventRecord(pp.selected[c & 1], sel) != hipSuccess) {
      cocodr_set_error("score_topk: event record failed");
      return COCODR_ERR_LAUNCH;
    }
  }
  if (piped)  // results are complete in the caller's stream order
    for (int i = 0; i < std::min(c, 2); ++i)
      if (hipStreamWaitEvent(st, pp.selected[i], 0) != hipSuccess) {
        cocodr_set_error("score_topk: final stream wait failed");
        return COCODR_ERR_LAUNCH;
      }
  return COCODR_OK;
}
