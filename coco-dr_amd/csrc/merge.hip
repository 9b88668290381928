// k-way merge of per-shard top-k lists (SURVEY 8e "scoring" row: reduce-scatter by query block, each rank merges Nq / W
// queries).  What it replaces in the reference: rank 0 un-pickles every shard's embeddings, concatenates them and runs ONE
// faiss search over the whole corpus (ANCE/utils/util.py:117-155 + evaluate/evaluation/evaluate_beir.py:200-224); with the
// corpus resident and searched shard by shard, the per-shard (score, position) lists have to be merged into the list the
// single search would have returned: (score descending, global position ascending), global position = shard offset + local.
//
// One workgroup per query.  The W lists are SORTED (cocodr_score_topk's contract), so no sort is needed: the merged rank of
// element i of list w is i + sum over the other lists w' of the number of their elements that precede it, each found by a
// binary search in LDS (elements of a lower shard win ties, those of a higher shard lose them).  W k log2(k) LDS reads per
// query, deterministic, exact; positions travel as int32 (local to their shard) and leave as int64 global positions.
#include <math.h>

#include "common.h"

namespace {
struct MergeArgs {
  const float* D;
  const int32_t* I;
  const long long* shard_off;
  float* outD;
  long long* outI;
  int W, Nq, k, k_out;
  long long stride_w;  // elements between the lists of shard w and w + 1 (>= Nq * k)
};

__global__ __launch_bounds__(256) void topk_merge_kernel(const MergeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sc[];  // [W][k] scores
  __shared__ int n_valid[64];
  const int q = blockIdx.x, tid = threadIdx.x;
  const int W = a.W, k = a.k, n = W * k;
  const float NEG = -INFINITY;
  // validity comes from the POSITION alone (I < 0 = empty slot, at the end of a list): a real candidate whose score is -inf is a
  // candidate like any other.  (NaN scores are not supported: they break the order the lists are sorted by - include/cocodr.h.)
  if (tid < W) {
    const int32_t* Iw = a.I + (size_t)tid * a.stride_w + (size_t)q * k;
    int lo = 0, hi = k;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (Iw[mid] >= 0) lo = mid + 1;
      else hi = mid;
    }
    n_valid[tid] = lo;
  }
  for (int e = tid; e < n; e += 256) {
    const int w = e / k, i = e - w * k;
    sc[e] = a.D[(size_t)w * a.stride_w + (size_t)q * k + i];
  }
  __syncthreads();
  for (int e = tid; e < n; e += 256) {
    const int w = e / k, i = e - w * k;
    if (i >= n_valid[w]) continue;
    const float s = sc[e];
    int rank = i;
    for (int w2 = 0; w2 < W && rank < a.k_out; ++w2) {
      if (w2 == w) continue;
      const float* L = sc + w2 * k;
      int lo = 0, hi = n_valid[w2];
      if (w2 < w) {  // a lower shard wins ties: count its elements >= s
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (L[mid] >= s) lo = mid + 1;
          else hi = mid;
        }
      } else {       // a higher shard loses them: count its elements > s
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (L[mid] > s) lo = mid + 1;
          else hi = mid;
        }
      }
      rank += lo;
    }
    if (rank < a.k_out) {
      const size_t g = (size_t)w * a.stride_w + (size_t)q * k + i;
      a.outD[(size_t)q * a.k_out + rank] = s;
      a.outI[(size_t)q * a.k_out + rank] = a.shard_off[w] + (long long)a.I[g];
    }
  }
  int total = 0;
  for (int w = 0; w < W; ++w) total += n_valid[w];
  for (int r = total + tid; r < a.k_out; r += 256) {  // fewer candidates than k_out: (-inf, -1) padding, as cocodr_score_topk
    a.outD[(size_t)q * a.k_out + r] = NEG;
    a.outI[(size_t)q * a.k_out + r] = -1;
  }
}
}  // namespace

extern "C" int cocodr_topk_merge(const float* D, const int32_t* I, const long long* shard_offset, int W, int Nq, int k,
                                 long long stride_w, float* outD, long long* outI, int k_out, cocodr_stream_t stream) {
  CK_ARG(D && I && shard_offset && outD && outI, "topk_merge: null pointer");
  CK_ARG(W >= 1 && W <= 64 && Nq >= 0 && k >= 1 && k_out >= 1, "topk_merge: bad sizes W=%d Nq=%d k=%d k_out=%d", W, Nq, k, k_out);
  CK_ARG(k_out <= W * k, "topk_merge: k_out=%d exceeds the %d candidates per query", k_out, W * k);
  CK_ARG(stride_w >= (long long)Nq * k, "topk_merge: stride_w smaller than one shard's lists");
  const size_t lds = (size_t)W * k * sizeof(float);
  CK_ARG(lds <= 156 * 1024, "topk_merge: W * k = %d candidates per query exceed the LDS (max 39936)", W * k);
  if (Nq == 0) return COCODR_OK;
  static cocodr_lds_once attr_once;  // (the raised LDS limit is a per-device attribute)
  if (attr_once.pending()) {
    hipFuncSetAttribute((const void*)topk_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    attr_once.done();
  }
  MergeArgs a{D, I, shard_offset, outD, outI, W, Nq, k, k_out, stride_w};
  hipLaunchKernelGGL(topk_merge_kernel, dim3(Nq), dim3(256), lds, (hipStream_t)stream, a);
  CK_LAUNCH("topk_merge");
  return COCODR_OK;
}
