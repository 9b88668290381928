// On-device Condenser / coCondenser collator: COCO/data.py:24-156 - random truncation window,
// whole-word-mask proxy (words = a token plus its "##" continuations, shuffled, taken greedily up to
// round(len * mlm_probability) tokens), [CLS] .. [SEP] + padding, and the 80 / 10 / 10 replacement rule of
// DataCollatorForWholeWordMask.torch_mask_tokens.  One 64-lane wave per span; L <= 512.
//
// The reference draws from Python's `random` / torch's global generator, whose streams cannot be reproduced on a
// device; randomness here is a counter-based hash of (seed, span index, purpose, position), identical in
// oracle/collate_oracle.py, so kernel and oracle agree bit for bit while the algorithm is pinned to the reference's own
// methods on the CPU (tests/test_oracle_golden.py drives COCO/data.py with the same permutations).
// One deliberate difference: the reference calls `_truncate` twice with independent draws (data.py:131 for the mask,
// :137 for the ids), which shifts the mask against the tokens of an over-long span; here both use one window.
#include "common.h"

namespace {

constexpr int CL_MAX = 512;
enum { RS_TRUNC = 1, RS_SHUFFLE = 2, RS_REPLACE = 3, RS_RANDOM = 4, RS_WORD = 5 };

__host__ __device__ __forceinline__ uint32_t collate_rand(uint64_t seed, uint64_t ex, uint64_t stream, uint64_t ctr) {
  uint64_t x = seed * 0x9E3779B97F4A7C15ull + ex * 0xBF58476D1CE4E5B9ull + stream * 0x94D049BB133111EBull + ctr;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 32);
}
__device__ __forceinline__ float collate_uniform(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

__global__ __launch_bounds__(64) void collate_kernel(const int32_t* __restrict__ tokens, const long long* __restrict__ offsets,
                                                     const uint8_t* __restrict__ is_subword, int vocab, int L, int cls_id, int sep_id,
                                                     int pad_id, int mask_id, double mlm_prob, unsigned long long seed,
                                                     long long ex_base, int32_t* __restrict__ input_ids, int32_t* __restrict__ labels,
                                                     int32_t* __restrict__ attn) {
  __shared__ int tok[CL_MAX];
  __shared__ int word_of[CL_MAX];                 // word index of each token (-1: a special token, part of no word)
  __shared__ int word_len[CL_MAX];
  __shared__ unsigned char chosen[CL_MAX];        // per word: selected for masking
  __shared__ unsigned long long key[CL_MAX];      // (random key << 32) | word index, padded with ~0 for the bitonic sort
  __shared__ unsigned char masked[CL_MAX];
  __shared__ int s_nwords;
  const int lane = threadIdx.x;
  const long long ex = blockIdx.x;
  const unsigned long long exg = (unsigned long long)(ex_base + ex);
  const long long o0 = offsets[ex], o1 = offsets[ex + 1];
  const int len_raw = (int)(o1 - o0);
  const int tgt = L - 2;                           // num_special_tokens_to_add(pair=False) == 2, data.py:103
  int n = len_raw, left = 0;
  if (len_raw > tgt) {                             // data.py:104-112
    const int trunc = len_raw - tgt;
    left = (int)(collate_rand(seed, exg, RS_TRUNC, 0) % (uint32_t)(trunc + 1));
    n = tgt;
  }
  // ---- tokens and word structure (data.py:44-55): a "##" piece joins the word in front of it; a special token inside
  // the span ([UNK], a stray [SEP] ...: vocabulary class 2) is skipped - it belongs to no word and is never masked, and a
  // "##" piece behind it still joins the last word that was opened
  for (int i = lane; i < n; i += 64) {
    int t = tokens[o0 + left + i];
    t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
    tok[i] = t;
    masked[i] = 0;
    chosen[i] = 0;
  }
  __syncthreads();
  if (lane == 0) {                                 // n <= 510: a short serial scan
    int nw = 0;
    for (int i = 0; i < n; ++i) {
      const int cls = is_subword[tok[i]];
      if (cls == 2) { word_of[i] = -1; continue; }
      const bool cont = nw >= 1 && cls == 1;
      if (!cont) { word_len[nw] = 0; ++nw; }
      word_of[i] = nw - 1;
      word_len[nw - 1]++;
    }
    s_nwords = nw;
  }
  __syncthreads();
  const int nw = s_nwords;
  // ---- random.shuffle(cand_indexes) (data.py:76): sort the words by a random key
  int npad = 1;
  while (npad < nw) npad <<= 1;
  for (int w = lane; w < npad; w += 64)
    key[w] = w < nw ? (((unsigned long long)collate_rand(seed, exg, RS_SHUFFLE, (uint64_t)w) << 32) | (unsigned)w) : ~0ull;
  __syncthreads();
  for (int size = 2; size <= npad; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = lane; t < (npad >> 1); t += 64) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = key[lo], b = key[hi];
        if ((a > b) == up) { key[lo] = b; key[hi] = a; }
      }
      __syncthreads();
    }
  // ---- greedy whole-word selection (data.py:77-99)
  if (lane == 0 && n > 0) {
    long long want = (long long)rint((double)n * mlm_prob);   // Python round(): half to even on the float64 product
    if (want < 1) want = 1;
    if (want > 512) want = 512;                               // max_predictions
    int taken = 0;
    for (int j = 0; j < nw; ++j) {
      if (taken >= want) break;
      const int w = (int)(key[j] & 0xffffffffu);
      if (taken + word_len[w] > want) continue;
      chosen[w] = 1;
      taken += word_len[w];
    }
  }
  __syncthreads();
  for (int i = lane; i < n; i += 64) masked[i] = word_of[i] >= 0 && chosen[word_of[i]];
  __syncthreads();
  // ---- [CLS] tokens [SEP] pad (data.py:135-144) + 80 / 10 / 10 (torch_mask_tokens)
  int32_t* ids_row = input_ids + ex * L;
  int32_t* lab_row = labels + ex * L;
  int32_t* att_row = attn + ex * L;
  for (int p = lane; p < L; p += 64) {
    int id = pad_id, lab = -100, att = 0;
    if (p == 0) { id = cls_id; att = 1; }
    else if (p <= n) {
      const int i = p - 1;
      id = tok[i];
      att = 1;
      if (masked[i]) {
        lab = id;
        if (collate_uniform(collate_rand(seed, exg, RS_REPLACE, (uint64_t)p)) < 0.8f) id = mask_id;
        else if (collate_uniform(collate_rand(seed, exg, RS_RANDOM, (uint64_t)p)) < 0.5f)
          id = (int)(collate_rand(seed, exg, RS_WORD, (uint64_t)p) % (uint32_t)vocab);
      }
    } else if (p == n + 1) { id = sep_id; att = 1; }
    ids_row[p] = id;
    lab_row[p] = lab;
    att_row[p] = att;
  }
}

}  // namespace

extern "C" int cocodr_mlm_collate(const int32_t* tokens, const long long* offsets, int n_spans, const uint8_t* is_subword, int vocab,
                                  int max_seq_length, int cls_id, int sep_id, int pad_id, int mask_id, double mlm_probability,
                                  unsigned long long seed, long long span_index_base, int32_t* input_ids, int32_t* labels,
                                  int32_t* attention_mask, cocodr_stream_t stream) {
  CK_ARG(tokens && offsets && is_subword && input_ids && labels && attention_mask, "mlm_collate: null pointer");
  CK_ARG(n_spans > 0 && vocab > 0, "mlm_collate: bad sizes");
  CK_ARG(max_seq_length >= 3 && max_seq_length <= CL_MAX, "mlm_collate: max_seq_length=%d must be in [3,%d]", max_seq_length, CL_MAX);
  CK_ARG(mlm_probability >= 0.0 && mlm_probability <= 1.0, "mlm_collate: bad mlm_probability");
  hipLaunchKernelGGL(collate_kernel, dim3(n_spans), dim3(64), 0, (hipStream_t)stream, tokens, offsets, is_subword, vocab, max_seq_length,
                     cls_id, sep_id, pad_id, mask_id, mlm_probability, seed, span_index_base, input_ids, labels, attention_mask);
  CK_LAUNCH("mlm_collate");
  return COCODR_OK;
}
