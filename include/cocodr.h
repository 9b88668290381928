/*
 * cocodr.h - C ABI of libcocodr_hip.so: the MI355X (gfx950) native implementation of the
 * COCO-DR contrastive dense-retrieval hot path.
 *
 * Everything the reference runs below its model-wrapper boundary is a stock
 * ATen/cuBLAS op reached through third-party `transformers`; the reference itself has no native
 * code and no FFI.  This header is therefore the boundary a maintainer binds INSTEAD of those
 * ops: plain pointers + sizes, a hipStream_t (the caller's current stream), no torch types.
 * Each entry point cites the reference call site(s) it replaces (paths under /root/reference,
 * `hf:` = transformers/models/bert/modeling_bert.py).
 *
 * Conventions
 *  - all pointers are DEVICE pointers unless the parameter name ends in _host;
 *  - bf16 tensors are passed as `const uint16_t*` (raw bfloat16 bits), row-major;
 *  - every function returns 0 on success, <0 on error (no exceptions cross the ABI);
 *    cocodr_last_error() returns a thread-local message for the last failure;
 *  - kernels are enqueued asynchronously on `stream`; the library owns no threads, allocates no
 *    device memory (workspaces are caller-provided) and is re-entrant per stream.
 */
#ifndef COCODR_H_
#define COCODR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cocodr_stream_t; /* hipStream_t */

enum {
  COCODR_OK = 0,
  COCODR_ERR_INVALID = -1, /* bad shape / alignment / null pointer */
  COCODR_ERR_LAUNCH = -2,  /* HIP launch or runtime failure */
  COCODR_ERR_WORKSPACE = -3 /* caller workspace too small */
};

const char* cocodr_last_error(void);
/* "gfx950" + build id; lets the host verify it loaded the native library, not a fallback */
const char* cocodr_build_info(void);

/* ------------------------------------------------------------------------------------------
 * Dropout  (hf: the nn.Dropout of BertEmbeddings :68-108, BertSelfAttention / eager_attention_forward :111-203 on the
 * attention probabilities, BertSelfOutput :282-293 and BertOutput :338-351 on the dense outputs in front of the residual
 * add; active under model.train(): ANCE/drivers/run_ann.py:293, and in the Condenser head layers of
 * COCO/modeling.py:212-220 - COCO keeps only the backbone in eval, :198).
 * torch's Philox stream cannot be reproduced bit for bit, so the mask is a counter-based hash with the same
 * distribution: for the element with row-major flat index i of the site's tensor
 *     w = lowbias32((i >> 1) ^ k0) ^ k1,   u = (i & 1) ? w >> 16 : w & 0xffff,   keep iff u >= threshold,
 * lowbias32(x): x ^= x >> 16; x *= 0x7feb352d; x ^= x >> 15; x *= 0x846ca68b; x ^= x >> 16   (32-bit wrap-around),
 * kept elements are multiplied by scale.  threshold = round(p * 65536), scale = 65536 / (65536 - threshold): the keep
 * probability is exact to 2^-16 and the scale matches it.  The mask is never stored: the backward regenerates it from
 * the same keys.  Site tensors: attention probabilities [B, heads, L, L]; dense outputs and the embedding output [M, H].
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  uint32_t k0, k1;
  uint32_t threshold; /* 0 = no dropout */
  float scale;
} cocodr_dropout_mask;
enum { COCODR_DROP_ATTN_PROBS = 0, COCODR_DROP_ATTN_OUT = 1, COCODR_DROP_FFN_OUT = 2, COCODR_DROP_EMBED = 3 };
/* Keys of one site of one forward call (host function, no device work):  site = 4 * layer + kind (embedding: layer 0),
 *   z = splitmix64(seed + 0x9E3779B97F4A7C15 * (call + 1));  z = splitmix64(z ^ (0xD1B54A32D192ED03 * (site + 1)));
 *   k0 = low 32 bits of z, k1 = high 32 bits,
 * splitmix64(x): x += 0x9E3779B97F4A7C15; x = (x ^ x >> 30) * 0xBF58476D1CE4E5B9; x = (x ^ x >> 27) * 0x94D049BB133111EB;
 * x ^= x >> 31.  p must be in [0, 1); p == 0 gives threshold 0. */
int cocodr_dropout_mask_for(double p, unsigned long long seed, unsigned long long call, int layer, int kind,
                            cocodr_dropout_mask* out);

/* ------------------------------------------------------------------------------------------
 * GEMM  (hf: nn.Linear in BertSelfAttention :111-203, BertSelfOutput :282-293,
 *        BertIntermediate/BertOutput :325-351, and their autograd backward)
 *
 *   C[z] = epilogue( opA(A[z]) * opB(B[z]) ),  z = 0..batch-1, fp32 accumulation on MFMA.
 *   trans_a = 0: A is [M,K] row-major (lda)        trans_a = 1: A is stored [K,M] (lda)
 *   trans_b = 0: B is [N,K] row-major (ldb) i.e. a torch Linear weight; trans_b = 1: B is [K,N]
 *   forward   Y = X W^T      -> (0,0)      dgrad dX = dY W -> (0,1)     wgrad dW = dY^T X -> (1,1)
 * Epilogues (COCODR_EPI_*): bias add, bias+exact-erf GELU (also emits GELU'(pre-activation), bf16, the only
 * thing the backward needs from the pre-activation), residual add, and the element-wise multiply by that saved
 * derivative for the FFN backward.
 * Requirements: N % 128 == 0; K % 8 == 0; M % 8 == 0 when trans_a; all leading dims % 8 == 0.
 * ------------------------------------------------------------------------------------------ */
enum {
  COCODR_EPI_NONE = 0,      /* C = acc (+bias)                                  */
  COCODR_EPI_GELU = 1,      /* u = acc+bias: C = gelu(u), C2 = gelu'(u)         */
  COCODR_EPI_ADD = 2,       /* C = acc (+bias) + R                              */
  COCODR_EPI_DGELU = 3      /* C = acc * R          (R = the C2 of EPI_GELU)    */
};
typedef struct {
  const uint16_t* A;
  const uint16_t* B;
  void* C;            /* bf16 [M,N] (out_f32 = 0) or fp32 [M,N] (out_f32 = 1) */
  uint16_t* C2;       /* bf16 [M,N], EPI_GELU only; NULL = do not emit the derivative (inference) */
  const float* bias;  /* fp32 [N] or NULL */
  const uint16_t* R;  /* bf16 [M,N], EPI_ADD / EPI_DGELU */
  int M, N, K;
  int lda, ldb, ldc, ldr;
  int trans_a, trans_b, epi, out_f32;
  int batch;
  long long strideA, strideB, strideC, strideR, strideBias; /* elements, per batch index */
  /* optional (batch == 1): colsum[n] = sum_m C[m][n] of the fp32 epilogue result, i.e. the bias gradient of a Linear
   * whose output gradient this GEMM produces; colsum_partial is a workspace of cocodr_gemm_colsum_partial_floats(M,N) */
  float* colsum;
  float* colsum_partial;
  /* EPI_ADD, batch == 1 only: C = dropout(acc + bias) + R with flat index m * N + n (threshold 0 = off) */
  cocodr_dropout_mask drop;
  /* non-zero: A and B hold IEEE half instead of bfloat16 (plain NT form with an fp32 result only: trans_a = trans_b = 0,
   * out_f32 = 1, no epilogue / bias / column sums, N % 256 == 0, K % 64 == 0) - the split-precision score GEMM of the search */
  int ab_f16;
  /* optional workspace (16-byte aligned, batch == 1): lets a launch on the 256 x 256-tile pipeline cut the tiles of its last
   * partial round of the compute units into contraction slices (fp32 partial tiles; a second small kernel adds them in a fixed
   * order and applies the epilogue) instead of running one more nearly empty round of whole tiles - what keeps arbitrary row
   * counts (packed batches) on that pipeline.  cocodr_gemm_split_workspace_floats() floats serve any call; NULL: whole tiles. */
  float* split_ws;
  size_t split_ws_floats;
} cocodr_gemm_args;
size_t cocodr_gemm_split_workspace_floats(void);
size_t cocodr_gemm_colsum_partial_floats(int M, int N);
/* Deferred form: with colsum == NULL and colsum_partial != NULL the call only leaves its per-row-panel sums
 * [rows][N] in colsum_partial (rows = cocodr_gemm_colsum_rows(args), which is 0 when the call would not run on a
 * pipeline with fused sums - then the deferred form is refused); the caller adds the panels later, e.g. for
 * all layers of a backward range in one launch. */
int cocodr_gemm_colsum_rows(const cocodr_gemm_args* args);
int cocodr_gemm(const cocodr_gemm_args* args, cocodr_stream_t stream);
/* n (1..4) independent problems, same result as n cocodr_gemm calls.  Weight-gradient problems (trans_a = trans_b = 1, fp32
 * result, no epilogue, one contraction length) run as ONE launch when together they fill the chip: the four weight matrices of a
 * layer range (dW = dY^T X per nn.Linear, hf BertLayer's backward) have different shapes, so they cannot be batch items of one
 * problem, and launched one after the other each pays its own partial last round of the 256 CUs.
 * workspace (optional, cocodr_gemm_multi_workspace_floats() floats, 16-byte aligned): lets the merged launch cut the tiles of its
 * last partial round of the 256 CUs into contraction slices (fp32 partial tiles, added in a fixed order by a second small
 * kernel) instead of running one more nearly empty round of whole tiles; NULL / too small: whole tiles only. */
size_t cocodr_gemm_multi_workspace_floats(void);
/* what THIS call would use of it (0: the problems do not run merged, or their tiles leave no last round to cut); only the
 * shapes of `problems` are read, so an arena layout can ask before any buffer exists */
size_t cocodr_gemm_multi_workspace_floats_for(const cocodr_gemm_args* problems, int n);
int cocodr_gemm_multi(const cocodr_gemm_args* problems, int n, float* workspace, size_t workspace_floats, cocodr_stream_t stream);
/* tuning / test hook: 0 = auto, 1 = register-staged pipeline,
 * direct-to-LDS <BM,BK,wave rows/32>: 2 = <128,64,2>, 3 = <256,64,2>, 4 = <128,32,2>, 5 = <256,32,2>, 6 = <256,32,4>, 7 = <256,64,4>, 8 = 128x192 tile,
 * 9 / 10 = 3 / 7 with four dedicated loader waves per workgroup, 11 = 256x256 tile, 12 = 256x96 tile (4x3 MFMA waves of
 * 64x32 + four loader waves; N % 96 == 0, makes the N = 768 GEMMs of 8192 tokens exactly one tile per CU)
 *  * 13 = ping-pong pipeline (256x256 tile, eight waves), 18 = the same with two fat phases per K-tile,
 * 14 = hand-scheduled one-wave-per-SIMD kernel (csrc/gemm_a4.hip: NT form, K % 128 == 0), 15 = the same as a persistent tile walk
 * with a register epilogue
 * (also settable through the COCODR_GEMM_IMPL environment variable) */
int cocodr_gemm_set_impl(int impl);

/* ------------------------------------------------------------------------------------------
 * Fused self-attention  (hf: eager_attention_forward :111-203 - softmax(QK^T/sqrt(d) + mask) V,
 * key-padding mask only, head_dim = 64).  qkv is the fused projection output [B*L, 3H]
 * (Q | K | V along columns), mask is int32 [B,L] (non-zero = attend).  L % 32 == 0, L <= 512.
 * lse [B, heads, L] fp32 receives the row log-sum-exp needed by the backward.
 * ------------------------------------------------------------------------------------------ */
int cocodr_attn_fwd(const uint16_t* qkv, const int32_t* mask, uint16_t* ctx, float* lse,
                    int B, int L, int heads, cocodr_stream_t stream);
/* backward: dqkv [B*L, 3H] (dQ | dK | dV) from dctx; L <= 512 (one kernel with all four [L,64] tiles of a
 * (batch, head) in LDS up to L = 256, a dQ kernel + a dK/dV kernel above that).
 * qk_bias_partial (or NULL): [4 B, 2H] fp32 partial column sums of dQ | dK from the fp32 accumulators (rows 4b .. 4b+3
 * together cover the L rows of sequence b; one row per wave of the workgroup) - summed over all 4 B rows they are the
 * query / key bias gradients (hf BertSelfAttention's nn.Linear biases), so the 2H dQ | dK columns need not be read
 * again for them.  The key half is exact zeros: the rows of dS sum to zero, so the key-bias gradient vanishes identically
 * (the fp32 reference produces rounding noise ~1e-7 of the other gradients there). */
int cocodr_attn_bwd(const uint16_t* qkv, const int32_t* mask, const uint16_t* ctx, const uint16_t* dctx,
                    const float* lse, uint16_t* dqkv, float* qk_bias_partial, int B, int L, int heads,
                    cocodr_stream_t stream);
/* The same pair with dropout on the attention probabilities: ctx = (dropout(softmax(..)) V); lse stays the log-sum-exp of
 * the un-dropped scores.  drop == NULL or threshold 0: identical to the calls above. */
int cocodr_attn_fwd_drop(const uint16_t* qkv, const int32_t* mask, uint16_t* ctx, float* lse, int B, int L, int heads,
                         const cocodr_dropout_mask* drop, cocodr_stream_t stream);
int cocodr_attn_bwd_drop(const uint16_t* qkv, const int32_t* mask, const uint16_t* ctx, const uint16_t* dctx,
                         const float* lse, uint16_t* dqkv, float* qk_bias_partial, int B, int L, int heads,
                         const cocodr_dropout_mask* drop, cocodr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Row kernels (HBM bound)
 * embed_ln: hf BertEmbeddings.forward :68-108 - LN(word[ids] + pos[0..L-1] + type[0]), eps 1e-12;
 *           the reference never passes token_type_ids / position_ids (COCO/data.py:140,
 *           ANCE/model/models.py:226-227).
 * ln:       the LayerNorm of BertSelfOutput / BertOutput; input already holds dense+bias+residual.
 * ------------------------------------------------------------------------------------------ */
int cocodr_embed_ln_fwd(const int32_t* ids, const float* word, const float* pos, const float* type0,
                        const float* gamma, const float* beta, uint16_t* out, float* mean, float* rstd,
                        int B, int L, int H, int vocab, float eps, cocodr_stream_t stream);
/* dword must be zeroed by the caller (sparse scatter-add); dpos rows [0,L) are overwritten;
 * partial: fp32 workspace of cocodr_embed_bwd_partial_floats(L,H) floats */
size_t cocodr_embed_bwd_partial_floats(int L, int H);
int cocodr_embed_ln_bwd(const uint16_t* dout, const int32_t* ids, const float* word, const float* pos,
                        const float* type0, const float* gamma, const float* mean, const float* rstd,
                        float* dword, float* dpos, float* dtype0, float* dgamma, float* dbeta,
                        float* partial, int B, int L, int H, int vocab, cocodr_stream_t stream);
/* The embedding pair with dropout on the LayerNorm output (hf BertEmbeddings.forward: dropout(LayerNorm(..))); the
 * backward masks and scales dout before the LayerNorm backward.  drop == NULL or threshold 0: the calls above. */
int cocodr_embed_ln_fwd_drop(const int32_t* ids, const float* word, const float* pos, const float* type0,
                             const float* gamma, const float* beta, uint16_t* out, float* mean, float* rstd,
                             int B, int L, int H, int vocab, float eps, const cocodr_dropout_mask* drop,
                             cocodr_stream_t stream);
int cocodr_embed_ln_bwd_drop(const uint16_t* dout, const int32_t* ids, const float* word, const float* pos,
                             const float* type0, const float* gamma, const float* mean, const float* rstd,
                             float* dword, float* dpos, float* dtype0, float* dgamma, float* dbeta,
                             float* partial, int B, int L, int H, int vocab, const cocodr_dropout_mask* drop,
                             cocodr_stream_t stream);
int cocodr_ln_fwd(const uint16_t* y, const float* gamma, const float* beta, uint16_t* out, float* mean,
                  float* rstd, float* cls_out /* fp32 [M/cls_stride, H] or NULL */, int cls_stride,
                  int M, int H, float eps, cocodr_stream_t stream);
/* dy_colsum (fp32 [H] or NULL) receives the column sums of dy: the bias gradient of the Linear whose output
 * (+ residual) this LayerNorm normalises, so the backward needs no separate pass over dy for it */
size_t cocodr_ln_bwd_partial_floats(int M, int H);
int cocodr_ln_bwd(const uint16_t* dout, const uint16_t* y, const float* gamma, const float* mean,
                  const float* rstd, uint16_t* dy, float* dgamma, float* dbeta, float* dy_colsum, float* partial,
                  int M, int H, cocodr_stream_t stream);
/* LayerNorm(dropout(dense) + residual): dy stays the gradient w.r.t. the LayerNorm input (= the residual branch's),
 * dy_drop [M,H] receives dy masked and scaled (= the gradient w.r.t. the dense output, what the dgrad / wgrad GEMMs of
 * that Linear take), and dy_colsum sums dy_drop (its bias gradient).  drop == NULL or threshold 0: dy_drop is not
 * written and the call is cocodr_ln_bwd. */
int cocodr_ln_bwd_drop(const uint16_t* dout, const uint16_t* y, const float* gamma, const float* mean,
                       const float* rstd, uint16_t* dy, uint16_t* dy_drop, float* dgamma, float* dbeta,
                       float* dy_colsum, float* partial, int M, int H, const cocodr_dropout_mask* drop,
                       cocodr_stream_t stream);
/* column sums of a bf16 [M,N] matrix (bias gradients), batched: out[z][n] = sum_m X[z][m][n] */
size_t cocodr_colsum_partial_floats(int M, int N, int batch);
int cocodr_colsum(const uint16_t* X, float* out, float* partial, int M, int N, int ldx, int batch,
                  long long strideX, long long strideOut, cocodr_stream_t stream);
int cocodr_cast_f32_bf16(const float* src, uint16_t* dst, size_t n, cocodr_stream_t stream);
/* dst[0 .. n) = 0 (fp32, 16-byte aligned): the embedding tables' gradient block, which the embedding backward accumulates into
 * with atomics (what autograd's zero-initialised sparse-to-dense embedding gradient is in the reference, hf BertEmbeddings) */
int cocodr_zero_f32(float* dst, size_t n, cocodr_stream_t stream);
/* Row plumbing of the label-sparse MLM head (COCO/modeling.py:87-93, 222-224 run lm.cls on all B L rows and let the cross entropy
 * ignore -100; here only the labelled rows go through it): dst[r] = src[idx[r]] (bf16 rows of width H, idx int64 [n]);
 * scatter: dst[idx[r]] = src[r] (bf16) or, add_f32 != 0, dst fp32 [.,H] row idx[r] += src[r] (idx unique); and the element-wise
 * bf16 product of the transform's backward (dense-output gradient = LayerNorm-input gradient x saved GELU'). */
int cocodr_gather_rows(const uint16_t* src, const long long* idx, uint16_t* dst, int n, int H, cocodr_stream_t stream);
int cocodr_scatter_rows(const uint16_t* src, const long long* idx, void* dst, int n, int H, int add_f32, cocodr_stream_t stream);
int cocodr_mul_bf16(const uint16_t* a, const uint16_t* b, uint16_t* out, size_t n, cocodr_stream_t stream);
/* idx[b] = first row of sequence b: seq_off[b] (packed batches; seq_off int32 [B+1] on the device) or b * L (seq_off == NULL) */
int cocodr_cls_rows(const int32_t* seq_off, int L, int B, long long* idx, cocodr_stream_t stream);
/* d_last[b*L + 0, :] = bf16(dE[b, :]), all other rows zero (gradient enters at [CLS] only) */
int cocodr_scatter_cls_grad(const float* dE, uint16_t* d_last, int B, int L, int H, cocodr_stream_t stream);

/* Fused AdamW over one flat fp32 parameter (torch.optim.AdamW semantics; the reference steps AdamW through the HF
 * Trainer, COCO/trainer.py:66-70, or its own loop, ANCE/drivers/run_ann.py:345-356).  g is multiplied by grad_scale
 * (and by *grad_scale_dev when that device pointer is not NULL - the clip coefficient below) first.  When
 * shadow != NULL the updated values of elements [shadow_begin, n) are also written as bf16 to
 * shadow[0 .. n - shadow_begin) - the weight-matrix shadow the GEMMs read - in the same pass. */
int cocodr_adamw_step(float* p, const float* g, float* m, float* v, uint16_t* shadow, size_t shadow_begin, size_t n,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                      const float* grad_scale_dev, cocodr_stream_t stream);

/* torch.nn.utils.clip_grad_norm_ (ANCE/drivers/run_ann.py:347-352) without a host round trip: the total L2 norm of
 * up to 8 gradient tensors goes to out[0] and the coefficient min(1, max_norm / (norm + 1e-6)) to out[1] (device
 * memory); the optimizer passes multiply the gradient by it (grad_scale_dev).  grads / numels are host arrays;
 * partial: device workspace of count * 1024 floats. */
int cocodr_grad_norm_clip(const float* const* grads, const size_t* numels, int count, float max_norm, float* partial,
                          float* out, cocodr_stream_t stream);

/* LAMB as the reference implements it (ANCE/utils/lamb.py:61-121; ANCE's default optimizer, run_ann.py:128-133):
 * m, v without bias correction, u = m / (sqrt(v) + eps) + weight_decay * w, and per parameter TENSOR
 * w -= lr * [clamp(||w||, 0, 10) / ||u||] * u (ratio 1 when either norm is 0).  The tensors live inside one flat
 * parameter; `plan` (device arrays, built once by the host) cuts it into chunks of at most a few thousand elements
 * that never straddle a tensor: chunk c = elements [chunk_start[c], chunk_start[c] + chunk_len[c]) (multiples of 4) of
 * tensor chunk_seg[c]; the chunks of tensor s are seg_chunk_begin[s] .. seg_chunk_begin[s+1]-1.
 * workspace: 2 * nchunk + nseg floats; stats (NULL or [nseg][2]) receives (weight_norm, adam_norm) per tensor, the
 * quantities the reference logs (lamb.py:12-22).  The reference loops over ~200 tensors with ~10 torch kernels
 * each; this is three launches per flat. */
typedef struct {
  const long long* chunk_start;
  const int* chunk_len;
  const int* chunk_seg;
  const int* seg_chunk_begin;
  int nchunk, nseg;
} cocodr_lamb_plan;
int cocodr_lamb_step(float* p, const float* g, float* m, float* v, uint16_t* shadow, size_t shadow_begin, size_t n,
                     const cocodr_lamb_plan* plan, float lr, float beta1, float beta2, float eps, float weight_decay,
                     float grad_scale, const float* grad_scale_dev, float* workspace, float* stats,
                     cocodr_stream_t stream);
/* The same update in ONE pass (30 instead of 42 B / parameter) for the tensors of a flat parameter that fit the chip's register
 * files - the encoder's weight matrices: a persistent grid of co-resident workgroups spreads each tensor over all of them, keeps w
 * and u in registers between the norm and the update, and exchanges per-workgroup partial norms through `workspace` (added in a
 * fixed order: deterministic; same arithmetic per element as cocodr_lamb_step).  plan: entry k = one tensor = elements [seg_start[k],
 * seg_start[k] + seg_len[k]) (multiples of 4), seg_index[k] = its row in trust / stats (the tensor numbering of the cocodr_lamb_plan
 * the rest of the flat goes through: give those tensors NO chunks there).  The kernel runs in ROUNDS: round r holds the entries
 * round_first[r] .. round_first[r + 1] - 1, and entry k is worked on by the workgroups wg_begin[k] .. wg_begin[k] + wg_count[k] - 1 of
 * the G = cocodr_lamb_fused_workgroups() (disjoint ranges inside [0, G) within a round; seg_len[k] <= wg_count[k] *
 * cocodr_lamb_fused_workgroup_elements()): several small tensors share a round, a tensor of cocodr_lamb_fused_capacity() elements
 * (= G x that; 0 where the kernel cannot run) has one to itself.  workspace: cocodr_lamb_fused_workspace_floats(nfused) floats, ZEROED
 * once by the caller and then left to this function (it holds the tagged partial-norm granules of the previous calls); the int at
 * float index cocodr_lamb_fused_error_index(nfused) is set to 1 if a workgroup gave up waiting for the others (never, unless the
 * device cannot hold the grid; the numbers of that step are then wrong).  trust fp32 [>= max seg_index + 1]. */
typedef struct {
  const long long* seg_start;
  const int* seg_len;
  const int* seg_index;
  const int* wg_begin;
  const int* wg_count;
  const int* round_first; /* [nrounds + 1] */
  int nfused, nrounds;
} cocodr_lamb_fused_plan;
size_t cocodr_lamb_fused_capacity(void);
int cocodr_lamb_fused_workgroups(void);
size_t cocodr_lamb_fused_workgroup_elements(void);
size_t cocodr_lamb_fused_workspace_floats(int nfused);
size_t cocodr_lamb_fused_error_index(int nfused);
int cocodr_lamb_step_fused(float* p, const float* g, float* m, float* v, uint16_t* shadow, size_t shadow_begin,
                           const cocodr_lamb_fused_plan* plan, float lr, float beta1, float beta2, float eps, float weight_decay,
                           float grad_scale, const float* grad_scale_dev, float* workspace, float* trust, float* stats,
                           cocodr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Losses
 * simce: COCO/modeling.py:244-248 compute_contrastive_loss + :172-177 co_target + the `.mean()`
 *        at :229, and its gradient w.r.t. the LOCAL rows [row0, row0+m_local) of the gathered
 *        E [M,H] (COCO/modeling.py:182-186 keeps autograd history only in the local slot).
 *        loss_rows [M] fp32, loss [1] fp32, dE_local [m_local,H] fp32.
 *        workspace: cocodr_simce_workspace_floats(M) floats.
 * triplet: ANCE/model/models.py:97-106 + :260-261 - logits=[q.a, q.b], -log_softmax[:,0],
 *        (loss*weights).mean(); gradients w.r.t. q, a, b.
 * ------------------------------------------------------------------------------------------ */
size_t cocodr_simce_workspace_floats(int M);
int cocodr_simce_fwd_bwd(const float* E, int M, int H, int world, int row0, int m_local, float* loss_rows,
                         float* loss, float* dE_local, float* workspace, cocodr_stream_t stream);
int cocodr_triplet_nll_fwd_bwd(const float* q, const float* a, const float* b, const float* weights /* or NULL */,
                               int B, int H, float* loss_rows, float* logits /* [B,2] */, float* loss,
                               float* dq, float* da, float* db, cocodr_stream_t stream);

/* COCO's negative gather (COCO/modeling.py:182-190: all_gather of the [2b, H] [CLS] block; the reference builds a list of W
 * tensors, overwrites its own slot and concatenates): ONE RCCL all-gather of `rows` x H fp32 rows per rank into the contiguous
 * [W rows, H] matrix cocodr_simce_fwd_bwd takes, enqueued on `stream`.  nccl_comm is the caller's ncclComm_t (rank r's rows land
 * at gathered + r rows H; no gradient exchange is needed afterwards: simce_fwd_bwd returns the gradient of the local rows).
 * RCCL is resolved at run time from the copy the process has loaded (e.g. torch's) - the library itself does not link it.  The
 * Python host reaches the same collective through torch.distributed.all_gather_into_tensor on its process group. */
int cocodr_allgather_rows(const float* local_rows, float* gathered, int rows, int H, void* nccl_comm, cocodr_stream_t stream);

/* Masked-LM cross entropy over the vocabulary (hf BertForMaskedLM loss / COCO/modeling.py:87-93 mlm_loss):
 * logits fp32 [n, ld] (columns >= V are padding), labels int32 [n] in [0,V); row_scale fp32 [n] carries the
 * 1/(number of labelled rows of the row's group) factor of the mean.  loss_rows[i] = lse_i - logit_i[label_i];
 * dlogits bf16 [n, ld] = row_scale_i * (softmax_i - onehot_i), zero in the padding columns. */
int cocodr_ce_fwd_bwd(const float* logits, const int32_t* labels, const float* row_scale, int n, int V, int ld,
                      float* loss_rows, uint16_t* dlogits, cocodr_stream_t stream);

/* out [G,G] = A A^T for a short, very wide fp32 matrix A [G, D] (row stride lda), G <= 64: iDRO's gram of the per-group
 * gradients (ANCE/model/dro_loss.py:236-238 `all_grads @ all_grads.T`, D = the parameters of BertLayers 9-11).  One pass
 * over A per 8-row block pair, deterministic; workspace: cocodr_gram_f32_workspace_floats(G, D) floats. */
size_t cocodr_gram_f32_workspace_floats(int G, long long D);
int cocodr_gram_f32(const float* A, long long lda, int G, long long D, float* out, float* workspace, cocodr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Brute-force inner-product search: faiss.IndexFlatIP(dim).add(P); .search(Q,k)
 * (evaluate/evaluation/evaluate_beir.py:220-224, ANCE/drivers/run_ann_data_gen.py:310-317,390,
 *  ANCE/utils/eval_mrr.py:81-90).  Q [Nq,H] fp32, P [Np,H] fp32; D [Nq,k] fp32 descending,
 * I [Nq,k] int64 positions into P (+ id_offset), ties -> lower position first, (-inf,-1) padding.
 * ------------------------------------------------------------------------------------------ */
/* Three score pipelines, same selection:
 *  - split precision (default): Q and P are scaled by a power of two and split into two IEEE halves each (x = xh + xl to 22
 *    bits); ql.ph + qh.pl + qh.ph run as ONE half-precision MFMA GEMM of depth 3H with fp32 accumulation (small terms
 *    first).  Every partial product is exact in fp32; the result is closer to the real q.p than a sequential fp32 dot
 *    product is (measured max error 0.5e-6 of the largest score against 1.1e-6), identical passages still give
 *    bit-identical scores, and integer-valued inputs below 2^11 stay exact.  3/16 of the fp32 matrix pipe's time per score.
 *  - exact fp32 MFMA (cocodr_score_set_mode(1) or COCODR_SCORE_EXACT=1, and whenever the workspace is too small for the
 *    half operands): scores bit-identical to an fmaf chain over the contraction index.
 *  - half-precision scores (cocodr_score_set_mode(2), opt-in, never selected automatically): ONE product of the operands rounded
 *    to IEEE half (11 significant bits, after the same power-of-two scaling) with fp32 accumulation - the arithmetic of a faiss
 *    fp16 flat index; a third of the default pipeline's matrix work.  Score error ~1e-5 of |q||p| on embedding-shaped data
 *    (worst case 1e-3): rankings differ from the fp32 ones only between near-ties; nDCG@10 / recall@1000 within 1e-3 of the exact
 *    search on the config-5 workload (tests/test_gpu_retrieval.py).  Identical passages still score bit-identically.
 * Embeddings with non-finite components get NaN scores on the split path (ranked last).
 * Selection.  Small searches score a chunk of query rows into an fp32 slab and select from it (radix select per row).  On the
 * 16-bit pipelines a search over >= 32 768 passages with Nq Np >= 100 Mi and 16 k <= Np is FILTERED instead
 * (cocodr_score_filter_plan tells; the window is where it was measured to win, tools/search_crossover.py): every
 * query row gets a threshold - the j-th best of its scores against a strided sample of ~Np / 32 passages, j a few deviations past
 * the sample's share of the top k - the score GEMM's epilogue keeps only the scores at or above it (~2 k of a row) in fixed
 * candidate blocks, and the k best are selected from those.  No [Nq, Np] score slab is written or read.  The result is the
 * exhaustive search's, bit for bit and tie for tie: a row is answered from its candidates only if they number >= k and none of
 * its blocks overflowed (then they provably contain its k best); any other row - thresholds that came out too high, heavy ties,
 * clustered duplicates - is scored and selected exhaustively by a device-gated second pass.  COCODR_SCORE_NOFILTER=1: never filter.
 * workspace_bytes_dim: for embeddings of width H; workspace_bytes: the same for H = 1024 (enough for any H <= 1024).
 * The workspace must be 256-byte aligned. */
size_t cocodr_score_topk_workspace_bytes_dim(int Nq, int Np, int H, int k);
/* The filtered search's plan for these sizes in the current score mode: out[0] = 1 if cocodr_score_topk filters (given the
 * workspace cocodr_score_topk_workspace_bytes_dim asks for), out[1] = sampled passages, [2] = their stride, [3] = j, [4] = slots per
 * (row, 256-passage tile) block incl. its header, [5] = query rows per pass, [6] = rows per exhaustive pass, [7] = byte offset in
 * the workspace of the int32 count of rows the LAST pass handed back to the exhaustive pass (diagnostics, tests). */
int cocodr_score_filter_plan(int Nq, int Np, int H, int k, long long out[8]);
size_t cocodr_score_topk_workspace_bytes(int Nq, int Np, int k);
int cocodr_score_set_mode(int mode);
int cocodr_score_topk(const float* Q, const float* P, int Nq, int Np, int H, int k, long long id_offset,
                      float* D, long long* I, void* workspace, size_t workspace_bytes, cocodr_stream_t stream);
/* faiss.IndexFlatIP: add(P) once, search(Q, k) many times (ANCE/drivers/run_ann_data_gen.py:310-317,390;
 * evaluate/evaluation/evaluate_beir.py:220-224).  The same search with p_resident != 0: the caller vouches that `workspace` still
 * holds what the previous cocodr_score_topk / cocodr_score_topk_resident call with the SAME (P, Np, H, Nq, k, score mode) left in it
 * - the passages' scale, their half-precision split image and the filter's passage sample - and none of it is rebuilt (three
 * launches and a pass over P less per search).  p_resident == 0 is cocodr_score_topk.  (Host side: cocodr_amd.retrieval.FlatIPIndex.) */
int cocodr_score_topk_resident(const float* Q, const float* P, int Nq, int Np, int H, int k, long long id_offset, float* D,
                               long long* I, void* workspace, size_t workspace_bytes, int p_resident, cocodr_stream_t stream);

/* k-way merge of per-shard top-k lists into the list ONE search over the rank-major merged corpus would return - what
 * replaces the reference's gather-everything-then-search (ANCE/utils/util.py:117-155 barrier_array_merge +
 * evaluate/evaluation/evaluate_beir.py:200-224 IndexFlatIP over the concatenated shards) when the corpus stays sharded in
 * HBM (each rank merges its Nq / W query block).
 * D fp32 / I int32 [W][Nq][k] (shard w's lists start at w * stride_w elements): for every query W lists sorted by
 * (score descending, position ascending) - cocodr_score_topk's output with id_offset 0, positions LOCAL to the shard and
 * narrowed to int32, empty slots (I < 0) at the end.  shard_offset int64 [W] (device): first global position of shard w,
 * ascending in w.  outD fp32 / outI int64 [Nq][k_out], k_out <= W * k: (score descending, global position ascending),
 * (-inf, -1) padding.  W <= 64, W * k <= 39936.  Deterministic, no workspace.  A slot is empty iff its position is negative - a
 * candidate whose score is -inf is a candidate; NaN scores are not supported (they have no place in the order). */
int cocodr_topk_merge(const float* D, const int32_t* I, const long long* shard_offset, int W, int Nq, int k,
                      long long stride_w, float* outD, long long* outI, int k_out, cocodr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Whole-encoder entry points: the layer loop lives in native code so one host call enqueues every
 * kernel of BertModel.forward (COCO/modeling.py:199-204, ANCE/model/models.py:225-229) or of its
 * backward.  Parameters are borrowed (torch owns them); activations live in a caller arena.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int hidden, heads, layers, inter, vocab, max_pos;
  float ln_eps;
  /* dropout of a training forward (training != 0) and of its backward: hf hidden_dropout_prob /
   * attention_probs_dropout_prob, and the (seed, call) pair the site keys derive from (cocodr_dropout_mask_for).  The
   * backward of a forward must be given the same four values; 0 probabilities = no dropout (eval). */
  float hidden_dropout, attn_dropout;
  unsigned long long drop_seed, drop_call;
  /* cls_tail != 0: the caller consumes the last layer's [CLS] rows only (every reference wrapper of the contrastive / ANCE /
   * inference paths: COCO/modeling.py:199-204 hidden_states[-1][:, :1], ANCE/model/models.py:225-232 [0][:, 0]).  The last layer
   * then computes QKV and the attention over all rows (its keys and values come from every token) but the attention output
   * projection, both LayerNorms and the FFN on the B [CLS] rows alone; cls_f32 is identical, hidden_states[layers] is NOT
   * produced (its arena slot is scratch), and the backward of such a forward takes the [B,H] bf16 gradient of the [CLS] rows as
   * d_last and must be given the same flag.  Not available with dropout (the masks are indexed by token row); a training
   * forward needs B % 8 == 0. */
  int cls_tail;
} cocodr_config;

typedef struct { /* one BertLayer; w* are bf16 shadows [out,in], vectors are the fp32 masters */
  const uint16_t* wqkv; /* [3H,H]  query|key|value rows */
  const uint16_t* wo;   /* [H,H]   attention.output.dense */
  const uint16_t* w1;   /* [I,H]   intermediate.dense */
  const uint16_t* w2;   /* [H,I]   output.dense */
  const float *bqkv, *bo, *b1, *b2, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
} cocodr_layer_params;

typedef struct { /* fp32 gradient destinations, same shapes as the masters */
  float *wqkv, *wo, *w1, *w2, *bqkv, *bo, *b1, *b2, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
} cocodr_layer_grads;

typedef struct {
  const float *word, *pos, *type0, *ln_g, *ln_b; /* fp32 masters */
} cocodr_embed_params;
typedef struct {
  float *word, *pos, *type0, *ln_g, *ln_b;
} cocodr_embed_grads;

typedef struct { /* byte offsets into the arena, filled by cocodr_encoder_layout */
  size_t total_bytes;
  size_t hidden;      /* bf16 [layers+1][M,H] - hidden_states tuple (output_hidden_states=True) */
  size_t cls_f32;     /* fp32 [B,H]  last-layer [CLS] rows */
  size_t qkv, ctx, y1, x1, u, h, y2;        /* bf16 per-layer activations, [layers][M,*] */
  size_t lse, mean1, rstd1, mean2, rstd2;    /* fp32 per-layer statistics */
  size_t emb_mean, emb_rstd;
  size_t bwd_scratch;  /* backward-only region (dgrad chain, saved dY for the grouped wgrad) */
  size_t bwd_bytes;
  size_t bwd_dx;       /* bf16 [M,H]: where a backward range leaves dL/d(hidden_states[layer_lo]) */
  size_t split_ws;     /* fp32 workspace the forward / dgrad GEMMs may cut their last partial round into (cocodr_gemm_args.split_ws) */
  size_t split_ws_floats; /* 0: no GEMM of this shape has more tiles than compute units */
} cocodr_encoder_layout_t;

/* training = 0 keeps only what inference needs (hidden states + one layer of scratch) */
int cocodr_encoder_layout(const cocodr_config* cfg, int B, int L, int training, cocodr_encoder_layout_t* out);

/* Where a backward range leaves what it computed besides the parameter gradients (byte offsets from the arena base,
 * training arenas only): the output gradients of the four Linears of every layer in the range (bf16, [layers][M, .],
 * indexed by ABSOLUTE layer: dy2 = d(FFN output + residual), du = d(FFN pre-activation), dy1 = d(attention output +
 * residual), dqkv) and - for ranges of more than one layer - the LayerNorm-backward partial rows of the range
 * (fp32 [layer - layer_lo][ln_blocks][3][H] = dgamma, dbeta, column sums of dy, over ln_rows consecutive tokens each).
 * Used by hosts that need per-sequence gradients (iDRO's per-group gradients, coco-dr_amd/idro.py). */
typedef struct {
  size_t dy2, du, dy1, dqkv;
  size_t ln2_partial, ln1_partial;
  int ln_blocks, ln_rows;
} cocodr_encoder_bwd_layout_t;
int cocodr_encoder_bwd_layout(const cocodr_config* cfg, int B, int L, cocodr_encoder_bwd_layout_t* out);

int cocodr_encoder_fwd(const cocodr_config* cfg, const cocodr_embed_params* emb,
                       const cocodr_layer_params* layers_host, const int32_t* ids, const int32_t* mask,
                       int B, int L, int training, void* arena, size_t arena_bytes, cocodr_stream_t stream);

/* The same forward for the layers [layer_lo, layer_hi) only (the embedding runs with the range that starts at 0), for
 * hosts that interleave other work or waits between parts of the stack; consecutive ranges over one arena leave exactly
 * what the single call leaves. */
int cocodr_encoder_fwd_range(const cocodr_config* cfg, const cocodr_embed_params* emb, const cocodr_layer_params* layers_host,
                             const int32_t* ids, const int32_t* mask, int B, int L, int training, void* arena,
                             size_t arena_bytes, int layer_lo, int layer_hi, cocodr_stream_t stream);

/* A bare stack of BertLayers (no embeddings): the Condenser head of COCO/modeling.py:43-46,73-79,212-220
 * (`c_head`, n_head_layers BertLayers applied to cat(cls, hidden_states[skip_from][:,1:])).  Same arena layout as
 * the encoder with cfg->layers = number of stacked layers; the caller fills hidden slot 0 ([M,H] bf16 at
 * layout.hidden) before the call.  Its backward is cocodr_encoder_bwd_range(..., layer_lo = 0, do_embed = 0)
 * (emb / emb_grads / ids may then be NULL); the input gradient is left at layout.bwd_dx. */
int cocodr_stack_fwd(const cocodr_config* cfg, const cocodr_layer_params* layers_host, const int32_t* mask, int B, int L,
                     int training, void* arena, size_t arena_bytes, cocodr_stream_t stream);

/* d_last: bf16 [M,H] gradient of the loss w.r.t. hidden_states[-1].  Layer gradient blocks must be
 * laid out with a uniform stride between consecutive layers (grads_host[l+1].x - grads_host[l].x
 * constant) so the weight gradients of all layers are computed by one grouped launch per matrix.
 * All gradient destinations are OVERWRITTEN except emb_grads->word which is accumulated into
 * (caller zeroes it). */
int cocodr_encoder_bwd(const cocodr_config* cfg, const cocodr_embed_params* emb,
                       const cocodr_layer_params* layers_host, const cocodr_embed_grads* emb_grads,
                       const cocodr_layer_grads* grads_host, const int32_t* ids, const int32_t* mask,
                       const uint16_t* d_last, int B, int L, void* arena, size_t arena_bytes,
                       cocodr_stream_t stream);

/* Same backward, restricted to the layers [layer_lo, layer_hi) (walked top-down) plus - when do_embed != 0 and
 * layer_lo == 0 - the embedding backward.  The top range (layer_hi == layers) takes d_in = dL/d(hidden_states[-1]);
 * continuation ranges pass d_in = NULL and pick up the gradient the previous range left in the arena.  The weight /
 * bias gradients of the range are complete when the call returns (enqueued), so a data-parallel host can start
 * all-reducing them while the next range runs (gradient layout: layer blocks at a uniform stride, see above). */
int cocodr_encoder_bwd_range(const cocodr_config* cfg, const cocodr_embed_params* emb,
                             const cocodr_layer_params* layers_host, const cocodr_embed_grads* emb_grads,
                             const cocodr_layer_grads* grads_host, const int32_t* ids, const int32_t* mask,
                             const uint16_t* d_in, int B, int L, void* arena, size_t arena_bytes, int layer_hi,
                             int layer_lo, int do_embed, cocodr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Packed (variable-length) batches - SURVEY 7 (iii).  The reference pads every sequence of a batch to one length and
 * lets the attention mask hide the padding (COCO/data.py:135-144, ANCE/utils/util.py); on MS MARCO-shaped batches 40 % of
 * the token rows are padding.  Here the B sequences may be stored back to back instead: sequence b owns rows
 * [seq_off[b], seq_off[b+1]) of every [T, .] activation, an extent of ANY number of rows >= max(its length, 1) (mask [T] is 0
 * on the rows of an extent behind the sequence's tokens; cocodr_amd stores every sequence on exactly its length and spreads the
 * <= 31 rows that make T a multiple of 32 over the last sequences), seq_off is int32 [B+1] in device memory, T % 32 == 0,
 * max_len = the longest extent rounded up to a multiple of 32.  Every row kernel and GEMM simply sees T rows; the kernels below
 * are the ones that need the sequence structure (the attention walks an extent in 32-row blocks: rows of the last block that are
 * behind the extent - the next sequence's - are read as zeros, masked, and never written).  lse is [heads, T] on this layout.  drop_L: the padded length the dropout indices of the
 * attention probabilities are defined on ((b, h, q, k) -> ((b heads + h) drop_L + q) drop_L + k), so a packed and a padded
 * run of one batch draw the same masks; hidden-state dropout indexes rows, which differ between the layouts.
 * Results at the real tokens equal the padded path's (same arithmetic per token); padding rows hold other values.
 * ------------------------------------------------------------------------------------------ */
/* seq_order (optional, int32 [B] on the device, a permutation of 0..B-1): the order in which the attention launches hand the
 * sequences to workgroups - longest first, so that the last, partial round over the CUs holds the cheap ones.  NULL: 0..B-1.
 * Results do not depend on it. */
int cocodr_attn_fwd_packed(const uint16_t* qkv, const int32_t* mask, uint16_t* ctx, float* lse, const int32_t* seq_off,
                           const int32_t* seq_order, int B, int T, int max_len, int heads, const cocodr_dropout_mask* drop, int drop_L,
                           cocodr_stream_t stream);
int cocodr_attn_bwd_packed(const uint16_t* qkv, const int32_t* mask, const uint16_t* ctx, const uint16_t* dctx, const float* lse,
                           uint16_t* dqkv, float* qk_bias_partial, const int32_t* seq_off, const int32_t* seq_order, int B, int T,
                           int max_len, int heads, const cocodr_dropout_mask* drop, int drop_L, cocodr_stream_t stream);
/* positions: int32 [T], the position id of every row (0.. within its sequence) */
int cocodr_embed_ln_fwd_packed(const int32_t* ids, const int32_t* positions, const float* word, const float* pos,
                               const float* type0, const float* gamma, const float* beta, uint16_t* out, float* mean,
                               float* rstd, int T, int H, int vocab, float eps, const cocodr_dropout_mask* drop,
                               cocodr_stream_t stream);
size_t cocodr_embed_bwd_packed_partial_floats(int T, int H);
/* one workgroup per position (as on the padded path: the position rows are plain sums over the sequences that reach that
 * position, only the sparse word rows are fp32 atomics); ids int32 [T], seq_off int32 [B+1] */
int cocodr_embed_ln_bwd_packed(const uint16_t* dout, const int32_t* ids, const int32_t* seq_off, const float* word,
                               const float* pos, const float* type0, const float* gamma, const float* mean, const float* rstd,
                               float* dword, float* dpos, float* dtype0, float* dgamma, float* dbeta, float* partial, int B, int T,
                               int max_len, int H, int vocab, const cocodr_dropout_mask* drop, cocodr_stream_t stream);
/* The packed-layout description of a padded batch, built on the device in ONE launch (the collator of the reference pads on
 * the CPU - COCO/data.py:135-144, ANCE/data/msmarco_data.py:381-382 - so a host that keeps the lengths passes them; one that
 * has only the device mask asks cocodr_mask_lengths first and reads 2 B integers back).
 *   cocodr_mask_lengths: mask [B, L] with elem_bytes 1 / 4 / 8 per entry, rows row_stride entries apart -> lens[b] = set
 *     entries of row b, prefix_ok[b] = 1 when they are exactly the first lens[b] (only such batches can be packed).
 *   cocodr_pack_index: ids [B, .] (int32 or int64: elem_bytes 4 / 8; rows row_stride entries apart), lens int32 [B] (<= row
 *     length), seq_off int32 [B+1] = running sum of the extents (each >= max(len, 1)); writes the int32 [T] arrays of
 *     cocodr_packed_batch (token id or 0 on the rows of an extent behind the tokens, position, mask, cls_slot) and, when src != NULL, the row of the
 *     padded [B, L] layout every packed row came from (int64 [T]; L % 32 == 0 is that layout's row length). */
int cocodr_mask_lengths(const void* mask, int elem_bytes, int B, int L, long long row_stride, int32_t* lens, int32_t* prefix_ok,
                        cocodr_stream_t stream);
int cocodr_pack_index(const void* ids, int elem_bytes, long long row_stride, const int32_t* lens, const int32_t* seq_off, int B,
                      int L, int32_t* out_ids, int32_t* positions, int32_t* mask, int32_t* cls_slot, long long* src,
                      cocodr_stream_t stream);
/* The host arithmetic of the layout on the device, for callers that hand over the reference's batch unchanged
 * ({input_ids, attention_mask}, COCO/data.py:150-154; ANCE/data/msmarco_data.py:381-382) and do not know the lengths:
 *   cocodr_pack_plan: lens / prefix_ok int32 [B] (cocodr_mask_lengths' outputs), cap = the padded row length rounded up to 32 ->
 *     plan int32 [3B + 1] = lens [B] | seq_off [B + 1] | seq_order [B] (extent = max(len, 1); the <= 31 rows that make T a
 *     multiple of 32 go to the last sequences with room below cap; order = longest extent first, ties in batch order) and
 *     result int32 [4] = {T, longest extent rounded up to 32, 1 when every mask is a prefix mask, B}.  One workgroup, B <= 4096.
 * A host then reads `result` back (16 bytes, the only thing the launches behind it need from the device: T sizes the GEMMs)
 * while cocodr_pack_index (which takes lens / seq_off from `plan`) is already queued. */
int cocodr_pack_plan(const int32_t* lens, const int32_t* prefix_ok, int B, int cap, int32_t* plan, int32_t* result,
                     cocodr_stream_t stream);
/* cocodr_ln_fwd whose fp32 [CLS] copies are named row by row: cls_slot int32 [M], -1 or the row of cls_out a row goes to */
int cocodr_ln_fwd_slots(const uint16_t* y, const float* gamma, const float* beta, uint16_t* out, float* mean, float* rstd,
                        float* cls_out, int cls_stride, const int32_t* cls_slot, int M, int H, float eps,
                        cocodr_stream_t stream);

/* Whole-encoder calls on a packed batch.  ids / positions / mask / cls_slot are int32 [T] (cls_slot: b on the first row of
 * sequence b, -1 elsewhere); the arena layout is cocodr_encoder_layout_packed's (every [M, .] block has T rows, cls_f32 B
 * rows, lse [heads, T]).  The backward mirrors cocodr_encoder_bwd_range (same ranges, same gradient contract). */
typedef struct {
  const int32_t* ids;
  const int32_t* positions;
  const int32_t* mask;
  const int32_t* seq_off;
  const int32_t* cls_slot;
  int B, T, max_len, drop_L;
  const int32_t* seq_order; /* optional (NULL: 0..B-1): see cocodr_attn_fwd_packed */
} cocodr_packed_batch;
int cocodr_encoder_layout_packed(const cocodr_config* cfg, int T, int B, int training, cocodr_encoder_layout_t* out);
int cocodr_encoder_fwd_packed(const cocodr_config* cfg, const cocodr_embed_params* emb, const cocodr_layer_params* layers_host,
                              const cocodr_packed_batch* batch, int training, void* arena, size_t arena_bytes,
                              cocodr_stream_t stream);
/* cocodr_stack_fwd on a packed batch: a bare BertLayer stack (the Condenser head, COCO/modeling.py:43-46, 216-220) whose
 * input the caller has written to hidden slot 0 of the arena ([T, H]); batch->ids / positions are not read */
int cocodr_stack_fwd_packed(const cocodr_config* cfg, const cocodr_layer_params* layers_host, const cocodr_packed_batch* batch,
                            int training, void* arena, size_t arena_bytes, cocodr_stream_t stream);
int cocodr_encoder_bwd_packed(const cocodr_config* cfg, const cocodr_embed_params* emb, const cocodr_layer_params* layers_host,
                              const cocodr_embed_grads* emb_grads, const cocodr_layer_grads* grads_host,
                              const cocodr_packed_batch* batch, const uint16_t* d_in, void* arena, size_t arena_bytes,
                              int layer_hi, int layer_lo, int do_embed, cocodr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * On-device Condenser / coCondenser collator (COCO/data.py:24-156): per span a random truncation window of
 * max_seq_length - 2 tokens (:101-117), the whole-word-mask proxy (:44-55, 68-99: words = a token plus its "##"
 * continuations, shuffled, taken greedily up to round(len * mlm_probability) tokens), [CLS] .. [SEP] + padding
 * (:135-144) and the 80 % [MASK] / 10 % random / 10 % keep rule of torch_mask_tokens; labels are -100 off the mask.
 * tokens: int32, all spans back to back (no special tokens); offsets: int64 [n_spans + 1]; is_subword: uint8 [vocab]
 * (1 for "##" pieces); outputs int32 [n_spans, max_seq_length].  Randomness is a counter-based hash of
 * (seed, span_index_base + span, purpose, position) - reproducible, and identical in the oracle.
 * ------------------------------------------------------------------------------------------ */
int cocodr_mlm_collate(const int32_t* tokens, const long long* offsets, int n_spans, const uint8_t* is_subword, int vocab,
                       int max_seq_length, int cls_id, int sep_id, int pad_id, int mask_id, double mlm_probability,
                       unsigned long long seed, long long span_index_base, int32_t* input_ids, int32_t* labels,
                       int32_t* attention_mask, cocodr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py roofline): bracket every launch of one kernel class with HIP events
 * on the launch stream.  kind: 0 = off, 1 = GEMM launches, 2 = attention, 3 = score_topk GEMM.
 * ------------------------------------------------------------------------------------------ */
int cocodr_prof_begin(int kind);
/* paused != 0: launches are not bracketed until resumed (the events cost ~1.5 us of stream time each, so a bench
 * samples some of its timed steps instead of all of them); totals keep accumulating */
int cocodr_prof_pause(int paused);
/* synchronises, returns launches / summed ms / summed algorithmic FLOPs, and disables profiling */
int cocodr_prof_end(int* launches, double* total_ms, double* total_flops);
/* what one begin / end event pair adds to the duration of the kernel it brackets, in microseconds: the pair around an empty
 * one-workgroup kernel (median of 33) minus that kernel's own ~1 us; a bench subtracts it per bracketed launch so that its
 * roofline prices kernel time (what rocprofv3 --kernel-trace reports), not event-to-event time */
int cocodr_prof_event_overhead_us(cocodr_stream_t stream, double* overhead_us);

/* Hardware probes used by the GPU tests to pin the MFMA / transposed-LDS-read lane layouts the
 * kernels rely on (out: fp32 / int32 device buffers, see tests/test_gpu_probe.py). */
int cocodr_probe_mfma32(const uint16_t* a /*[32,16]*/, const uint16_t* b /*[32,16]*/, float* out /*[32,32]*/,
                        cocodr_stream_t stream);
int cocodr_probe_tr16(const uint16_t* tile /*[16,64]*/, uint16_t* out /*[64 lanes][4]*/, cocodr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* COCODR_H_ */
