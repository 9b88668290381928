/* CPU twins of the C ABI ("CPU twins *_ref with identical signatures on host pointers").
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): plain C, one thread, straightforward loops - the checker and the
 * `cpu_baseline` of a kernel, never part of the product.  Every function takes exactly the arguments of the entry point of
 * include/cocodr.h it twins (host pointers instead of device pointers, the stream argument ignored) and computes in fp32 with
 * fp64 accumulation, rounding to bf16 (nearest even) only where the device function stores bf16.  What each one restates:
 *
 *   cocodr_gemm_ref              hf nn.Linear forward / backward with the fused epilogues (modeling_bert.py:282-293, 325-351)
 *   cocodr_ln_fwd_ref            hf nn.LayerNorm (modeling_bert.py:288-292), eps inside the square root
 *   cocodr_attn_fwd_ref          hf eager_attention_forward (modeling_bert.py:111-203), key-padding mask, head_dim 64
 *   cocodr_attn_bwd_ref          the autograd backward of the same function: dQ | dK | dV from dctx (+ the query-bias partial sums)
 *   cocodr_ln_bwd_ref            the autograd backward of nn.LayerNorm (+ dgamma, dbeta, column sums of dy)
 *   cocodr_embed_ln_fwd_ref      hf BertEmbeddings.forward (modeling_bert.py:68-108): LN(word[ids] + pos[0..L-1] + type[0])
 *   cocodr_embed_ln_bwd_ref      its backward: scatter-add into the word table, position / type / LayerNorm gradients
 *   cocodr_simce_fwd_bwd_ref     COCO/modeling.py:244-248 compute_contrastive_loss, :172-177 co_target, .mean() at :229
 *   cocodr_triplet_nll_fwd_bwd_ref  ANCE/model/models.py:97-106, 260-261
 *   cocodr_score_topk_ref        faiss.IndexFlatIP(dim).search (evaluate/evaluation/evaluate_beir.py:220-224): exact inner
 *                                products, (score descending, position ascending)
 *   cocodr_topk_merge_ref        the per-shard list merge that replaces ANCE/utils/util.py:117-155 + one search over the
 *                                concatenated shards
 *
 * Pinned by tests/test_ref_twins_cpu.py against the numpy oracle (itself pinned to the reference's golden vectors). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/cocodr.h"

static float bf2f(uint16_t v) {
  uint32_t u = (uint32_t)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint16_t f2bf(float f) { /* round to nearest even; NaN stays NaN */
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static double gelu_erf(double x) { return 0.5 * x * (1.0 + erf(x * 0.7071067811865476)); }
static double gelu_erf_grad(double x) {
  return 0.5 * (1.0 + erf(x * 0.7071067811865476)) + x * 0.3989422804014327 * exp(-0.5 * x * x);
}
/* the dropout mask word of include/cocodr.h "Dropout" */
static uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
static int drop_keep(const cocodr_dropout_mask* d, uint64_t flat) {
  const uint32_t w = lowbias32((uint32_t)(flat >> 1) ^ d->k0) ^ d->k1;
  const uint32_t h = (flat & 1) ? (w >> 16) : (w & 0xffffu);
  return h >= d->threshold;
}

int cocodr_gemm_ref(const cocodr_gemm_args* a, cocodr_stream_t stream) {
  (void)stream;
  if (!a || !a->A || !a->B || !a->C || a->M <= 0 || a->N <= 0 || a->K <= 0) return COCODR_ERR_INVALID;
  const int batch = a->batch > 0 ? a->batch : 1;
  for (int z = 0; z < batch; ++z) {
    const uint16_t* A = a->A + (size_t)z * a->strideA;
    const uint16_t* B = a->B + (size_t)z * a->strideB;
    const float* bias = a->bias ? a->bias + (size_t)z * a->strideBias : NULL;
    const uint16_t* R = a->R ? a->R + (size_t)z * a->strideR : NULL;
    for (int m = 0; m < a->M; ++m)
      for (int n = 0; n < a->N; ++n) {
        double acc = 0.0;
        for (int k = 0; k < a->K; ++k) {
          const float x = bf2f(a->trans_a ? A[(size_t)k * a->lda + m] : A[(size_t)m * a->lda + k]);
          const float w = bf2f(a->trans_b ? B[(size_t)k * a->ldb + n] : B[(size_t)n * a->ldb + k]);
          acc += (double)x * (double)w;
        }
        double v = acc + (bias ? (double)bias[n] : 0.0);
        const size_t o = (size_t)z * a->strideC + (size_t)m * a->ldc + n;
        if (a->epi == COCODR_EPI_GELU) {
          if (a->C2) a->C2[o] = f2bf((float)gelu_erf_grad(v));
          v = gelu_erf(v);
        } else if (a->epi == COCODR_EPI_ADD) {
          if (a->drop.threshold) v = drop_keep(&a->drop, (uint64_t)m * a->N + n) ? v * (double)a->drop.scale : 0.0;
          v += (double)bf2f(R[(size_t)m * a->ldr + n]);
        } else if (a->epi == COCODR_EPI_DGELU) {
          v *= (double)bf2f(R[(size_t)m * a->ldr + n]);
        }
        if (a->out_f32) ((float*)a->C)[o] = (float)v;
        else ((uint16_t*)a->C)[o] = f2bf((float)v);
      }
    if (a->colsum && batch == 1) /* column sums of the epilogue result as stored (bf16 outputs are summed after rounding) */
      for (int n = 0; n < a->N; ++n) {
        double s = 0.0;
        for (int m = 0; m < a->M; ++m)
          s += a->out_f32 ? (double)((float*)a->C)[(size_t)m * a->ldc + n] : (double)bf2f(((uint16_t*)a->C)[(size_t)m * a->ldc + n]);
        a->colsum[n] = (float)s;
      }
  }
  return COCODR_OK;
}

int cocodr_ln_fwd_ref(const uint16_t* y, const float* gamma, const float* beta, uint16_t* out, float* mean, float* rstd,
                      float* cls_out, int cls_stride, int M, int H, float eps, cocodr_stream_t stream) {
  (void)stream;
  if (!y || !gamma || !beta || !out || M <= 0 || H <= 0) return COCODR_ERR_INVALID;
  for (int m = 0; m < M; ++m) {
    double mu = 0.0, var = 0.0;
    for (int h = 0; h < H; ++h) mu += (double)bf2f(y[(size_t)m * H + h]);
    mu /= H;
    for (int h = 0; h < H; ++h) {
      const double d = (double)bf2f(y[(size_t)m * H + h]) - mu;
      var += d * d;
    }
    const double rs = 1.0 / sqrt(var / H + (double)eps);
    if (mean) mean[m] = (float)mu;
    if (rstd) rstd[m] = (float)rs;
    for (int h = 0; h < H; ++h) {
      const double v = ((double)bf2f(y[(size_t)m * H + h]) - mu) * rs * (double)gamma[h] + (double)beta[h];
      out[(size_t)m * H + h] = f2bf((float)v);
      if (cls_out && cls_stride > 0 && m % cls_stride == 0) cls_out[(size_t)(m / cls_stride) * H + h] = (float)v;
    }
  }
  return COCODR_OK;
}

int cocodr_attn_fwd_ref(const uint16_t* qkv, const int32_t* mask, uint16_t* ctx, float* lse, int B, int L, int heads,
                        cocodr_stream_t stream) {
  (void)stream;
  if (!qkv || !mask || !ctx || B <= 0 || L <= 0 || heads <= 0) return COCODR_ERR_INVALID;
  const int H = heads * 64, ld = 3 * H;
  double* s = (double*)malloc((size_t)L * sizeof(double));
  if (!s) return COCODR_ERR_INVALID;
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < heads; ++h)
      for (int i = 0; i < L; ++i) {
        const uint16_t* q = qkv + (size_t)(b * L + i) * ld + h * 64;
        double mx = -INFINITY;
        int any = 0;
        for (int j = 0; j < L; ++j) any |= mask[b * L + j] != 0;
        for (int j = 0; j < L; ++j) {
          const uint16_t* kk = qkv + (size_t)(b * L + j) * ld + H + h * 64;
          double d = 0.0;
          for (int e = 0; e < 64; ++e) d += (double)bf2f(q[e]) * (double)bf2f(kk[e]);
          /* additive key-padding mask: finfo.min in the reference = the key drops out whenever some key is unmasked; a row whose
           * keys are ALL masked keeps a softmax over its raw scores */
          s[j] = (mask[b * L + j] != 0 || !any) ? d * 0.125 : -INFINITY;
          if (s[j] > mx) mx = s[j];
        }
        double sum = 0.0;
        for (int j = 0; j < L; ++j) {
          s[j] = exp(s[j] - mx);
          sum += s[j];
        }
        if (lse) lse[((size_t)b * heads + h) * L + i] = (float)(mx + log(sum));
        for (int e = 0; e < 64; ++e) {
          double o = 0.0;
          for (int j = 0; j < L; ++j) o += s[j] * (double)bf2f(qkv[(size_t)(b * L + j) * ld + 2 * H + h * 64 + e]);
          ctx[(size_t)(b * L + i) * H + h * 64 + e] = f2bf((float)(o / sum));
        }
      }
  free(s);
  return COCODR_OK;
}

/* backward of the attention above: P = softmax(S), dV = P^T dO, dP = dO V^T, dS = P o (dP - rowsum(dP o P)), dQ = dS K / 8,
 * dK = dS^T Q / 8.  The device function recomputes P from the saved log-sum-exp; here it is recomputed from scratch (lse is not
 * read).  qk_bias_partial [4 B, 2 H]: rows 4 b .. 4 b + 3 together hold the column sums of dQ | dK over sequence b - this twin
 * writes the whole sum to row 4 b and zeros to the other three, and exact zeros for the dK half (the rows of dS sum to zero). */
int cocodr_attn_bwd_ref(const uint16_t* qkv, const int32_t* mask, const uint16_t* ctx, const uint16_t* dctx, const float* lse,
                        uint16_t* dqkv, float* qk_bias_partial, int B, int L, int heads, cocodr_stream_t stream) {
  (void)stream; (void)ctx; (void)lse;
  if (!qkv || !mask || !dctx || !dqkv || B <= 0 || L <= 0 || heads <= 0) return COCODR_ERR_INVALID;
  const int H = heads * 64, ld = 3 * H;
  double* P = (double*)malloc((size_t)L * L * sizeof(double));
  double* dS = (double*)malloc((size_t)L * L * sizeof(double));
  if (!P || !dS) { free(P); free(dS); return COCODR_ERR_INVALID; }
  if (qk_bias_partial) memset(qk_bias_partial, 0, (size_t)4 * B * 2 * H * sizeof(float));
  for (int b = 0; b < B; ++b) {
    int any = 0;
    for (int j = 0; j < L; ++j) any |= mask[b * L + j] != 0;
    for (int h = 0; h < heads; ++h) {
      const uint16_t* Q = qkv + (size_t)b * L * ld + h * 64;
      const uint16_t* K = Q + H;
      const uint16_t* V = Q + 2 * H;
      const uint16_t* dO = dctx + (size_t)b * L * H + h * 64;
      for (int i = 0; i < L; ++i) {
        double mx = -INFINITY, sum = 0.0;
        for (int j = 0; j < L; ++j) {
          double d = 0.0;
          for (int e = 0; e < 64; ++e) d += (double)bf2f(Q[(size_t)i * ld + e]) * (double)bf2f(K[(size_t)j * ld + e]);
          P[(size_t)i * L + j] = (mask[b * L + j] != 0 || !any) ? d * 0.125 : -INFINITY;
          if (P[(size_t)i * L + j] > mx) mx = P[(size_t)i * L + j];
        }
        for (int j = 0; j < L; ++j) { P[(size_t)i * L + j] = exp(P[(size_t)i * L + j] - mx); sum += P[(size_t)i * L + j]; }
        double dot = 0.0;
        for (int j = 0; j < L; ++j) {
          P[(size_t)i * L + j] /= sum;
          double dp = 0.0;
          for (int e = 0; e < 64; ++e) dp += (double)bf2f(dO[(size_t)i * H + e]) * (double)bf2f(V[(size_t)j * ld + e]);
          dS[(size_t)i * L + j] = dp;
          dot += dp * P[(size_t)i * L + j];
        }
        for (int j = 0; j < L; ++j) dS[(size_t)i * L + j] = P[(size_t)i * L + j] * (dS[(size_t)i * L + j] - dot);
      }
      for (int i = 0; i < L; ++i)
        for (int e = 0; e < 64; ++e) {
          double dq = 0.0, dk = 0.0, dv = 0.0;
          for (int j = 0; j < L; ++j) {
            dq += dS[(size_t)i * L + j] * (double)bf2f(K[(size_t)j * ld + e]);
            dk += dS[(size_t)j * L + i] * (double)bf2f(Q[(size_t)j * ld + e]);
            dv += P[(size_t)j * L + i] * (double)bf2f(dO[(size_t)j * H + e]);
          }
          uint16_t* o = dqkv + (size_t)(b * L + i) * ld + h * 64 + e;
          o[0] = f2bf((float)(dq * 0.125));
          o[H] = f2bf((float)(dk * 0.125));
          o[2 * H] = f2bf((float)dv);
          if (qk_bias_partial) qk_bias_partial[(size_t)(4 * b) * 2 * H + h * 64 + e] += (float)(dq * 0.125);
        }
    }
  }
  free(P);
  free(dS);
  return COCODR_OK;
}

/* backward of out = (y - mean) rstd gamma + beta w.r.t. y, gamma, beta (mean / rstd as the forward saved them):
 * xhat = (y - mean) rstd, g = dout gamma, dy = rstd (g - mean_h(g) - xhat mean_h(g xhat)); dy_colsum = column sums of dy
 * (before its rounding to bf16) - the bias gradient of the Linear in front of the LayerNorm.  partial is device scratch: ignored. */
int cocodr_ln_bwd_ref(const uint16_t* dout, const uint16_t* y, const float* gamma, const float* mean, const float* rstd, uint16_t* dy,
                      float* dgamma, float* dbeta, float* dy_colsum, float* partial, int M, int H, cocodr_stream_t stream) {
  (void)stream; (void)partial;
  if (!dout || !y || !gamma || !mean || !rstd || !dy || M <= 0 || H <= 0) return COCODR_ERR_INVALID;
  double* dg = (double*)calloc((size_t)3 * H, sizeof(double));
  if (!dg) return COCODR_ERR_INVALID;
  double *db = dg + H, *cs = dg + 2 * H;
  for (int m = 0; m < M; ++m) {
    const double mu = mean[m], rs = rstd[m];
    double s1 = 0.0, s2 = 0.0;
    for (int h = 0; h < H; ++h) {
      const double xh = ((double)bf2f(y[(size_t)m * H + h]) - mu) * rs, d = (double)bf2f(dout[(size_t)m * H + h]);
      const double g = d * (double)gamma[h];
      s1 += g;
      s2 += g * xh;
      dg[h] += d * xh;
      db[h] += d;
    }
    s1 /= H;
    s2 /= H;
    for (int h = 0; h < H; ++h) {
      const double xh = ((double)bf2f(y[(size_t)m * H + h]) - mu) * rs;
      const double g = (double)bf2f(dout[(size_t)m * H + h]) * (double)gamma[h];
      const double v = rs * (g - s1 - xh * s2);
      dy[(size_t)m * H + h] = f2bf((float)v);
      cs[h] += v;   /* the device sums its fp32 values, not the rounded ones it stores */
    }
  }
  for (int h = 0; h < H; ++h) {
    if (dgamma) dgamma[h] = (float)dg[h];
    if (dbeta) dbeta[h] = (float)db[h];
    if (dy_colsum) dy_colsum[h] = (float)cs[h];
  }
  free(dg);
  return COCODR_OK;
}

/* hf BertEmbeddings.forward: the reference never passes token_type_ids / position_ids (COCO/data.py:140,
 * ANCE/model/models.py:226-227): segment row 0, positions 0 .. L - 1.  ids outside [0, vocab) are an error. */
int cocodr_embed_ln_fwd_ref(const int32_t* ids, const float* word, const float* pos, const float* type0, const float* gamma,
                            const float* beta, uint16_t* out, float* mean, float* rstd, int B, int L, int H, int vocab, float eps,
                            cocodr_stream_t stream) {
  (void)stream;
  if (!ids || !word || !pos || !type0 || !gamma || !beta || !out || B <= 0 || L <= 0 || H <= 0) return COCODR_ERR_INVALID;
  double* x = (double*)malloc((size_t)H * sizeof(double));
  if (!x) return COCODR_ERR_INVALID;
  for (int m = 0; m < B * L; ++m) {
    const int id = ids[m], l = m % L;
    if (id < 0 || id >= vocab) { free(x); return COCODR_ERR_INVALID; }
    double mu = 0.0, var = 0.0;
    for (int h = 0; h < H; ++h) {
      x[h] = (double)word[(size_t)id * H + h] + (double)pos[(size_t)l * H + h] + (double)type0[h];
      mu += x[h];
    }
    mu /= H;
    for (int h = 0; h < H; ++h) var += (x[h] - mu) * (x[h] - mu);
    const double rs = 1.0 / sqrt(var / H + (double)eps);
    if (mean) mean[m] = (float)mu;
    if (rstd) rstd[m] = (float)rs;
    for (int h = 0; h < H; ++h) out[(size_t)m * H + h] = f2bf((float)((x[h] - mu) * rs * (double)gamma[h] + (double)beta[h]));
  }
  free(x);
  return COCODR_OK;
}

size_t cocodr_embed_bwd_partial_floats_ref(int L, int H) { (void)L; (void)H; return 0; }

/* backward of the above: dx = LayerNorm backward of dout at x = word[id] + pos[l] + type0 (recomputed), then dword[id] += dx
 * (ACCUMULATED: the caller zeroes dword), dpos[l] = sum over the batch (overwritten, rows [0, L)), dtype0 = sum over all
 * tokens, dgamma / dbeta as in cocodr_ln_bwd_ref.  partial is device scratch: ignored. */
int cocodr_embed_ln_bwd_ref(const uint16_t* dout, const int32_t* ids, const float* word, const float* pos, const float* type0,
                            const float* gamma, const float* mean, const float* rstd, float* dword, float* dpos, float* dtype0,
                            float* dgamma, float* dbeta, float* partial, int B, int L, int H, int vocab, cocodr_stream_t stream) {
  (void)stream; (void)partial;
  if (!dout || !ids || !word || !pos || !type0 || !gamma || !mean || !rstd || !dword || !dpos || !dtype0 || B <= 0 || L <= 0 || H <= 0)
    return COCODR_ERR_INVALID;
  double* acc = (double*)calloc((size_t)(L + 3) * H, sizeof(double));   /* dpos [L][H], dtype0, dgamma, dbeta */
  if (!acc) return COCODR_ERR_INVALID;
  double *dp = acc, *dt = acc + (size_t)L * H, *dg = dt + H, *db = dg + H;
  for (int m = 0; m < B * L; ++m) {
    const int id = ids[m], l = m % L;
    if (id < 0 || id >= vocab) { free(acc); return COCODR_ERR_INVALID; }
    const double mu = mean[m], rs = rstd[m];
    double s1 = 0.0, s2 = 0.0;
    for (int h = 0; h < H; ++h) {
      const double x = (double)word[(size_t)id * H + h] + (double)pos[(size_t)l * H + h] + (double)type0[h];
      const double xh = (x - mu) * rs, d = (double)bf2f(dout[(size_t)m * H + h]), g = d * (double)gamma[h];
      s1 += g;
      s2 += g * xh;
      dg[h] += d * xh;
      db[h] += d;
    }
    s1 /= H;
    s2 /= H;
    for (int h = 0; h < H; ++h) {
      const double x = (double)word[(size_t)id * H + h] + (double)pos[(size_t)l * H + h] + (double)type0[h];
      const double xh = (x - mu) * rs, g = (double)bf2f(dout[(size_t)m * H + h]) * (double)gamma[h];
      const double dx = rs * (g - s1 - xh * s2);
      dword[(size_t)id * H + h] += (float)dx;
      dp[(size_t)l * H + h] += dx;
      dt[h] += dx;
    }
  }
  for (size_t i = 0; i < (size_t)L * H; ++i) dpos[i] = (float)dp[i];
  for (int h = 0; h < H; ++h) {
    dtype0[h] = (float)dt[h];
    if (dgamma) dgamma[h] = (float)dg[h];
    if (dbeta) dbeta[h] = (float)db[h];
  }
  free(acc);
  return COCODR_OK;
}

size_t cocodr_simce_workspace_floats_ref(int M) { return (size_t)(M > 0 ? M : 0) * (size_t)(M > 0 ? M : 0); }

int cocodr_simce_fwd_bwd_ref(const float* E, int M, int H, int world, int row0, int m_local, float* loss_rows, float* loss,
                             float* dE_local, float* workspace, cocodr_stream_t stream) {
  (void)stream;
  if (!E || !loss_rows || !loss || !dE_local || !workspace || M <= 0 || (M & 1) || H <= 0 || world <= 0) return COCODR_ERR_INVALID;
  double* S = (double*)malloc((size_t)M * M * sizeof(double));
  double* lsev = (double*)malloc((size_t)M * sizeof(double));
  if (!S || !lsev) { free(S); free(lsev); return COCODR_ERR_INVALID; }
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < M; ++j) {
      double d = 0.0;
      for (int h = 0; h < H; ++h) d += (double)E[(size_t)i * H + h] * (double)E[(size_t)j * H + h];
      S[(size_t)i * M + j] = i == j ? -INFINITY : d;  /* fill_diagonal_(-inf), COCO/modeling.py:246 */
    }
  double total = 0.0;
  for (int i = 0; i < M; ++i) {
    double mx = -INFINITY, sum = 0.0;
    for (int j = 0; j < M; ++j) if (S[(size_t)i * M + j] > mx) mx = S[(size_t)i * M + j];
    for (int j = 0; j < M; ++j) sum += exp(S[(size_t)i * M + j] - mx);
    lsev[i] = mx + log(sum);
    const double li = (lsev[i] - S[(size_t)i * M + (i ^ 1)]) * world;  /* co_target: [1,0,3,2,...]; x world_size (:247) */
    loss_rows[i] = (float)li;
    total += li;
  }
  loss[0] = (float)(total / M);
  /* d loss / d E_r for the local rows r: (world / M) * sum_j Gs[r][j] E_j with Gs = G + G^T, G = softmax(S) - onehot(target) */
  const double scale = (double)world / M;
  for (int r = 0; r < m_local; ++r) {
    const int i = row0 + r;
    for (int h = 0; h < H; ++h) dE_local[(size_t)r * H + h] = 0.f;
    for (int j = 0; j < M; ++j) {
      if (j == i) continue;
      const double g = exp(S[(size_t)i * M + j] - lsev[i]) + exp(S[(size_t)j * M + i] - lsev[j]) - ((j == (i ^ 1)) ? 2.0 : 0.0);
      for (int h = 0; h < H; ++h) dE_local[(size_t)r * H + h] += (float)(scale * g * (double)E[(size_t)j * H + h]);
    }
  }
  free(S);
  free(lsev);
  return COCODR_OK;
}

int cocodr_triplet_nll_fwd_bwd_ref(const float* q, const float* a, const float* b, const float* weights, int B, int H,
                                   float* loss_rows, float* logits, float* loss, float* dq, float* da, float* db,
                                   cocodr_stream_t stream) {
  (void)stream;
  if (!q || !a || !b || !loss_rows || !logits || !loss || !dq || !da || !db || B <= 0 || H <= 0) return COCODR_ERR_INVALID;
  double total = 0.0;
  for (int i = 0; i < B; ++i) {
    double la = 0.0, lb = 0.0;
    for (int h = 0; h < H; ++h) {
      la += (double)q[(size_t)i * H + h] * (double)a[(size_t)i * H + h];
      lb += (double)q[(size_t)i * H + h] * (double)b[(size_t)i * H + h];
    }
    logits[2 * i] = (float)la;
    logits[2 * i + 1] = (float)lb;
    const double mx = la > lb ? la : lb;
    const double lsev = mx + log(exp(la - mx) + exp(lb - mx));
    const double li = lsev - la;  /* -log_softmax(logits)[:, 0] */
    const double w = weights ? (double)weights[i] : 1.0;
    loss_rows[i] = (float)li;
    total += li * w;
    const double pb = exp(lb - lsev), c = w / B;  /* d(mean(loss * w)) / d la = -(1 - pa) * w / B = -pb w / B */
    for (int h = 0; h < H; ++h) {
      const double qa = a[(size_t)i * H + h], qb = b[(size_t)i * H + h], qq = q[(size_t)i * H + h];
      dq[(size_t)i * H + h] = (float)(c * pb * (qb - qa));
      da[(size_t)i * H + h] = (float)(-c * pb * qq);
      db[(size_t)i * H + h] = (float)(c * pb * qq);
    }
  }
  loss[0] = (float)(total / B);
  return COCODR_OK;
}

typedef struct { float d; long long i; } cand_t;
static int cand_cmp(const void* x, const void* y) {
  const cand_t *a = (const cand_t*)x, *b = (const cand_t*)y;
  if (a->d > b->d) return -1;
  if (a->d < b->d) return 1;
  return a->i < b->i ? -1 : (a->i > b->i ? 1 : 0);
}

size_t cocodr_score_topk_workspace_bytes_ref(int Nq, int Np, int k) { (void)Nq; (void)k; return (size_t)(Np > 0 ? Np : 0) * sizeof(cand_t); }

int cocodr_score_topk_ref(const float* Q, const float* P, int Nq, int Np, int H, int k, long long id_offset, float* D,
                          long long* I, void* workspace, size_t workspace_bytes, cocodr_stream_t stream) {
  (void)stream;
  if (!Q || !P || !D || !I || !workspace || Nq <= 0 || Np <= 0 || H <= 0 || k <= 0) return COCODR_ERR_INVALID;
  if (workspace_bytes < (size_t)Np * sizeof(cand_t)) return COCODR_ERR_WORKSPACE;
  cand_t* c = (cand_t*)workspace;
  for (int qi = 0; qi < Nq; ++qi) {
    for (int p = 0; p < Np; ++p) {
      float acc = 0.f;  /* the fp32 fma chain the exact-fp32 device pipeline is bit-identical to */
      for (int h = 0; h < H; ++h) acc = fmaf(Q[(size_t)qi * H + h], P[(size_t)p * H + h], acc);
      c[p].d = acc;
      c[p].i = p;
    }
    qsort(c, (size_t)Np, sizeof(cand_t), cand_cmp);
    for (int j = 0; j < k; ++j) {
      D[(size_t)qi * k + j] = j < Np ? c[j].d : -INFINITY;
      I[(size_t)qi * k + j] = j < Np ? c[j].i + id_offset : -1;
    }
  }
  return COCODR_OK;
}

int cocodr_topk_merge_ref(const float* D, const int32_t* I, const long long* shard_offset, int W, int Nq, int k, long long stride_w,
                          float* outD, long long* outI, int k_out, cocodr_stream_t stream) {
  (void)stream;
  if (!D || !I || !shard_offset || !outD || !outI || W < 1 || Nq < 0 || k < 1 || k_out < 1 || k_out > W * k) return COCODR_ERR_INVALID;
  cand_t* c = (cand_t*)malloc((size_t)W * k * sizeof(cand_t));
  if (!c) return COCODR_ERR_INVALID;
  for (int q = 0; q < Nq; ++q) {
    int n = 0;
    for (int w = 0; w < W; ++w)
      for (int j = 0; j < k; ++j) {
        const size_t g = (size_t)w * stride_w + (size_t)q * k + j;
        if (I[g] >= 0) { c[n].d = D[g]; c[n].i = shard_offset[w] + I[g]; ++n; }
      }
    qsort(c, (size_t)n, sizeof(cand_t), cand_cmp);
    for (int j = 0; j < k_out; ++j) {
      outD[(size_t)q * k_out + j] = j < n ? c[j].d : -INFINITY;
      outI[(size_t)q * k_out + j] = j < n ? c[j].i : -1;
    }
  }
  free(c);
  return COCODR_OK;
}
