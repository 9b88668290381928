// Whole-encoder orchestration: the BertModel layer loop (hf: BertEncoder / BertLayer,
// modeling_bert.py) and its backward, enqueued from native code on the caller's stream so one host
// call covers ~85 (forward) / ~100 (backward) kernel launches with no Python in between.
//
// Memory plan (sized for 288 GB HBM, nothing is recomputed and nothing is freed mid-step): every
// per-layer activation the backward needs is kept in one caller-provided arena as [layers][M, *]
// arrays with a uniform layer stride.  That uniform stride is what lets the weight gradients of ALL
// layers be computed after the dgrad chain by ONE batched launch per weight matrix
// (batch index = layer): thousands of 128x128 tiles per launch instead of 36..144, no split-K, no
// atomics, deterministic, fp32 results written straight into the caller's gradient buffers.
#include <algorithm>

#include <string.h>

#include "common.h"

namespace {

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct Carver {
  size_t off = 0;
  size_t take(size_t bytes) {
    const size_t o = off;
    off = align_up(off + bytes);
    return o;
  }
};

struct BwdLayout {
  size_t dy2, du, dy1, dqkv;  // bf16 [layers][M,*]
  size_t dxa, dxb, dctx;      // bf16 [M,H]
  size_t dres;                // bf16 [M,H]: with dropout, the un-masked LayerNorm-input gradient (the residual branch's)
  size_t ln_partial, colsum_partial, emb_partial;  // fp32
  size_t ln2_slots, ln1_slots, b1_slots, bv_slots, bqk_slots;  // fp32 per-layer partial rows of the deferred reductions
  size_t multi_ws;  // fp32 partial tiles of the merged weight-gradient launch's cut last round (cocodr_gemm_multi)
  size_t multi_ws_floats;
  size_t total;
};

// shapes (no pointers) of the four grouped weight-gradient problems of a backward range of ng layers over M token rows:
// dWqkv [3H,H], dWo [H,H], dW1 [I,H], dW2 [H,I] = dY^T X, contraction over the tokens, batch = layer
void weight_grad_shapes(cocodr_gemm_args (&wg)[4], int M, int H, int I, int ng) {
  const int rows[4] = {3 * H, H, I, H}, cols[4] = {H, H, H, I};
  for (int q = 0; q < 4; ++q) {
    memset(&wg[q], 0, sizeof(wg[q]));
    wg[q].M = rows[q]; wg[q].N = cols[q]; wg[q].K = M;
    wg[q].lda = rows[q]; wg[q].ldb = cols[q]; wg[q].ldc = cols[q];
    wg[q].trans_a = wg[q].trans_b = 1; wg[q].out_f32 = 1; wg[q].batch = ng; wg[q].epi = COCODR_EPI_NONE;
  }
}

BwdLayout bwd_layout_m(const cocodr_config* c, size_t M, int B, int L) {
  const size_t H = c->hidden, I = c->inter, N = c->layers;
  Carver cv;
  BwdLayout b;
  b.dy2 = cv.take(N * M * H * 2);
  b.du = cv.take(N * M * I * 2);
  b.dy1 = cv.take(N * M * H * 2);
  b.dqkv = cv.take(N * M * 3 * H * 2);
  b.dxa = cv.take(M * H * 2);
  b.dxb = cv.take(M * H * 2);
  b.dctx = cv.take(M * H * 2);
  b.dres = cv.take(M * H * 2);
  b.ln_partial = cv.take(cocodr_ln_bwd_partial_floats((int)M, (int)H) * 4);
  b.colsum_partial = cv.take(std::max(cocodr_colsum_partial_floats((int)M, (int)std::max(I, 3 * H), (int)N),
                                      cocodr_gemm_colsum_partial_floats((int)M, (int)std::max(I, 3 * H))) * 4);
  b.emb_partial = cv.take(std::max(cocodr_embed_bwd_partial_floats(L, (int)H), cocodr_embed_bwd_packed_partial_floats((int)M, (int)H)) * 4);
  b.ln2_slots = cv.take(N * cocodr_ln_bwd_partial_floats((int)M, (int)H) * 4);
  b.ln1_slots = cv.take(N * cocodr_ln_bwd_partial_floats((int)M, (int)H) * 4);
  b.b1_slots = cv.take(N * cocodr_gemm_colsum_partial_floats((int)M, (int)I) * 4);
  b.bv_slots = cv.take(N * cocodr_gemm_colsum_partial_floats((int)M, (int)H) * 4);
  b.bqk_slots = cv.take(N * (size_t)4 * B * 2 * H * 4);  // attention backward: 4 B partial rows of dQ | dK column sums
  // the merged weight-gradient launch of a backward range [lo, hi) cuts its last partial round into contraction slices: the
  // partial tiles of the largest cut over all range lengths (0 - not 64 MiB - where no range is merged or none leaves a tail)
  size_t ws = 0;
  for (size_t ng = 1; ng <= N; ++ng) {
    cocodr_gemm_args wg[4];
    weight_grad_shapes(wg, (int)M, (int)H, (int)I, (int)ng);
    ws = std::max(ws, cocodr_gemm_multi_workspace_floats_for(wg, 4));
    if (ng > 1) {  // a top range behind a [CLS] tail: the top layer's dWo, dW1, dW2 are not in the launch
      wg[1].batch = wg[2].batch = wg[3].batch = (int)ng - 1;
      ws = std::max(ws, cocodr_gemm_multi_workspace_floats_for(wg, 4));
    } else {
      ws = std::max(ws, cocodr_gemm_multi_workspace_floats_for(wg, 1));
    }
  }
  b.multi_ws_floats = ws;
  b.multi_ws = cv.take(ws * 4);
  b.total = cv.off;
  return b;
}
BwdLayout bwd_layout(const cocodr_config* c, int B, int L) { return bwd_layout_m(c, (size_t)B * L, B, L); }

int check_cfg(const cocodr_config* c, int B, int L) {
  CK_ARG(c != nullptr, "encoder: null config");
  CK_ARG(c->heads > 0 && c->hidden == c->heads * 64, "encoder: hidden=%d must be heads*64 (heads=%d)", c->hidden, c->heads);
  CK_ARG(c->hidden % 128 == 0 && c->hidden <= 1024, "encoder: hidden=%d must be a multiple of 128, <= 1024", c->hidden);
  CK_ARG(c->inter % 128 == 0 && c->inter > 0, "encoder: intermediate=%d must be a multiple of 128", c->inter);
  CK_ARG(c->layers > 0 && c->vocab > 0, "encoder: bad layers/vocab");
  CK_ARG(B > 0 && L >= 32 && L % 32 == 0 && L <= 512 && L <= c->max_pos, "encoder: L=%d must be a multiple of 32 in [32, min(512,%d)]", L, c->max_pos);
  CK_ARG(c->hidden_dropout >= 0.f && c->hidden_dropout < 1.f && c->attn_dropout >= 0.f && c->attn_dropout < 1.f, "encoder: dropout probabilities must be in [0, 1)");
  return COCODR_OK;
}

// [CLS] tail (cocodr_config.cls_tail): the carve-up of hidden_states[layers]'s slot, which such a forward does not fill
struct TailScratch {
  uint16_t *ctx_c, *xin_c, *out_c;  // bf16 [B,H] each: gathered attention context rows, gathered residual rows, LayerNorm output
  long long* idx;                   // [B] first row of every sequence
};
TailScratch tail_scratch(uint16_t* slot, int B, int H) {  // M >= 32 B rows of H: 3 B H bf16 + B indices fit with room to spare
  TailScratch t;
  t.ctx_c = slot; t.xin_c = slot + (size_t)B * H; t.out_c = slot + (size_t)2 * B * H;
  t.idx = reinterpret_cast<long long*>(slot + (size_t)3 * B * H);
  return t;
}

#define TRY(expr)                \
  do {                           \
    const int rc_ = (expr);      \
    if (rc_ != COCODR_OK) return rc_; \
  } while (0)

// dropout masks of one layer (threshold 0 everywhere when the call runs without dropout)
struct LayerDrop {
  cocodr_dropout_mask probs, attn_out, ffn_out;
};
bool drop_active(const cocodr_config* c) { return c->hidden_dropout > 0.f || c->attn_dropout > 0.f; }
int layer_drop(const cocodr_config* c, bool active, int l, LayerDrop* d) {
  const double ph = active ? c->hidden_dropout : 0.0, pa = active ? c->attn_dropout : 0.0;
  TRY(cocodr_dropout_mask_for(pa, c->drop_seed, c->drop_call, l, COCODR_DROP_ATTN_PROBS, &d->probs));
  TRY(cocodr_dropout_mask_for(ph, c->drop_seed, c->drop_call, l, COCODR_DROP_ATTN_OUT, &d->attn_out));
  TRY(cocodr_dropout_mask_for(ph, c->drop_seed, c->drop_call, l, COCODR_DROP_FFN_OUT, &d->ffn_out));
  return COCODR_OK;
}

// (split workspace of the arena the current call works on: set by encoder_fwd_impl / encoder_bwd_impl before their first GEMM;
//  the library is single-threaded per call chain - one host thread enqueues a pass - and every gemm_base call re-reads it)
thread_local float* t_split_ws = nullptr;
thread_local size_t t_split_ws_floats = 0;
cocodr_gemm_args gemm_base(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int ta, int tb) {
  cocodr_gemm_args g = {};
  g.split_ws = t_split_ws; g.split_ws_floats = t_split_ws_floats;
  g.A = (const uint16_t*)A; g.B = (const uint16_t*)B; g.C = C;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.trans_a = ta; g.trans_b = tb; g.epi = COCODR_EPI_NONE; g.batch = 1;
  return g;
}

}  // namespace

namespace {
int check_packed_batch(const cocodr_config* c, const cocodr_packed_batch* pk) {
  CK_ARG(pk != nullptr && pk->mask && pk->seq_off, "encoder(packed): null batch");
  TRY(check_cfg(c, 1, 32));
  CK_ARG(pk->B > 0 && pk->T >= pk->B && pk->T % 32 == 0, "encoder(packed): T=%d must be a multiple of 32 and >= B (B=%d)", pk->T, pk->B);
  CK_ARG(pk->max_len >= 32 && pk->max_len % 32 == 0 && pk->max_len <= 512 && pk->max_len <= c->max_pos && pk->drop_L >= pk->max_len,
         "encoder(packed): max_len=%d must be a multiple of 32 in [32, min(512,%d)] and <= drop_L=%d", pk->max_len, c->max_pos, pk->drop_L);
  return COCODR_OK;
}
int layout_m(const cocodr_config* c, size_t M, int B, int L, int training, cocodr_encoder_layout_t* out);
}  // namespace

extern "C" int cocodr_encoder_layout(const cocodr_config* c, int B, int L, int training, cocodr_encoder_layout_t* out) {
  TRY(check_cfg(c, B, L));
  CK_ARG(out != nullptr, "encoder_layout: null out");
  return layout_m(c, (size_t)B * L, B, L, training, out);
}
extern "C" int cocodr_encoder_layout_packed(const cocodr_config* c, int T, int B, int training, cocodr_encoder_layout_t* out) {
  TRY(check_cfg(c, 1, 32));
  CK_ARG(out != nullptr && B > 0 && T >= B && T % 32 == 0, "encoder_layout_packed: T=%d must be a multiple of 32 and >= B (B=%d)", T, B);
  return layout_m(c, (size_t)T, B, 32, training, out);
}

namespace {
int layout_m(const cocodr_config* c, size_t M, int B, int L, int training, cocodr_encoder_layout_t* out) {
  const size_t H = c->hidden, I = c->inter;
  const size_t NL = training ? c->layers : 1;
  Carver cv;
  out->hidden = cv.take((size_t)(c->layers + 1) * M * H * 2);
  out->cls_f32 = cv.take((size_t)B * H * 4);
  out->qkv = cv.take(NL * M * 3 * H * 2);
  out->ctx = cv.take(NL * M * H * 2);
  out->y1 = cv.take(NL * M * H * 2);
  out->x1 = cv.take(NL * M * H * 2);
  out->u = cv.take(NL * M * I * 2);
  out->h = cv.take(NL * M * I * 2);
  out->y2 = cv.take(NL * M * H * 2);
  out->lse = cv.take(NL * M * c->heads * 4);
  out->mean1 = cv.take(NL * M * 4);
  out->rstd1 = cv.take(NL * M * 4);
  out->mean2 = cv.take(NL * M * 4);
  out->rstd2 = cv.take(NL * M * 4);
  out->emb_mean = cv.take(M * 4);
  out->emb_rstd = cv.take(M * 4);
  // the widest GEMM of a layer has more 256 x 256 tiles than the chip has compute units: a workspace lets every forward / dgrad
  // launch cut its last partial round into contraction slices (gemm_pp.hip launch_split)
  out->split_ws_floats = ((M + 255) / 256) * (std::max(I, 3 * H) / 256) > 256 ? cocodr_gemm_split_workspace_floats() : 0;
  out->split_ws = cv.take(out->split_ws_floats * 4);
  out->bwd_scratch = cv.off;
  out->bwd_bytes = training ? bwd_layout_m(c, M, B, L).total : 0;
  out->bwd_dx = training ? cv.off + bwd_layout_m(c, M, B, L).dxb : 0;
  out->total_bytes = cv.off + out->bwd_bytes;
  return COCODR_OK;
}
}  // namespace

namespace {
int encoder_fwd_impl(const cocodr_config* c, const cocodr_embed_params* emb, const cocodr_layer_params* lp, const int32_t* ids,
                     const int32_t* mask, int B, int L, int training, void* arena, size_t arena_bytes, bool from_hidden,
                     cocodr_stream_t stream, int layer_lo = 0, int layer_hi = -1, const cocodr_packed_batch* pk = nullptr);
}

extern "C" int cocodr_encoder_bwd_layout(const cocodr_config* c, int B, int L, cocodr_encoder_bwd_layout_t* out) {
  CK_ARG(out != nullptr, "encoder_bwd_layout: null out");
  cocodr_encoder_layout_t lay;
  TRY(cocodr_encoder_layout(c, B, L, 1, &lay));
  const BwdLayout bl = bwd_layout(c, B, L);
  const size_t base = lay.bwd_scratch;
  out->dy2 = base + bl.dy2; out->du = base + bl.du; out->dy1 = base + bl.dy1; out->dqkv = base + bl.dqkv;
  out->ln2_partial = base + bl.ln2_slots; out->ln1_partial = base + bl.ln1_slots;
  const int M = B * L;
  out->ln_blocks = cocodr_ln_bwd_blocks(M);
  out->ln_rows = (M + out->ln_blocks - 1) / out->ln_blocks;
  return COCODR_OK;
}

extern "C" int cocodr_encoder_fwd(const cocodr_config* c, const cocodr_embed_params* emb, const cocodr_layer_params* lp,
                                  const int32_t* ids, const int32_t* mask, int B, int L, int training, void* arena,
                                  size_t arena_bytes, cocodr_stream_t stream) {
  CK_ARG(emb && ids, "encoder_fwd: null pointer");
  return encoder_fwd_impl(c, emb, lp, ids, mask, B, L, training, arena, arena_bytes, false, stream);
}

extern "C" int cocodr_encoder_fwd_range(const cocodr_config* c, const cocodr_embed_params* emb, const cocodr_layer_params* lp,
                                        const int32_t* ids, const int32_t* mask, int B, int L, int training, void* arena,
                                        size_t arena_bytes, int layer_lo, int layer_hi, cocodr_stream_t stream) {
  CK_ARG(layer_lo > 0 || (emb && ids), "encoder_fwd_range: the range that starts at layer 0 needs the embedding inputs");
  return encoder_fwd_impl(c, emb, lp, ids, mask, B, L, training, arena, arena_bytes, false, stream, layer_lo, layer_hi);
}

extern "C" int cocodr_encoder_fwd_packed(const cocodr_config* c, const cocodr_embed_params* emb, const cocodr_layer_params* lp,
                                         const cocodr_packed_batch* batch, int training, void* arena, size_t arena_bytes,
                                         cocodr_stream_t stream) {
  CK_ARG(emb && batch && batch->ids && batch->positions && batch->cls_slot, "encoder_fwd_packed: null pointer");
  return encoder_fwd_impl(c, emb, lp, nullptr, nullptr, 0, 0, training, arena, arena_bytes, false, stream, 0, -1, batch);
}

extern "C" int cocodr_stack_fwd(const cocodr_config* c, const cocodr_layer_params* lp, const int32_t* mask, int B, int L,
                                int training, void* arena, size_t arena_bytes, cocodr_stream_t stream) {
  return encoder_fwd_impl(c, nullptr, lp, nullptr, mask, B, L, training, arena, arena_bytes, true, stream);
}

extern "C" int cocodr_stack_fwd_packed(const cocodr_config* c, const cocodr_layer_params* lp, const cocodr_packed_batch* batch,
                                       int training, void* arena, size_t arena_bytes, cocodr_stream_t stream) {
  CK_ARG(batch && batch->cls_slot, "stack_fwd_packed: null batch");
  return encoder_fwd_impl(c, nullptr, lp, nullptr, nullptr, 0, 0, training, arena, arena_bytes, true, stream, 0, -1, batch);
}

namespace {
int encoder_fwd_impl(const cocodr_config* c, const cocodr_embed_params* emb, const cocodr_layer_params* lp, const int32_t* ids,
                     const int32_t* mask, int B, int L, int training, void* arena, size_t arena_bytes, bool from_hidden,
                     cocodr_stream_t stream, int layer_lo, int layer_hi, const cocodr_packed_batch* pk) {
  cocodr_encoder_layout_t lay;
  if (pk) {  // packed batch: T rows, per-sequence extents (ids / mask / B come from the batch)
    TRY(check_packed_batch(c, pk));
    TRY(cocodr_encoder_layout_packed(c, pk->T, pk->B, training, &lay));
    ids = pk->ids; mask = pk->mask; B = pk->B; L = pk->max_len;
  } else {
    TRY(cocodr_encoder_layout(c, B, L, training, &lay));
  }
  CK_ARG(lp && mask && arena, "encoder_fwd: null pointer");
  if (arena_bytes < lay.total_bytes) {
    cocodr_set_error("encoder_fwd: arena %zu B < required %zu B", arena_bytes, lay.total_bytes);
    return COCODR_ERR_WORKSPACE;
  }
  CK_ARG(((uintptr_t)arena & 255) == 0, "encoder_fwd: arena must be 256-byte aligned");
  char* base = (char*)arena;
  const int M = pk ? pk->T : B * L, H = c->hidden, I = c->inter, NL = c->layers;
  t_split_ws = lay.split_ws_floats ? (float*)(base + lay.split_ws) : nullptr;
  t_split_ws_floats = lay.split_ws_floats;
  const size_t ls = training ? 1 : 0;  // per-layer stride multiplier
  uint16_t* hidden = (uint16_t*)(base + lay.hidden);
  float* cls = (float*)(base + lay.cls_f32);

  if (layer_hi < 0) layer_hi = NL;
  CK_ARG(0 <= layer_lo && layer_lo <= layer_hi && layer_hi <= NL, "encoder_fwd: bad layer range [%d,%d)", layer_lo, layer_hi);
  // dropout (hf nn.Dropout under model.train()): only a training forward drops; the site keys follow (seed, call, layer, kind)
  const bool dropping = training && drop_active(c);
  CK_ARG(!(c->cls_tail && dropping), "encoder_fwd: the [CLS] tail is not available with dropout (masks are indexed by token row)");
  CK_ARG(!(c->cls_tail && training && B % 8 != 0), "encoder_fwd: the [CLS] tail of a training forward needs B %% 8 == 0 (its weight gradients contract over the B rows)");
  if (!from_hidden && layer_lo == 0) {  // a bare layer stack (Condenser head) starts from hidden slot 0, filled by the caller
    cocodr_dropout_mask de;
    TRY(cocodr_dropout_mask_for(dropping ? c->hidden_dropout : 0.0, c->drop_seed, c->drop_call, 0, COCODR_DROP_EMBED, &de));
    if (pk) {
      CK_ARG(pk->ids && pk->positions, "encoder_fwd(packed): the embedding needs ids and positions");
      TRY(cocodr_embed_ln_fwd_packed(pk->ids, pk->positions, emb->word, emb->pos, emb->type0, emb->ln_g, emb->ln_b, hidden,
                                     (float*)(base + lay.emb_mean), (float*)(base + lay.emb_rstd), M, H, c->vocab, c->ln_eps, &de, stream));
    } else {
      TRY(cocodr_embed_ln_fwd_drop(ids, emb->word, emb->pos, emb->type0, emb->ln_g, emb->ln_b, hidden, (float*)(base + lay.emb_mean),
                                   (float*)(base + lay.emb_rstd), B, L, H, c->vocab, c->ln_eps, &de, stream));
    }
  }
  for (int l = layer_lo; l < layer_hi; ++l) {
    const cocodr_layer_params& w = lp[l];
    const size_t lo = ls * l;
    uint16_t* x_in = hidden + (size_t)l * M * H;
    uint16_t* x_out = hidden + (size_t)(l + 1) * M * H;
    uint16_t* qkv = (uint16_t*)(base + lay.qkv) + lo * M * 3 * H;
    uint16_t* ctx = (uint16_t*)(base + lay.ctx) + lo * M * H;
    uint16_t* y1 = (uint16_t*)(base + lay.y1) + lo * M * H;
    uint16_t* x1 = (uint16_t*)(base + lay.x1) + lo * M * H;
    uint16_t* u = (uint16_t*)(base + lay.u) + lo * M * I;
    uint16_t* h = (uint16_t*)(base + lay.h) + lo * M * I;
    uint16_t* y2 = (uint16_t*)(base + lay.y2) + lo * M * H;
    float* lse = (float*)(base + lay.lse) + lo * M * c->heads;
    float* mean1 = (float*)(base + lay.mean1) + lo * M;
    float* rstd1 = (float*)(base + lay.rstd1) + lo * M;
    float* mean2 = (float*)(base + lay.mean2) + lo * M;
    float* rstd2 = (float*)(base + lay.rstd2) + lo * M;

    LayerDrop ld;
    TRY(layer_drop(c, dropping, l, &ld));
    cocodr_gemm_args g = gemm_base(x_in, w.wqkv, qkv, M, 3 * H, H, H, H, 3 * H, 0, 0);
    g.bias = w.bqkv;
    TRY(cocodr_gemm(&g, stream));
    if (pk) TRY(cocodr_attn_fwd_packed(qkv, mask, ctx, lse, pk->seq_off, pk->seq_order, B, M, pk->max_len, c->heads, &ld.probs, pk->drop_L, stream));
    else TRY(cocodr_attn_fwd_drop(qkv, mask, ctx, lse, B, L, c->heads, &ld.probs, stream));
    if (c->cls_tail && l == NL - 1) {
      // [CLS] tail (cocodr_config.cls_tail): everything behind the attention on the B first rows of the sequences only.  The
      // per-layer slots keep their first B rows; hidden_states[NL]'s slot is scratch: the gathered context rows, the gathered
      // residual rows, the bf16 LayerNorm output and the B row indices (the backward reads the first two and the indices)
      const TailScratch ts = tail_scratch(x_out, B, H);
      TRY(cocodr_cls_rows(pk ? pk->seq_off : nullptr, L, B, ts.idx, stream));
      TRY(cocodr_gather_rows(ctx, ts.idx, ts.ctx_c, B, H, stream));
      TRY(cocodr_gather_rows(x_in, ts.idx, ts.xin_c, B, H, stream));
      g = gemm_base(ts.ctx_c, w.wo, y1, B, H, H, H, H, H, 0, 0);
      g.bias = w.bo; g.epi = COCODR_EPI_ADD; g.R = ts.xin_c; g.ldr = H;
      TRY(cocodr_gemm(&g, stream));
      TRY(cocodr_ln_fwd(y1, w.ln1_g, w.ln1_b, x1, mean1, rstd1, nullptr, 0, B, H, c->ln_eps, stream));
      g = gemm_base(x1, w.w1, h, B, I, H, H, H, I, 0, 0);
      g.bias = w.b1; g.epi = COCODR_EPI_GELU; g.C2 = training ? u : nullptr;
      TRY(cocodr_gemm(&g, stream));
      g = gemm_base(h, w.w2, y2, B, H, I, I, I, H, 0, 0);
      g.bias = w.b2; g.epi = COCODR_EPI_ADD; g.R = x1; g.ldr = H;
      TRY(cocodr_gemm(&g, stream));
      TRY(cocodr_ln_fwd(y2, w.ln2_g, w.ln2_b, ts.out_c, mean2, rstd2, cls, 1, B, H, c->ln_eps, stream));  // every row is a [CLS] row
      continue;
    }
    g = gemm_base(ctx, w.wo, y1, M, H, H, H, H, H, 0, 0);
    g.bias = w.bo; g.epi = COCODR_EPI_ADD; g.R = x_in; g.ldr = H; g.drop = ld.attn_out;
    TRY(cocodr_gemm(&g, stream));
    TRY(cocodr_ln_fwd(y1, w.ln1_g, w.ln1_b, x1, mean1, rstd1, nullptr, 0, M, H, c->ln_eps, stream));
    g = gemm_base(x1, w.w1, h, M, I, H, H, H, I, 0, 0);
    g.bias = w.b1; g.epi = COCODR_EPI_GELU; g.C2 = training ? u : nullptr;  // u: GELU'(pre-activation) for the backward
    TRY(cocodr_gemm(&g, stream));
    g = gemm_base(h, w.w2, y2, M, H, I, I, I, H, 0, 0);
    g.bias = w.b2; g.epi = COCODR_EPI_ADD; g.R = x1; g.ldr = H; g.drop = ld.ffn_out;
    TRY(cocodr_gemm(&g, stream));
    TRY(cocodr_ln_fwd_slots(y2, w.ln2_g, w.ln2_b, x_out, mean2, rstd2, (l == NL - 1) ? cls : nullptr, L, pk ? pk->cls_slot : nullptr, M, H,
                            c->ln_eps, stream));
  }
  return COCODR_OK;
}
}  // namespace

namespace {
int encoder_bwd_impl(const cocodr_config* c, const cocodr_embed_params* emb, const cocodr_layer_params* lp,
                     const cocodr_embed_grads* eg, const cocodr_layer_grads* lg, const int32_t* ids, const int32_t* mask,
                     const uint16_t* d_in, int B, int L, void* arena, size_t arena_bytes, int layer_hi, int layer_lo, int do_embed,
                     cocodr_stream_t stream, const cocodr_packed_batch* pk);
}
extern "C" int cocodr_encoder_bwd_range(const cocodr_config* c, const cocodr_embed_params* emb, const cocodr_layer_params* lp,
                                        const cocodr_embed_grads* eg, const cocodr_layer_grads* lg, const int32_t* ids,
                                        const int32_t* mask, const uint16_t* d_in, int B, int L, void* arena,
                                        size_t arena_bytes, int layer_hi, int layer_lo, int do_embed, cocodr_stream_t stream) {
  return encoder_bwd_impl(c, emb, lp, eg, lg, ids, mask, d_in, B, L, arena, arena_bytes, layer_hi, layer_lo, do_embed, stream, nullptr);
}
extern "C" int cocodr_encoder_bwd_packed(const cocodr_config* c, const cocodr_embed_params* emb, const cocodr_layer_params* lp,
                                         const cocodr_embed_grads* eg, const cocodr_layer_grads* lg, const cocodr_packed_batch* batch,
                                         const uint16_t* d_in, void* arena, size_t arena_bytes, int layer_hi, int layer_lo,
                                         int do_embed, cocodr_stream_t stream) {
  CK_ARG(batch != nullptr, "encoder_bwd_packed: null batch");
  return encoder_bwd_impl(c, emb, lp, eg, lg, nullptr, nullptr, d_in, 0, 0, arena, arena_bytes, layer_hi, layer_lo, do_embed, stream, batch);
}
namespace {
int encoder_bwd_impl(const cocodr_config* c, const cocodr_embed_params* emb, const cocodr_layer_params* lp,
                     const cocodr_embed_grads* eg, const cocodr_layer_grads* lg, const int32_t* ids, const int32_t* mask,
                     const uint16_t* d_in, int B, int L, void* arena, size_t arena_bytes, int layer_hi, int layer_lo, int do_embed,
                     cocodr_stream_t stream, const cocodr_packed_batch* pk) {
  cocodr_encoder_layout_t lay;
  if (pk) {
    TRY(check_packed_batch(c, pk));
    TRY(cocodr_encoder_layout_packed(c, pk->T, pk->B, 1, &lay));
    ids = pk->ids; mask = pk->mask; B = pk->B; L = pk->max_len;
  } else {
    TRY(cocodr_encoder_layout(c, B, L, 1, &lay));
  }
  CK_ARG(lp && lg && mask && arena && (!do_embed || (emb && eg && ids)), "encoder_bwd: null pointer");
  CK_ARG(0 <= layer_lo && layer_lo <= layer_hi && layer_hi <= c->layers, "encoder_bwd: bad layer range [%d,%d)", layer_lo, layer_hi);
  CK_ARG(d_in != nullptr || layer_hi < c->layers, "encoder_bwd: the first (top) range needs the upstream gradient d_in");
  if (arena_bytes < lay.total_bytes) {
    cocodr_set_error("encoder_bwd: arena %zu B < required %zu B (was the forward run with training=1?)", arena_bytes, lay.total_bytes);
    return COCODR_ERR_WORKSPACE;
  }
  const int M = pk ? pk->T : B * L, H = c->hidden, I = c->inter, NL = c->layers;
  // uniform layer stride of the gradient blocks (see header)
  long long s_wqkv = 0, s_wo = 0, s_w1 = 0, s_w2 = 0, s_bqkv = 0, s_bo = 0, s_b1 = 0, s_b2 = 0;
  if (NL > 1) {
    s_wqkv = lg[1].wqkv - lg[0].wqkv; s_wo = lg[1].wo - lg[0].wo; s_w1 = lg[1].w1 - lg[0].w1; s_w2 = lg[1].w2 - lg[0].w2;
    s_bqkv = lg[1].bqkv - lg[0].bqkv; s_bo = lg[1].bo - lg[0].bo; s_b1 = lg[1].b1 - lg[0].b1; s_b2 = lg[1].b2 - lg[0].b2;
    for (int l = 1; l < NL; ++l) {
      CK_ARG(lg[l].wqkv - lg[l - 1].wqkv == s_wqkv && lg[l].wo - lg[l - 1].wo == s_wo && lg[l].w1 - lg[l - 1].w1 == s_w1 &&
                 lg[l].w2 - lg[l - 1].w2 == s_w2 && lg[l].bqkv - lg[l - 1].bqkv == s_bqkv && lg[l].bo - lg[l - 1].bo == s_bo &&
                 lg[l].b1 - lg[l - 1].b1 == s_b1 && lg[l].b2 - lg[l - 1].b2 == s_b2,
             "encoder_bwd: gradient blocks of layer %d are not at a uniform stride", l);
    }
  }
  char* base = (char*)arena;
  t_split_ws = lay.split_ws_floats ? (float*)(base + lay.split_ws) : nullptr;
  t_split_ws_floats = lay.split_ws_floats;
  const BwdLayout bl = pk ? bwd_layout_m(c, (size_t)M, B, 32) : bwd_layout(c, B, L);
  char* bb = base + lay.bwd_scratch;
  uint16_t* hidden = (uint16_t*)(base + lay.hidden);
  uint16_t* dy2_all = (uint16_t*)(bb + bl.dy2);
  uint16_t* du_all = (uint16_t*)(bb + bl.du);
  uint16_t* dy1_all = (uint16_t*)(bb + bl.dy1);
  uint16_t* dqkv_all = (uint16_t*)(bb + bl.dqkv);
  uint16_t* dxa = (uint16_t*)(bb + bl.dxa);
  uint16_t* dxb = (uint16_t*)(bb + bl.dxb);
  uint16_t* dctx = (uint16_t*)(bb + bl.dctx);
  uint16_t* dres = (uint16_t*)(bb + bl.dres);
  const bool dropping = drop_active(c);  // the forward of this arena ran with the same config (header contract)
  float* ln_partial = (float*)(bb + bl.ln_partial);
  float* cs_partial = (float*)(bb + bl.colsum_partial);
  float* emb_partial = (float*)(bb + bl.emb_partial);

  // a continuation range (d_in == NULL) picks up the gradient the previous range left in the arena
  const uint16_t* dx = d_in ? d_in : dxb;
  const int NG = layer_hi - layer_lo;  // layers in this range

  // Deferred reductions: the LayerNorm backward and the two GEMMs with fused column sums leave per-layer partial rows
  // in the arena and ONE batched reduction per group finishes them for the whole range (instead of four short
  // launches per layer).  Needs the vector gradients of a layer group to sit at one common layer stride (true for the
  // flat parameter layout); otherwise every call reduces immediately.
  hipStream_t hst = (hipStream_t)stream;
  const int P_ln = cocodr_ln_bwd_blocks(M);
  const size_t ln_slot = (size_t)P_ln * 3 * H;
  float* ln2_slots = (float*)(bb + bl.ln2_slots);
  float* ln1_slots = (float*)(bb + bl.ln1_slots);
  float* b1_slots = (float*)(bb + bl.b1_slots);
  float* bv_slots = (float*)(bb + bl.bv_slots);
  float* bqk_slots = (float*)(bb + bl.bqk_slots);
  const size_t bqk_slot = (size_t)4 * B * 2 * H;
  long long s_vec = 0;
  bool defer = NG > 1;
  if (defer) {
    const cocodr_layer_grads &x0 = lg[layer_lo], &x1 = lg[layer_lo + 1];
    s_vec = x1.b2 - x0.b2;
    defer = (x1.ln2_g - x0.ln2_g == s_vec) && (x1.ln2_b - x0.ln2_b == s_vec) && (x1.ln1_g - x0.ln1_g == s_vec) &&
            (x1.ln1_b - x0.ln1_b == s_vec) && (x1.bo - x0.bo == s_vec) && (x1.b1 - x0.b1 == s_vec) && (x1.bqkv - x0.bqkv == s_vec);
    for (int l = layer_lo + 2; defer && l < layer_hi; ++l)
      defer = (lg[l].ln2_g - lg[l - 1].ln2_g == s_vec) && (lg[l].ln2_b - lg[l - 1].ln2_b == s_vec) &&
              (lg[l].ln1_g - lg[l - 1].ln1_g == s_vec) && (lg[l].ln1_b - lg[l - 1].ln1_b == s_vec) &&
              (lg[l].bo - lg[l - 1].bo == s_vec) && (lg[l].b2 - lg[l - 1].b2 == s_vec) && (lg[l].b1 - lg[l - 1].b1 == s_vec) &&
              (lg[l].bqkv - lg[l - 1].bqkv == s_vec);
  }
  // the value-bias shortcut (sum_k dV[k] = sum_q dctx[q]) needs softmax rows that sum to 1: with dropout on the
  // probabilities the bias gradient is the column sum of dV itself
  const bool drop_probs = dropping && c->attn_dropout > 0.f;
  int rows_b1 = 0, rows_bv = 0;
  if (defer) {
    cocodr_gemm_args q = gemm_base(dy2_all, lp[layer_lo].w2, du_all, M, I, H, H, I, I, 0, 1);
    rows_b1 = cocodr_gemm_colsum_rows(&q);
    q = gemm_base(dy1_all, lp[layer_lo].wo, dctx, M, H, H, H, H, H, 0, 1);
    rows_bv = drop_probs ? 0 : cocodr_gemm_colsum_rows(&q);
  }
  for (int l = layer_hi - 1; l >= layer_lo; --l) {
    const cocodr_layer_params& w = lp[l];
    const cocodr_layer_grads& gr = lg[l];
    const size_t lo = (size_t)l;
    const uint16_t* qkv = (const uint16_t*)(base + lay.qkv) + lo * M * 3 * H;
    const uint16_t* ctx = (const uint16_t*)(base + lay.ctx) + lo * M * H;
    const uint16_t* y1 = (const uint16_t*)(base + lay.y1) + lo * M * H;
    const uint16_t* u = (const uint16_t*)(base + lay.u) + lo * M * I;
    const uint16_t* y2 = (const uint16_t*)(base + lay.y2) + lo * M * H;
    const float* lse = (const float*)(base + lay.lse) + lo * M * c->heads;
    const float* mean1 = (const float*)(base + lay.mean1) + lo * M;
    const float* rstd1 = (const float*)(base + lay.rstd1) + lo * M;
    const float* mean2 = (const float*)(base + lay.mean2) + lo * M;
    const float* rstd2 = (const float*)(base + lay.rstd2) + lo * M;
    uint16_t* dy2 = dy2_all + lo * M * H;
    uint16_t* du = du_all + lo * M * I;
    uint16_t* dy1 = dy1_all + lo * M * H;
    uint16_t* dqkv = dqkv_all + lo * M * 3 * H;

    // bias gradients ride on the kernels that produce the matrices they sum: b2 / bo on the LayerNorm backward, b1 on the
    // GELU' epilogue, and the value bias on the context-gradient GEMM (sum_k dV[k] = sum_q dctx[q]: softmax rows sum to 1)
    const size_t li = (size_t)(l - layer_lo);
    // With dropout the LayerNorm input was dropout(dense) + residual: dy2 / dy1 (what the dgrad / grouped wgrad GEMMs and the
    // bias sums take) hold the masked gradient of the dense output, the un-masked one goes to dres for the residual add.
    LayerDrop ld;
    TRY(layer_drop(c, dropping, l, &ld));
    const bool dh = ld.ffn_out.threshold != 0;  // hidden dropout on (both dense sites share the probability)
    // [CLS] tail (cocodr_config.cls_tail): the top layer's FFN / LayerNorms / output projection ran on the B [CLS] rows only - their
    // gradients live in the first B rows of the layer's slots, dx (d_in) is the [B,H] gradient of those rows, and the layer
    // reduces its vector gradients at once (its partial rows do not have the range's common shape)
    const bool tail = c->cls_tail != 0 && l == NL - 1;
    const int Mr = tail ? B : M;
    const bool dl = defer && !tail;
    uint16_t* res2 = dh ? dres : dy2;
    if (dl) TRY(cocodr_ln_bwd_partials(dx, y2, w.ln2_g, mean2, rstd2, res2, ln2_slots + li * ln_slot, Mr, H, 3, hst, dy2, &ld.ffn_out));
    else TRY(cocodr_ln_bwd_drop(dx, y2, w.ln2_g, mean2, rstd2, res2, dy2, gr.ln2_g, gr.ln2_b, gr.b2, ln_partial, Mr, H, &ld.ffn_out, stream));
    cocodr_gemm_args g = gemm_base(dy2, w.w2, du, Mr, I, H, H, I, I, 0, 1);  // dh = dy2 W2, fused with GELU'(u)
    g.epi = COCODR_EPI_DGELU; g.R = u; g.ldr = I;
    if (dl && rows_b1 > 0) g.colsum_partial = b1_slots + li * rows_b1 * I;
    else { g.colsum = gr.b1; g.colsum_partial = cs_partial; }
    TRY(cocodr_gemm(&g, stream));
    g = gemm_base(du, w.w1, dxa, Mr, H, I, I, H, H, 0, 1);  // dx1 = du W1 + dy2 (residual branch)
    g.epi = COCODR_EPI_ADD; g.R = res2; g.ldr = H;
    TRY(cocodr_gemm(&g, stream));
    uint16_t* res1 = dh ? dres : dy1;  // res2 has been consumed by the GEMM above (stream order)
    if (dl) TRY(cocodr_ln_bwd_partials(dxa, y1, w.ln1_g, mean1, rstd1, res1, ln1_slots + li * ln_slot, Mr, H, 3, hst, dy1, &ld.attn_out));
    else TRY(cocodr_ln_bwd_drop(dxa, y1, w.ln1_g, mean1, rstd1, res1, dy1, gr.ln1_g, gr.ln1_b, gr.bo, ln_partial, Mr, H, &ld.attn_out, stream));
    uint16_t* dctx_rows = tail ? dxa : dctx;  // (tail: dxa is free again - B compact rows, scattered into the zeroed dctx below)
    g = gemm_base(dy1, w.wo, dctx_rows, Mr, H, H, H, H, H, 0, 1);  // dctx = dy1 Wo
    if (dl && rows_bv > 0) g.colsum_partial = bv_slots + li * rows_bv * H;
    else if (!drop_probs) { g.colsum = gr.bqkv + 2 * H; g.colsum_partial = cs_partial; }
    TRY(cocodr_gemm(&g, stream));
    if (tail) {
      // the attention backward and the QKV dgrad below see the layer as a whole again: context gradient and residual gradient are
      // zero off the [CLS] rows (no dropout on this path, so dres is free)
      const TailScratch ts = tail_scratch(hidden + (size_t)NL * M * H, B, H);
      if (hipMemsetAsync(dctx, 0, (size_t)M * H * 2, hst) != hipSuccess || hipMemsetAsync(dres, 0, (size_t)M * H * 2, hst) != hipSuccess) {
        cocodr_set_error("encoder_bwd: memset failed");
        return COCODR_ERR_LAUNCH;
      }
      TRY(cocodr_scatter_rows(dxa, ts.idx, dctx, B, H, 0, stream));
      TRY(cocodr_scatter_rows(dy1, ts.idx, dres, B, H, 0, stream));
      res1 = dres;
      // this layer's three weight gradients over the B rows (the grouped launch below takes the other layers of the range)
      cocodr_gemm_args tw = gemm_base(dy1, ts.ctx_c, gr.wo, H, H, B, H, H, H, 1, 1);
      tw.out_f32 = 1;
      TRY(cocodr_gemm(&tw, stream));
      tw = gemm_base(du, (const uint16_t*)(base + lay.x1) + lo * M * H, gr.w1, I, H, B, I, H, H, 1, 1);
      tw.out_f32 = 1;
      TRY(cocodr_gemm(&tw, stream));
      tw = gemm_base(dy2, (const uint16_t*)(base + lay.h) + lo * M * I, gr.w2, H, I, B, H, I, I, 1, 1);
      tw.out_f32 = 1;
      TRY(cocodr_gemm(&tw, stream));
    }
    // the query / key bias gradients are column sums of dQ | dK: the attention backward leaves four partial rows per sequence
    if (pk) TRY(cocodr_attn_bwd_packed(qkv, mask, ctx, dctx, lse, dqkv, bqk_slots + li * bqk_slot, pk->seq_off, pk->seq_order, B, M, pk->max_len, c->heads,
                                       &ld.probs, pk->drop_L, stream));
    else TRY(cocodr_attn_bwd_drop(qkv, mask, ctx, dctx, lse, dqkv, bqk_slots + li * bqk_slot, B, L, c->heads, &ld.probs, stream));
    if (!defer) TRY(cocodr_reduce_partials(bqk_slots + li * bqk_slot, gr.bqkv, nullptr, nullptr, 4 * B, 1, 2 * H, 1, 0, hst));
    if (drop_probs) TRY(cocodr_colsum(dqkv + 2 * H, gr.bqkv + 2 * H, cs_partial, M, H, 3 * H, 1, 0, 0, stream));
    g = gemm_base(dqkv, w.wqkv, dxb, M, H, 3 * H, 3 * H, H, H, 0, 1);  // dx = dqkv Wqkv + dy1 (residual branch)
    g.epi = COCODR_EPI_ADD; g.R = res1; g.ldr = H;
    TRY(cocodr_gemm(&g, stream));
    dx = dxb;
  }
  if (do_embed) {
    CK_ARG(layer_lo == 0, "encoder_bwd: the embedding backward belongs to the range that ends at layer 0");
    cocodr_dropout_mask de;
    TRY(cocodr_dropout_mask_for(dropping ? c->hidden_dropout : 0.0, c->drop_seed, c->drop_call, 0, COCODR_DROP_EMBED, &de));
    if (pk) {
      CK_ARG(pk->ids, "encoder_bwd(packed): the embedding needs the token ids");
      TRY(cocodr_embed_ln_bwd_packed(dx, pk->ids, pk->seq_off, emb->word, emb->pos, emb->type0, emb->ln_g, (const float*)(base + lay.emb_mean),
                                     (const float*)(base + lay.emb_rstd), eg->word, eg->pos, eg->type0, eg->ln_g, eg->ln_b, emb_partial, B, M,
                                     pk->max_len, H, c->vocab, &de, stream));
    } else {
      TRY(cocodr_embed_ln_bwd_drop(dx, ids, emb->word, emb->pos, emb->type0, emb->ln_g, (const float*)(base + lay.emb_mean),
                                   (const float*)(base + lay.emb_rstd), eg->word, eg->pos, eg->type0, eg->ln_g, eg->ln_b, emb_partial,
                                   B, L, H, c->vocab, &de, stream));
    }
  }
  if (NG == 0) return COCODR_OK;

  // ---- grouped weight gradients of this range: one batched TN launch per matrix, batch = layer
  const long long sMH = (long long)M * H, sMI = (long long)M * I, sM3H = (long long)M * 3 * H;
  const size_t l0 = (size_t)layer_lo;
  const cocodr_layer_grads& g0 = lg[layer_lo];
  // (the four matrices of the range in ONE launch where the pipeline allows - cocodr_gemm_multi: launched one after the other,
  // each of the four pays its own partial last round of the 256 CUs)
  // ([CLS] tail: the top layer computed dWo, dW1, dW2 and its vector gradients itself - those three problems and the deferred
  //  LayerNorm / b1 / value-bias jobs cover the range's other layers, which come first in it)
  const int NGt = (c->cls_tail != 0 && layer_hi == NL) ? NG - 1 : NG;
  cocodr_gemm_args wg[4];
  wg[0] = gemm_base(dqkv_all + l0 * sM3H, hidden + l0 * sMH, g0.wqkv, 3 * H, H, M, 3 * H, H, H, 1, 1);
  wg[0].out_f32 = 1; wg[0].batch = NG; wg[0].strideA = sM3H; wg[0].strideB = sMH; wg[0].strideC = s_wqkv;
  wg[1] = gemm_base(dy1_all + l0 * sMH, (const uint16_t*)(base + lay.ctx) + l0 * sMH, g0.wo, H, H, M, H, H, H, 1, 1);
  wg[1].out_f32 = 1; wg[1].batch = NGt; wg[1].strideA = sMH; wg[1].strideB = sMH; wg[1].strideC = s_wo;
  wg[2] = gemm_base(du_all + l0 * sMI, (const uint16_t*)(base + lay.x1) + l0 * sMH, g0.w1, I, H, M, I, H, H, 1, 1);
  wg[2].out_f32 = 1; wg[2].batch = NGt; wg[2].strideA = sMI; wg[2].strideB = sMH; wg[2].strideC = s_w1;
  wg[3] = gemm_base(dy2_all + l0 * sMH, (const uint16_t*)(base + lay.h) + l0 * sMI, g0.w2, H, I, M, H, I, I, 1, 1);
  wg[3].out_f32 = 1; wg[3].batch = NGt; wg[3].strideA = sMH; wg[3].strideB = sMI; wg[3].strideC = s_w2;
  TRY(cocodr_gemm_multi(wg, NGt > 0 ? 4 : 1, bl.multi_ws_floats ? (float*)(bb + bl.multi_ws) : nullptr, bl.multi_ws_floats, stream));
  // ---- deferred reductions of the range (LayerNorm weight / bias + the Linear bias in front of it; b1; value bias)
  if (defer) {
    cocodr_reduce_job jobs[5];  // one launch for all of them
    int nj = 0;
    if (NGt > 0) {
      jobs[nj++] = {ln2_slots, g0.ln2_g, g0.ln2_b, g0.b2, P_ln, 3, H, NGt, s_vec};
      jobs[nj++] = {ln1_slots, g0.ln1_g, g0.ln1_b, g0.bo, P_ln, 3, H, NGt, s_vec};
      if (rows_b1 > 0) jobs[nj++] = {b1_slots, g0.b1, nullptr, nullptr, rows_b1, 1, I, NGt, s_vec};
      if (rows_bv > 0) jobs[nj++] = {bv_slots, g0.bqkv + 2 * H, nullptr, nullptr, rows_bv, 1, H, NGt, s_vec};
    }
    jobs[nj++] = {bqk_slots, g0.bqkv, nullptr, nullptr, 4 * B, 1, 2 * H, NG, s_vec};
    TRY(cocodr_reduce_partials_multi(jobs, nj, hst));
  }
  return COCODR_OK;
}
}  // namespace

extern "C" int cocodr_encoder_bwd(const cocodr_config* c, const cocodr_embed_params* emb, const cocodr_layer_params* lp,
                                  const cocodr_embed_grads* eg, const cocodr_layer_grads* lg, const int32_t* ids,
                                  const int32_t* mask, const uint16_t* d_last, int B, int L, void* arena, size_t arena_bytes,
                                  cocodr_stream_t stream) {
  CK_ARG(c && d_last, "encoder_bwd: null pointer");
  return cocodr_encoder_bwd_range(c, emb, lp, eg, lg, ids, mask, d_last, B, L, arena, arena_bytes, c->layers, 0, 1, stream);
}
