"""ctypes binding of libcocodr_hip.so (include/cocodr.h).  This is the only place the product touches
native code; there is NO fallback: if the library is missing or fails to load, every op raises."""
from __future__ import annotations

import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcocodr_hip.so")

c_void_p, c_int, c_float, c_size_t, c_longlong, c_double = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_longlong, C.c_double


class DropoutMask(C.Structure):  # mirrors cocodr_dropout_mask
    _fields_ = [("k0", C.c_uint32), ("k1", C.c_uint32), ("threshold", C.c_uint32), ("scale", c_float)]


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("C2", c_void_p), ("bias", c_void_p), ("R", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("lda", c_int), ("ldb", c_int), ("ldc", c_int), ("ldr", c_int),
        ("trans_a", c_int), ("trans_b", c_int), ("epi", c_int), ("out_f32", c_int),
        ("batch", c_int),
        ("strideA", c_longlong), ("strideB", c_longlong), ("strideC", c_longlong), ("strideR", c_longlong),
        ("strideBias", c_longlong),
        ("colsum", c_void_p), ("colsum_partial", c_void_p),
        ("drop", DropoutMask),
        ("ab_f16", c_int),
        ("split_ws", c_void_p), ("split_ws_floats", c_size_t),
    ]


class Config(C.Structure):
    _fields_ = [("hidden", c_int), ("heads", c_int), ("layers", c_int), ("inter", c_int), ("vocab", c_int),
                ("max_pos", c_int), ("ln_eps", c_float),
                ("hidden_dropout", c_float), ("attn_dropout", c_float), ("drop_seed", C.c_ulonglong), ("drop_call", C.c_ulonglong),
                ("cls_tail", c_int)]


class LayerParams(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("wqkv", "wo", "w1", "w2", "bqkv", "bo", "b1", "b2", "ln1_g", "ln1_b", "ln2_g", "ln2_b")]


class LayerGrads(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("wqkv", "wo", "w1", "w2", "bqkv", "bo", "b1", "b2", "ln1_g", "ln1_b", "ln2_g", "ln2_b")]


class EmbedParams(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("word", "pos", "type0", "ln_g", "ln_b")]


class EmbedGrads(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("word", "pos", "type0", "ln_g", "ln_b")]


class EncoderLayout(C.Structure):
    _fields_ = [(n, c_size_t) for n in (
        "total_bytes", "hidden", "cls_f32", "qkv", "ctx", "y1", "x1", "u", "h", "y2", "lse", "mean1", "rstd1", "mean2",
        "rstd2", "emb_mean", "emb_rstd", "bwd_scratch", "bwd_bytes", "bwd_dx", "split_ws", "split_ws_floats")]


class PackedBatch(C.Structure):  # mirrors cocodr_packed_batch
    _fields_ = ([(n, c_void_p) for n in ("ids", "positions", "mask", "seq_off", "cls_slot")] + [(n, c_int) for n in ("B", "T", "max_len", "drop_L")]
                + [("seq_order", c_void_p)])


class LambPlan(C.Structure):  # mirrors cocodr_lamb_plan
    _fields_ = [("chunk_start", c_void_p), ("chunk_len", c_void_p), ("chunk_seg", c_void_p), ("seg_chunk_begin", c_void_p),
                ("nchunk", c_int), ("nseg", c_int)]


class LambFusedPlan(C.Structure):  # mirrors cocodr_lamb_fused_plan
    _fields_ = [("seg_start", c_void_p), ("seg_len", c_void_p), ("seg_index", c_void_p), ("wg_begin", c_void_p), ("wg_count", c_void_p),
                ("round_first", c_void_p), ("nfused", c_int), ("nrounds", c_int)]


class EncoderBwdLayout(C.Structure):  # mirrors cocodr_encoder_bwd_layout_t
    _fields_ = [(n, c_size_t) for n in ("dy2", "du", "dy1", "dqkv", "ln2_partial", "ln1_partial")] + [("ln_blocks", c_int), ("ln_rows", c_int)]


EPI_NONE, EPI_GELU, EPI_ADD, EPI_DGELU = 0, 1, 2, 3

# name -> (restype, argtypes); every symbol include/cocodr.h declares
SIGNATURES = {
    "cocodr_last_error": (C.c_char_p, []),
    "cocodr_build_info": (C.c_char_p, []),
    "cocodr_gemm": (c_int, [C.POINTER(GemmArgs), c_void_p]),
    "cocodr_gemm_multi_workspace_floats": (c_size_t, []),
    "cocodr_gemm_split_workspace_floats": (c_size_t, []),
    "cocodr_gemm_multi_workspace_floats_for": (c_size_t, [c_void_p, c_int]),
    "cocodr_gemm_multi": (c_int, [C.POINTER(GemmArgs), c_int, c_void_p, c_size_t, c_void_p]),
    "cocodr_gemm_set_impl": (c_int, [c_int]),
    "cocodr_attn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "cocodr_attn_bwd": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_void_p]),
    "cocodr_dropout_mask_for": (c_int, [c_double, C.c_ulonglong, C.c_ulonglong, c_int, c_int, C.POINTER(DropoutMask)]),
    "cocodr_attn_fwd_drop": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, C.POINTER(DropoutMask), c_void_p]),
    "cocodr_attn_bwd_drop": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, C.POINTER(DropoutMask), c_void_p]),
    "cocodr_embed_ln_fwd_drop": (c_int, [c_void_p] * 9 + [c_int, c_int, c_int, c_int, c_float, C.POINTER(DropoutMask), c_void_p]),
    "cocodr_embed_ln_bwd_drop": (c_int, [c_void_p] * 14 + [c_int, c_int, c_int, c_int, C.POINTER(DropoutMask), c_void_p]),
    "cocodr_ln_bwd_drop": (c_int, [c_void_p] * 11 + [c_int, c_int, C.POINTER(DropoutMask), c_void_p]),
    "cocodr_attn_fwd_packed": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_int, C.POINTER(DropoutMask), c_int, c_void_p]),
    "cocodr_attn_bwd_packed": (c_int, [c_void_p] * 9 + [c_int, c_int, c_int, c_int, C.POINTER(DropoutMask), c_int, c_void_p]),
    "cocodr_embed_ln_fwd_packed": (c_int, [c_void_p] * 10 + [c_int, c_int, c_int, c_float, C.POINTER(DropoutMask), c_void_p]),
    "cocodr_embed_bwd_packed_partial_floats": (c_size_t, [c_int, c_int]),
    "cocodr_embed_ln_bwd_packed": (c_int, [c_void_p] * 15 + [c_int, c_int, c_int, c_int, c_int, C.POINTER(DropoutMask), c_void_p]),
    "cocodr_mask_lengths": (c_int, [c_void_p, c_int, c_int, c_int, c_longlong, c_void_p, c_void_p, c_void_p]),
    "cocodr_pack_plan": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "cocodr_pack_index": (c_int, [c_void_p, c_int, c_longlong, c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 5 + [c_void_p]),
    "cocodr_ln_fwd_slots": (c_int, [c_void_p] * 7 + [c_int, c_void_p, c_int, c_int, c_float, c_void_p]),
    "cocodr_encoder_layout_packed": (c_int, [C.POINTER(Config), c_int, c_int, c_int, C.POINTER(EncoderLayout)]),
    "cocodr_encoder_fwd_packed": (c_int, [C.POINTER(Config), C.POINTER(EmbedParams), C.POINTER(LayerParams), C.POINTER(PackedBatch), c_int,
                                          c_void_p, c_size_t, c_void_p]),
    "cocodr_stack_fwd_packed": (c_int, [C.POINTER(Config), C.POINTER(LayerParams), C.POINTER(PackedBatch), c_int, c_void_p, c_size_t, c_void_p]),
    "cocodr_encoder_bwd_packed": (c_int, [C.POINTER(Config), C.POINTER(EmbedParams), C.POINTER(LayerParams), C.POINTER(EmbedGrads),
                                          C.POINTER(LayerGrads), C.POINTER(PackedBatch), c_void_p, c_void_p, c_size_t, c_int, c_int, c_int,
                                          c_void_p]),
    "cocodr_embed_ln_fwd": (c_int, [c_void_p] * 9 + [c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "cocodr_embed_bwd_partial_floats": (c_size_t, [c_int, c_int]),
    "cocodr_embed_ln_bwd": (c_int, [c_void_p] * 14 + [c_int, c_int, c_int, c_int, c_void_p]),
    "cocodr_ln_fwd": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_float, c_void_p]),
    "cocodr_ln_bwd_partial_floats": (c_size_t, [c_int, c_int]),
    "cocodr_ln_bwd": (c_int, [c_void_p] * 10 + [c_int, c_int, c_void_p]),
    "cocodr_gemm_colsum_partial_floats": (c_size_t, [c_int, c_int]),
    "cocodr_gemm_colsum_rows": (c_int, [C.POINTER(GemmArgs)]),
    "cocodr_colsum_partial_floats": (c_size_t, [c_int, c_int, c_int]),
    "cocodr_gram_f32_workspace_floats": (c_size_t, [c_int, c_longlong]),
    "cocodr_gram_f32": (c_int, [c_void_p, c_longlong, c_int, c_longlong, c_void_p, c_void_p, c_void_p]),
    "cocodr_colsum": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_longlong, c_longlong, c_void_p]),
    "cocodr_cast_f32_bf16": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "cocodr_zero_f32": (c_int, [c_void_p, c_size_t, c_void_p]),
    "cocodr_gather_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "cocodr_scatter_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "cocodr_mul_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "cocodr_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_float, c_float, c_float,
                                  c_float, c_float, c_int, c_float, c_void_p, c_void_p]),
    "cocodr_grad_norm_clip": (c_int, [C.POINTER(c_void_p), C.POINTER(c_size_t), c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "cocodr_lamb_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, C.POINTER(LambPlan), c_float,
                                 c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cocodr_lamb_fused_capacity": (c_size_t, []),
    "cocodr_lamb_fused_workgroups": (c_int, []),
    "cocodr_lamb_fused_workgroup_elements": (c_size_t, []),
    "cocodr_lamb_fused_workspace_floats": (c_size_t, [c_int]),
    "cocodr_lamb_fused_error_index": (c_size_t, [c_int]),
    "cocodr_lamb_step_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, C.POINTER(LambFusedPlan), c_float,
                                       c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cocodr_scatter_cls_grad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "cocodr_cls_rows": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "cocodr_simce_workspace_floats": (c_size_t, [c_int]),
    "cocodr_simce_fwd_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cocodr_allgather_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "cocodr_triplet_nll_fwd_bwd": (c_int, [c_void_p] * 4 + [c_int, c_int] + [c_void_p] * 6 + [c_void_p]),
    "cocodr_score_topk_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "cocodr_score_topk_workspace_bytes_dim": (c_size_t, [c_int, c_int, c_int, c_int]),
    "cocodr_score_set_mode": (c_int, [c_int]),
    "cocodr_score_filter_plan": (c_int, [c_int, c_int, c_int, c_int, c_void_p]),
    "cocodr_score_topk": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_longlong, c_void_p, c_void_p, c_void_p,
                                  c_size_t, c_void_p]),
    "cocodr_score_topk_resident": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_longlong, c_void_p, c_void_p, c_void_p,
                                           c_size_t, c_int, c_void_p]),
    "cocodr_topk_merge": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_longlong, c_void_p, c_void_p, c_int, c_void_p]),
    "cocodr_encoder_layout": (c_int, [C.POINTER(Config), c_int, c_int, c_int, C.POINTER(EncoderLayout)]),
    "cocodr_encoder_bwd_layout": (c_int, [C.POINTER(Config), c_int, c_int, C.POINTER(EncoderBwdLayout)]),
    "cocodr_encoder_fwd": (c_int, [C.POINTER(Config), C.POINTER(EmbedParams), C.POINTER(LayerParams), c_void_p, c_void_p,
                                   c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "cocodr_encoder_fwd_range": (c_int, [C.POINTER(Config), C.POINTER(EmbedParams), C.POINTER(LayerParams), c_void_p, c_void_p,
                                         c_int, c_int, c_int, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "cocodr_stack_fwd": (c_int, [C.POINTER(Config), C.POINTER(LayerParams), c_void_p, c_int, c_int, c_int, c_void_p, c_size_t,
                                 c_void_p]),
    "cocodr_ce_fwd_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "cocodr_encoder_bwd": (c_int, [C.POINTER(Config), C.POINTER(EmbedParams), C.POINTER(LayerParams), C.POINTER(EmbedGrads),
                                   C.POINTER(LayerGrads), c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t,
                                   c_void_p]),
    "cocodr_encoder_bwd_range": (c_int, [C.POINTER(Config), C.POINTER(EmbedParams), C.POINTER(LayerParams),
                                         C.POINTER(EmbedGrads), C.POINTER(LayerGrads), c_void_p, c_void_p, c_void_p, c_int, c_int,
                                         c_void_p, c_size_t, c_int, c_int, c_int, c_void_p]),
    "cocodr_mlm_collate": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_double,
                                   C.c_ulonglong, c_longlong, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cocodr_prof_begin": (c_int, [c_int]),
    "cocodr_prof_pause": (c_int, [c_int]),
    "cocodr_prof_end": (c_int, [C.POINTER(c_int), C.POINTER(c_double), C.POINTER(c_double)]),
    "cocodr_prof_event_overhead_us": (c_int, [c_void_p, C.POINTER(c_double)]),
    "cocodr_probe_mfma32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "cocodr_probe_tr16": (c_int, [c_void_p, c_void_p, c_void_p]),
}

_lib = None
_lock = threading.Lock()


class NativeLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the native library.  Raises NativeLibraryError - never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU / PyTorch fallback for this path.")
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()  # torch's HIP runtime first: a fat binary registered before it leaves launches without a device
        except ImportError:
            pass
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:  # e.g. no ROCm runtime on this machine
            raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
        return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().cocodr_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise RuntimeError(f"{what}: {msg} (code {rc})")


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())
