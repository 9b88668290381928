"""ctypes front end of oracle/cocodr_ref.c - the CPU twins of the C ABI (``*_ref``: the signatures of include/cocodr.h on host
pointers).  TEST INFRASTRUCTURE ONLY (oracle/__init__.py).  ``build()`` compiles the C file with gcc next to it
(``oracle/libcocodr_ref.so``, git-ignored, travels to the GPU box with the snapshot); ``lib()`` loads it and declares the
signatures from the product's own table (``cocodr_amd._native.SIGNATURES``), so a twin whose argument list drifts from its device
function fails to bind."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cocodr_ref.c")
LIB = os.path.join(HERE, "libcocodr_ref.so")
TWINS = ["cocodr_gemm", "cocodr_ln_fwd", "cocodr_ln_bwd", "cocodr_attn_fwd", "cocodr_attn_bwd", "cocodr_embed_ln_fwd", "cocodr_embed_ln_bwd",
         "cocodr_simce_fwd_bwd", "cocodr_triplet_nll_fwd_bwd", "cocodr_score_topk", "cocodr_topk_merge"]
_lib = None


def build(force: bool = False) -> str:
    hdr = os.path.join(os.path.dirname(HERE), "include", "cocodr.h")
    if force or not os.path.exists(LIB) or any(os.path.getmtime(f) > os.path.getmtime(LIB) for f in (SRC, hdr)):
        subprocess.run(["gcc", "-O2", "-std=c11", "-shared", "-fPIC", "-Wall", "-o", LIB, SRC, "-lm"], check=True)
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        import cocodr_amd  # noqa: F401
        from cocodr_amd._native import SIGNATURES
        for name in TWINS:
            restype, argtypes = SIGNATURES[name]
            fn = getattr(_lib, name + "_ref")
            fn.restype, fn.argtypes = restype, argtypes
    return _lib
