"""cocodr_amd - MI355X (gfx950) native implementation of the COCO-DR contrastive dense-retrieval hot
path: BERT bi-encoder forward/backward, in-batch contrastive / triplet losses and brute-force
inner-product search, as hand-written HIP kernels behind the HuggingFace-style model API the
reference calls.  Import name: ``cocodr_amd`` (directory ``coco-dr_amd/``)."""
from . import _native  # noqa: F401
from ._native import NativeLibraryError  # noqa: F401

__all__ = ["ops", "NativeLibraryError"]


def __getattr__(name):  # lazy: torch is only imported when the tensor-level API is used
    import importlib
    if name in ("ops", "modeling", "retrieval", "optim", "condenser", "masked_lm", "data"):
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
