import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def cfg_from_golden(g):
    from oracle import OracleConfig
    v = [int(x) for x in g["cfg"]]
    return OracleConfig(vocab_size=v[0], hidden_size=v[1], num_hidden_layers=v[2], num_attention_heads=v[3],
                        intermediate_size=v[4], max_position_embeddings=v[5], type_vocab_size=v[6])


def params_from_golden(g, dtype=np.float32):
    """The seeded parameters a fixture was generated with (tests/golden/make_golden.py): oracle.make_params(seed, std),
    and - for the triplet / DRO fixtures - the last LayerNorm's gain and bias scaled by ``final_ln_scale`` so that the
    dot-product logits are O(5) instead of O(H)."""
    from oracle import make_params
    cfg = cfg_from_golden(g)
    P = make_params(cfg, int(g["seed"]), std=float(g["std"]))
    if "final_ln_scale" in g.files:
        s = float(g["final_ln_scale"])
        last = f"encoder.layer.{cfg.num_hidden_layers - 1}.output.LayerNorm."
        for k in (last + "weight", last + "bias"):
            P[k] = (P[k] * s).astype(P[k].dtype)
    return {k: v.astype(dtype) for k, v in P.items()}


@pytest.fixture(scope="session")
def golden_coco():
    return load_golden("coco_contrastive_tiny.npz")


@pytest.fixture(scope="session")
def golden_ance():
    return load_golden("ance_triplet_tiny.npz")


@pytest.fixture(scope="session")
def golden_loss():
    return load_golden("contrastive_loss.npz")
