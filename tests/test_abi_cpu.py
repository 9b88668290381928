"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every
symbol include/cocodr.h declares (no compute calls here - there is no GPU in this container); host-side
validation raises the documented errors; the product never imports the oracle."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    import cocodr_amd
    from cocodr_amd import _native
    return _native.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "cocodr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cocodr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(lib):
    from cocodr_amd import _native
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/cocodr.h but not exported"
        assert n in _native.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_native.SIGNATURES) <= set(names)
    assert b"gfx950" in lib.cocodr_build_info()


def test_library_has_no_unresolved_kernel_symbols():
    out = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(ROOT, "coco-dr_amd", "libcocodr_hip.so")],
                         capture_output=True, text=True).stdout
    assert not [l for l in out.splitlines() if "_kernel" in l or "cocodr" in l]


def test_argument_validation_without_gpu(lib):
    from cocodr_amd import _native as N
    g = N.GemmArgs()
    assert lib.cocodr_gemm(ctypes.byref(g), None) == -1 and b"null operand" in lib.cocodr_last_error()
    cfg = N.Config(100, 2, 2, 256, 1000, 64, 1e-12)  # hidden != heads*64
    lay = N.EncoderLayout()
    assert lib.cocodr_encoder_layout(ctypes.byref(cfg), 4, 32, 1, ctypes.byref(lay)) == -1
    cfg = N.Config(128, 2, 2, 256, 1000, 64, 1e-12)
    assert lib.cocodr_encoder_layout(ctypes.byref(cfg), 4, 33, 1, ctypes.byref(lay)) == -1  # L % 32
    assert lib.cocodr_encoder_layout(ctypes.byref(cfg), 4, 32, 1, ctypes.byref(lay)) == 0
    assert lay.total_bytes > lay.bwd_scratch > lay.hidden >= 0 and lay.bwd_bytes > 0
    lay0 = N.EncoderLayout()
    assert lib.cocodr_encoder_layout(ctypes.byref(cfg), 4, 32, 0, ctypes.byref(lay0)) == 0
    assert lay0.total_bytes < lay.total_bytes and lay0.bwd_bytes == 0
    assert lib.cocodr_score_topk_workspace_bytes(100, 1000, 10) >= 100 * 1000 * 4


def test_filtered_search_plan_arithmetic(lib, monkeypatch):
    """cocodr_score_filter_plan is host arithmetic (include/cocodr.h): which searches are filtered, and with what sample / rank /
    block sizes; the workspace the library asks for grows by the plan's extra buffers and shrinks back with the switch."""
    from cocodr_amd import ops
    for h in ("COCODR_SCORE_NOFILTER", "COCODR_SCORE_FILTER_FORCE", "COCODR_SCORE_FILTER_J", "COCODR_SCORE_FILTER_CAPT"):
        monkeypatch.delenv(h, raising=False)
    p = ops.score_filter_plan(10000, 125000, 1024, 1000)  # one GPU's shard of config 5
    assert p["filtered"] == 1 and p["sample_passages"] == 4096 and p["sample_stride"] == 30
    mu = 1000 * 4096 / 125000
    assert p["threshold_rank"] == int(mu + 4 * mu ** 0.5 + 8)
    assert p["block_slots"] % 8 == 0 and 24 <= p["block_slots"] <= 64
    assert p["rows_per_pass"] >= 10000 and p["rows_per_pass"] % 256 == 0 and p["rows_per_exhaustive_pass"] == 4096
    assert ops.score_filter_plan(256, 1_000_000, 1024, 1000)["filtered"] == 1
    # few passages / few scores in all (the extra launches do not pay) / k too large a share / candidates past the list
    for nq, npass, k in [(5000, 20000, 100), (256, 125000, 100), (64, 1_000_000, 100), (3000, 40000, 4000), (3000, 40000, 3000), (16, 8_000_000, 1000)]:
        assert ops.score_filter_plan(nq, npass, 768, k)["filtered"] == 0
    monkeypatch.setenv("COCODR_SCORE_FILTER_FORCE", "1")
    assert ops.score_filter_plan(256, 125000, 768, 100)["filtered"] == 1 and ops.score_filter_plan(70, 5003, 64, 10)["filtered"] == 1
    assert ops.score_filter_plan(3000, 40000, 768, 3000)["filtered"] == 0  # (the statistical conditions are not a size rule)
    monkeypatch.delenv("COCODR_SCORE_FILTER_FORCE")
    with_filter = lib.cocodr_score_topk_workspace_bytes_dim(10000, 125000, 1024, 1000)
    monkeypatch.setenv("COCODR_SCORE_NOFILTER", "1")
    assert ops.score_filter_plan(10000, 125000, 1024, 1000)["filtered"] == 0
    without = lib.cocodr_score_topk_workspace_bytes_dim(10000, 125000, 1024, 1000)
    monkeypatch.delenv("COCODR_SCORE_NOFILTER")
    assert 0 < with_filter - without < 200 << 20
    ops.score_set_mode(1)
    try:
        assert ops.score_filter_plan(10000, 125000, 1024, 1000)["filtered"] == 0  # the exact fp32 pipeline is never filtered
    finally:
        ops.score_set_mode(0)


def test_ops_reject_cpu_tensors_and_model_refuses_cpu():
    from cocodr_amd import ops
    from cocodr_amd.modeling import CocoBertConfig, CocoBertModel
    with pytest.raises(ValueError, match="GPU"):
        ops.gemm(torch.zeros(128, 64, dtype=torch.bfloat16), torch.zeros(128, 64, dtype=torch.bfloat16))
    m = CocoBertModel(CocoBertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=100, hidden_size=128, num_hidden_layers=1, num_attention_heads=2,
                                     intermediate_size=128, max_position_embeddings=32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 32, dtype=torch.long))
    with pytest.raises(ValueError):
        CocoBertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, hidden_size=100, num_attention_heads=2)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from cocodr_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.NativeLibraryError, match="no CPU / PyTorch fallback"):
        _native.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "coco-dr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
    code = "import sys; sys.path.insert(0, %r); import cocodr_amd, cocodr_amd.ops, cocodr_amd.modeling; " \
           "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)" % ROOT
    subprocess.run([sys.executable, "-c", code], check=True)


def test_state_dict_uses_hf_bert_names_and_flat_layout_is_uniform():
    from cocodr_amd.modeling import CocoBertConfig, CocoBertModel
    cfg = CocoBertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=200, hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256,
                         max_position_embeddings=64)
    m = CocoBertModel(cfg)
    from transformers import BertConfig, BertModel
    hf = BertModel(BertConfig(vocab_size=200, hidden_size=128, num_hidden_layers=3, num_attention_heads=2,
                              intermediate_size=256, max_position_embeddings=64), add_pooling_layer=False)
    assert set(m.state_dict().keys()) == {k for k in hf.state_dict().keys() if "position_ids" not in k}
    for k, v in hf.state_dict().items():
        if "position_ids" not in k:
            assert tuple(m.state_dict()[k].shape) == tuple(v.shape), k
    # load an HF checkpoint (with the 'bert.' prefix MLM checkpoints carry) and read it back
    m.load_state_dict({"bert." + k: v for k, v in hf.state_dict().items()}, strict=False)
    for k, v in hf.state_dict().items():
        if "position_ids" not in k:
            assert torch.equal(m.state_dict()[k], v), k
    lo = m.layout
    q0 = lo.names["encoder.layer.0.attention.self.query.weight"][1]
    q1 = lo.names["encoder.layer.1.attention.self.query.weight"][1]
    q2 = lo.names["encoder.layer.2.attention.self.query.weight"][1]
    assert q1 - q0 == q2 - q1 == lo.mat_stride
    groups = m.param_groups(0.01)
    assert groups[0]["weight_decay"] == 0.01 and groups[1]["weight_decay"] == 0.0
    # parameters() yields every HF tensor once, as nn.Parameter views of the two flats (tests/test_named_params_cpu.py)
    names = [n for n, _ in m.named_parameters()]
    assert sorted(names) == sorted(lo.names) and len(m.flat_parameters()) == 2


def test_gemm_a4_build_keeps_the_accumulators_intact(tmp_path):
    """csrc/gemm_a4.hip reads its 256 accumulator registers by NUMBER in the epilogue (inline asm); hipcc may use accumulator registers
    as spill space wherever it believes them dead.  The loop statement declares them as outputs and every reading statement as pinned
    inputs, which gives hipcc the right liveness - this audit of the generated assembly is the check: behind the
    loop statement no kernel writes an accumulator register before the epilogue has read it, and NO variant touches scratch (the two
    column-sum variants did until their 256 additions were pinned in program order: the spilled registers were the next tile's
    fragments, and their reload waited for every store of the tile - 2.6 % of the BERT-large step's GEMM time)."""
    import re
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "coco-dr_amd", "csrc", "gemm_a4.hip")
    out = tmp_path / "gemm_a4.o"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "coco-dr_amd", "csrc"), "-c", src, "-o", str(out), "-save-temps=obj"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = (tmp_path / "gemm_a4-hip-amdgcn-amd-amdhsa-gfx950.s").read_text()
    parts = re.split(r"\n(_ZN14cocodr_gemm_a4[^\n:]*):", text)
    kernels = 0
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1].split("s_endpgm")[0]
        if "walk_kernel" not in name:
            continue
        kernels += 1
        lines = body.split("\n")
        end = max(k for k, l in enumerate(lines) if "s_nop 15" in l)   # the loop statement ends with two of them
        assert "scratch_" not in body, name
        read = set()
        for l in lines[end:]:
            m = re.search(r"v_accvgpr_read_b32 v\d+, a\[(0x[0-9a-f]+|\d+)(?:\+(\d+))?\]", l)
            if m:
                read.add(int(m.group(1), 0) + int(m.group(2) or 0))
            w = re.search(r"v_accvgpr_(?:write_b32|mov_b32) a(\d+)", l)
            if w:
                assert int(w.group(1)) in read, (name, l.strip())
        assert len(read) == 256, (name, len(read))
    assert kernels >= 20
