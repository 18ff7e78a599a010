"""CPU: the C-ABI library builds, loads without a GPU, exports every symbol
include/slnlp.h declares, and its arena layout is the reference state_dict."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from slnlp import _lib
    return _lib.load()


def header_symbols():
    src = open(os.path.join(ROOT, "include", "slnlp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(slnlp_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from slnlp import _lib
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in slnlp.h but not exported by libslnlp.so"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in slnlp/_lib.py"
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.slnlp_abi_version() == 1


def test_layout_is_reference_state_dict(lib):
    from oracle import transformer_ref as tr
    from slnlp import tf_engine as te
    for (E, H, N, F, Vs, Vt) in [(32, 4, 2, 64, 64, 16), (128, 4, 2, 256, 3000, 202), (512, 8, 6, 512, 3000, 202)]:
        cfg = te.make_config(E, H, N, F, Vs, Vt, 50, 48)
        ents, total = te.layout(cfg)
        want = tr.param_shapes(E, H, N, F, Vs, Vt)
        assert [(n, s) for n, s, _ in ents] == [(n, tuple(s)) for n, s in want]
        end = 0
        for n, s, off in ents:
            assert off % 4 == 0 and off >= end                 # 16-byte aligned, non-overlapping
            numel = 1
            for d in s:
                numel *= d
            end = off + numel
        assert total >= end and total % 4 == 0
    assert sum(1 for _ in ents) == 2 + 12 * 6 + 2 + 18 * 6 + 2 + 2
    # parameter counts quoted in SURVEY.md section 8 (a1)
    cfg = te.make_config(128, 4, 2, 256, 3000, 200, 50, 48)
    ents, _ = te.layout(cfg)
    count = 0
    for _, s, _ in ents:
        k = 1
        for d in s:
            k *= d
        count += k
    assert count == 1098440


def test_argument_validation_returns_codes_not_aborts(lib):
    from slnlp import _lib, tf_engine as te
    bad = te.make_config(130, 4, 2, 256, 3000, 202, 50, 48)       # head_dim 32.5
    assert lib.slnlp_tf_num_params(C.byref(bad)) == -1
    assert b"divisible" in lib.slnlp_last_error()
    bad = te.make_config(128, 4, 2, 256, 3000, 202, 5, 5001)      # beyond the reference's 5000-row positional table
    assert lib.slnlp_tf_workspace_bytes(C.byref(bad)) == -1
    assert b"seq_len" in lib.slnlp_last_error()
    assert lib.slnlp_tf_workspace_bytes(C.byref(te.make_config(128, 4, 2, 256, 3000, 202, 50, 65))) > 0      # S > 64 is fine
    bad = te.make_config(128, 4, 2, 256, 3000, 202, 1024, 65)     # more tokens per step than the embedding backward indexes
    assert lib.slnlp_tf_workspace_bytes(C.byref(bad)) == -1 and b"tokens" in lib.slnlp_last_error()
    a = _lib.GemmArgs()
    assert lib.slnlp_gemm(C.byref(a), None) == 1                  # null operands -> SLNLP_ERR_INVALID_ARG
    assert lib.slnlp_gemm(None, None) == 1


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from slnlp import ops, tf_engine as te
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4), M=4, N=4, K=4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        te.TransformerEngine(te.make_config(32, 4, 2, 64, 64, 16, 4, 12))


def test_bench_flop_contract():
    """bench.py prices a step with SURVEY.md section 8d's algorithmic-FLOP formula (forward per sequence; x3 for training)."""
    import bench
    f = {k: bench.fwd_flops_per_seq(bench.WORKLOADS[k]) for k in ("cfg1", "cfg2", "e1024", "cfg5")}
    assert abs(f["cfg1"] / 3.46e7 - 1) < 0.01 and abs(f["cfg2"] / 1.26e9 - 1) < 0.01
    assert abs(f["e1024"] / 4.37e9 - 1) < 0.01 and abs(f["cfg5"] / 5.83e9 - 1) < 0.01
