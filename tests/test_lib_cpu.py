"""CPU: the C-ABI library builds, loads and exports every symbol include/stgcnn_hip.h declares;
argument validation that needs no GPU; host-side layout arithmetic."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from social_stgcnn_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_header_symbols_are_exported(L):
    from social_stgcnn_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "stgcnn_hip.h")).read()
    declared = set(re.findall(r"\b(stg_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert L.stg_abi_version() == _lib.ABI_VERSION


def test_layout_queries_match_reference_counts(L):
    from social_stgcnn_amd import ops
    d = ops.make_desc(1, 5, 2, 5, 8, 12, 3, 2, False, True)
    assert L.stg_model_param_count(ctypes.byref(d)) == 7563          # 7,563 parameters
    assert L.stg_model_buffer_count(ctypes.byref(d)) == 30           # 3 BatchNorm x (mean, var) x 5
    assert L.stg_model_stat_floats(ctypes.byref(d)) == 30
    assert L.stg_model_ws_floats(ctypes.byref(d), 32) == 64 + 32 * (16 + 8 + 40 + 40 + 40 + 4 * 60) + 5 * 12 * 240
    # batch tail of the workspace: the prepared bf16 A operands of the five input-gradient convs (24 x 64 lanes x 16 bytes
    # each) when the exact-bf16 backward serves the batch (V <= 128: one wave per scene up to 32 pedestrians, teams of two
    # or four waves beyond), nothing otherwise; behind them the scene order of a ragged batch (N indices, V + 2 tier
    # offsets, N sorted counts, rounded up to 4)
    order = lambda n, v: (2 * n + v + 2 + 3) & ~3
    assert L.stg_model_ws_tail_floats(ctypes.byref(d), 100, 32) == 5 * 24 * 64 * 4 + order(100, 32)
    assert L.stg_model_ws_tail_floats(ctypes.byref(d), 7, 64) == 5 * 24 * 64 * 4 + order(7, 64)
    assert L.stg_model_ws_tail_floats(ctypes.byref(d), 7, 128) == 5 * 24 * 64 * 4 + order(7, 128)
    assert L.stg_model_ws_tail_floats(ctypes.byref(d), 7, 129) == order(7, 129)
    d_f32 = ops.make_desc(1, 5, 2, 5, 8, 12, 3, 2, False, True)
    from social_stgcnn_amd import _lib as lib_mod
    d_f32.flags |= lib_mod.OPT_F32_MFMA
    assert L.stg_model_ws_tail_floats(ctypes.byref(d_f32), 0, 32) == order(0, 32)
    d2 = ops.make_desc(2, 3, 2, 5, 8, 12, 3, 2, False, False)
    # second block: identity residual -> no residual conv / BN parameters
    blk0 = 10 + 5 + 10 + 1 + 75 + 5 + 10 + 10 + 5 + 10 + 1
    blk1 = 25 + 5 + 10 + 1 + 75 + 5 + 10 + 1
    txp = (12 * 8 * 9 + 12) + 2 * (12 * 12 * 9 + 12) + (12 * 12 * 9 + 12) + 3
    assert L.stg_model_param_count(ctypes.byref(d2)) == blk0 + blk1 + txp
    d3 = ops.make_desc(1, 0, 5, 5, 8, 0, 3, 1, False, False)          # stand-alone st_gcn, identity residual
    assert L.stg_model_param_count(ctypes.byref(d3)) == blk1


def test_invalid_arguments_return_status_not_abort(L):
    from social_stgcnn_amd import ops
    assert L.stg_adj_build(None, 0, 0, 0, 0, None, 1, 4, 8, 1, None, None, None) == -1
    assert b"null" in L.stg_last_error()
    bad = ops.make_desc(1, 5, 2, 5, 6, 12, 3, 2, False, False)       # seq_len the kernels are not built for
    assert L.stg_model_param_count(ctypes.byref(bad)) == -2
    assert b"seq_len" in L.stg_last_error()
    assert L.stg_selftest_mfma(None, None, 3, None, None) == -1
    d = ops.make_desc(1, 5, 2, 5, 8, 12, 3, 2, False, True)
    assert L.stg_model_bwd_scratch_floats(ctypes.byref(d), 2048, 32) > 0
    assert L.stg_model_bwd_scratch_floats(ctypes.byref(d), 16, 400) == -3     # does not fit 160 KiB of LDS


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "social_stgcnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU or eager", ""), os.path.join(dirpath, f)


def test_module_surface_matches_reference_state_dict():
    import numpy as np
    import torch
    from social_stgcnn_amd.model import ConvTemporalGraphical, social_stgcnn, st_gcn   # noqa: F401
    g = np.load(os.path.join(ROOT, "tests", "golden", "init_seed0.npz"))
    torch.manual_seed(0)
    m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12)
    sd = m.state_dict()
    assert list(sd.keys()) == list(g.files)
    for k in g.files:                 # same constructor RNG order -> identical default initialisation
        assert np.array_equal(sd[k].numpy(), g[k]), k
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights_eth.npz"))
    m.load_state_dict({k: torch.from_numpy(np.array(w[k])) for k in w.files})
    with pytest.raises(RuntimeError):          # a CPU tensor must fail loudly, not fall back
        m.eval()(torch.zeros(1, 2, 8, 3), torch.zeros(8, 3, 3))


def test_weight_gradient_kernel_keeps_its_loads_in_flight_and_tracked(tmp_path):
    """Compile-only ISA check of txp_wgrad_bf16: its staging loads are issued one item ahead.  They
    must be loads the COMPILER tracks (no global_load in inline assembly: a destination register with data in flight must
    never be visible to the register allocator as if it were ready), they must survive the per-scene barrier (no
    s_waitcnt vmcnt(0) between a staging load and the loop's s_barrier), and the kernel must not address memory through
    flat_*."""
    import subprocess
    src = os.path.join(ROOT, "social_stgcnn_amd", "csrc", "txp_wgrad_bf16.hip")
    text = open(src).read()
    assert not re.search(r'asm[^;]*global_load', text), "staging loads must not be inline assembly"
    out = str(tmp_path / "wgrad.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--offload-device-only", "-S",
                           src, "-o", out])
    lines = open(out).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN3stg.*txp_wgrad_bf16_kernel.*:", l)]
    assert len(starts) == 4
    for st in starts:
        end = next(i for i in range(st, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
        body = [l.strip() for l in lines[st:end] if l.strip() and not l.strip().startswith((";", "."))]
        assert not any(l.startswith("flat_") for l in body)
        assert not any(l.startswith("scratch_") for l in body), "spilled registers in " + lines[st]
        loads = [i for i, l in enumerate(body) if l.startswith("global_load")]
        bars = [i for i, l in enumerate(body) if l.startswith("s_barrier")]
        assert loads and bars
        # a barrier with a staging load a few instructions in front of it and no full vmcnt drain in between: the load is
        # still in flight when the workgroup meets
        in_flight = [b for b in bars
                     if any(0 < b - ld <= 40 and not any("vmcnt(0)" in body[k] for k in range(ld, b)) for ld in loads)]
        assert in_flight, lines[st]
