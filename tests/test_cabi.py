"""CPU: libimpnn.so loads without a GPU, exports every symbol include/impnn.h declares, and
rejects bad arguments before touching the device (no compute calls here)."""
import ctypes as C
import re
from pathlib import Path

import pytest

from conftest import ROOT
from ionic_mpnn_amd import _lib

HEADER = (ROOT / "include" / "impnn.h").read_text()


def declared_symbols():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(impnn_[a-z0-9_]+)\s*\(", body)))


def test_library_is_built_in_tree():
    assert _lib.lib_path().exists(), "run python -m ionic_mpnn_amd.build (or __graft_entry__.build())"
    assert _lib.lib_path().parent == ROOT / "ionic_mpnn_amd" / "csrc"


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 15
    raw = C.CDLL(str(_lib.lib_path()))
    for n in names:
        assert hasattr(raw, n), f"{n} declared in impnn.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes binding table out of sync with impnn.h"


def test_identity():
    lib = _lib.load()
    assert lib.impnn_abi_version() == 1
    assert lib.impnn_target_arch() == b"gfx950"
    assert lib.impnn_encoder_step_floats(32, 8) == 8 * 1024 + 3 * (2048 + 32) + 64


def test_bad_arguments_are_status_codes_not_crashes():
    lib = _lib.load()
    null = None
    assert lib.impnn_embed_gather(null, null, null, 4, 10, 32, null) == -1
    assert b"null" in lib.impnn_last_error_string()
    assert lib.impnn_embed_gather(null, null, null, -1, 10, 32, null) == -1
    assert lib.impnn_gated_update(*([null] * 10), 1e-3, null, 5, 32, null) == -1
    assert lib.impnn_reduce_scatter_add(null, null, 0, null, 1, 4, 4, 8, null) == -1
    assert b"tgt_stride" in lib.impnn_last_error_string()
    assert lib.impnn_global_sum_pool(null, null, null, 2, 0, 8, null) == -1
    # zero-size work is a no-op success
    assert lib.impnn_embed_gather(null, null, null, 0, 10, 32, null) == 0
    assert lib.impnn_bmm_message(null, null, null, null, null, 0, 4, 4, 8, 2, null) == 0


def test_encoder_shape_coverage_is_reported():
    lib = _lib.load()
    need = C.c_size_t(0)
    assert lib.impnn_encoder_workspace_bytes(2, 4096, 40, 80, 32, 8, 3, 72, C.byref(need)) == 0
    assert 0 < need.value < (64 << 20)
    assert lib.impnn_encoder_workspace_bytes(2, 4096, 40, 80, 128, 8, 3, 72, C.byref(need)) == -2   # D=128
    assert lib.impnn_encoder_workspace_bytes(2, 4096, 40, 80, 32, 1024, 3, 72, C.byref(need)) == -2  # K=D*D
    assert b"not covered" in lib.impnn_last_error_string()
    assert lib.impnn_encoder_workspace_bytes(3, 16, 40, 80, 32, 8, 3, 72, C.byref(need)) == -1


def test_cpu_tensors_fail_loudly():
    import torch
    from ionic_mpnn_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.embed_gather(torch.zeros(2, 3, dtype=torch.int32), torch.zeros(5, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.global_sum_pool(torch.zeros(1, 2, 4), torch.ones(1, 2, dtype=torch.int32))
