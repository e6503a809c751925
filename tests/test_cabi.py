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
    assert lib.impnn_abi_version() == 3
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
    F32, F16X2, TYPED = 0, 1, 2
    for mode in (F32, F16X2, TYPED):
        assert lib.impnn_encoder_workspace_bytes(2, 4096, 40, 80, 32, 8, 3, 72, mode, 0, C.byref(need)) == 0
        assert 0 < need.value < (64 << 20)
        # wide states (train_viscosity.py with atom_dim=128): the per-bond-type mode only (csrc/encoder_wide.hip)
        rc = lib.impnn_encoder_workspace_bytes(2, 4096, 40, 80, 128, 8, 6, 72, mode, 0, C.byref(need))
        assert rc == (0 if mode == TYPED else -2)
    assert (256 << 20) < need.value < (2 << 30)
    assert lib.impnn_encoder_workspace_bytes(2, 64, 40, 80, 64, 8, 3, 72, TYPED, 0, C.byref(need)) == 0        # D=64
    assert lib.impnn_encoder_workspace_bytes(2, 64, 40, 80, 48, 8, 3, 72, TYPED, 0, C.byref(need)) == -2       # D=48
    assert lib.impnn_encoder_workspace_bytes(2, 64, 40, 1100, 128, 8, 3, 72, TYPED, 0, C.byref(need)) == -2    # E > 1024
    assert lib.impnn_encoder_workspace_bytes(2, 64, 160, 640, 128, 8, 3, 72, TYPED, 0, C.byref(need)) == 0 and need.value > 0
    assert lib.impnn_encoder_workspace_bytes(2, 64, 40, 80, 128, 8, 3, 72, 3, 0, C.byref(need)) == 0           # f32x3 (round 3)
    # K = D*D (train_melting_point.py:146): the typed mode covers it (BASELINE config 3), the pull form does not
    assert lib.impnn_encoder_workspace_bytes(2, 8192, 40, 80, 32, 1024, 4, 72, TYPED, 0, C.byref(need)) == 0
    assert lib.impnn_encoder_workspace_bytes(2, 4096, 40, 80, 32, 1024, 3, 72, F32, 0, C.byref(need)) == -2
    assert b"not covered" in lib.impnn_last_error_string()
    # the typed records take any padded shape (explicit-H molecules: N = 160, E = 640, train_viscosity.py:288-289): a
    # chunk is bounded by what a molecule HOLDS, checked per batch by the plan kernels (round 3); K = D^2 included
    assert lib.impnn_encoder_workspace_bytes(2, 16, 40, 300, 32, 8, 3, 72, TYPED, 0, C.byref(need)) == 0
    assert lib.impnn_encoder_workspace_bytes(2, 4096, 160, 640, 32, 8, 3, 72, TYPED, 0, C.byref(need)) == 0 and need.value > 0
    assert lib.impnn_encoder_workspace_bytes(2, 4096, 160, 640, 32, 1024, 4, 72, TYPED, 0, C.byref(need)) == 0 and need.value > 0
    assert lib.impnn_encoder_workspace_bytes(2, 4096, 160, 640, 32, 8, 3, 72, F32, 0, C.byref(need)) == -2   # pull form: N <= 128
    assert lib.impnn_encoder_workspace_bytes(3, 16, 40, 80, 32, 8, 3, 72, F32, 0, C.byref(need)) == -1
    assert lib.impnn_encoder_workspace_bytes(2, 16, 40, 80, 32, 8, 3, 72, 3, 0, C.byref(need)) == 0        # f32x3
    assert lib.impnn_encoder_workspace_bytes(2, 16, 40, 80, 32, 8, 3, 72, 4, 0, C.byref(need)) == -1       # no mode 4
    assert lib.impnn_encoder_workspace_bytes(2, 16, 40, 80, 32, 8, 3, 72, F32, -1, C.byref(need)) == -1
    assert lib.impnn_encoder_prepared_bytes(32, 3, 72, 3) > lib.impnn_encoder_prepared_bytes(32, 3, 72, TYPED) > lib.impnn_encoder_prepared_bytes(32, 3, 72, F32) > 0
    assert lib.impnn_encoder_prepared_bytes(128, 6, 72, TYPED) == 6 * (72 * 128 * 128 + 6 * 128 * 128 + 5 * 128) * 4
    # mode 3, wide: f32 type matrices | gate kernels as three bf16 planes | vectors | the type matrices as three bf16 planes
    assert lib.impnn_encoder_prepared_bytes(128, 6, 72, 3) == 6 * (72 * 128 * 128 + 9 * 128 * 128 + 5 * 128 + 72 * 128 * 128 * 3 // 2) * 4
    assert lib.impnn_encoder_prepared_bytes(128, 6, 72, F32) == 0 and lib.impnn_encoder_prepared_bytes(48, 6, 72, TYPED) == 0


def test_encoder_sizing_has_no_hidden_state():
    """The workgroup count is an argument, not library state: two host threads sizing workspaces with different
    counts at the same time always get the answer that belongs to their own arguments."""
    import threading
    lib = _lib.load()

    def size(wgs, mode=2):
        need = C.c_size_t(0)
        assert lib.impnn_encoder_workspace_bytes(2, 4096, 40, 80, 32, 8, 3, 72, mode, wgs, C.byref(need)) == 0
        return need.value

    want = {w: size(w) for w in (0, 64, 128, 200)}
    assert len(set(want.values())) >= 3 and want[64] != want[128]
    errors = []

    def worker(wgs):
        for _ in range(3000):
            got = size(wgs)
            if got != want[wgs]:
                errors.append((wgs, got))
                return

    threads = [threading.Thread(target=worker, args=(w,)) for w in (64, 128, 200, 0)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    # neither entry exists any more: the mode and the workgroup count travel with every call
    raw = C.CDLL(str(_lib.lib_path()))
    assert not hasattr(raw, "impnn_encoder_set_workgroups") and not hasattr(raw, "impnn_encoder_set_mode")


def test_plan_info_is_checked_on_the_host():
    """impnn_encoder_run validates the plan info before anything is enqueued (no device needed: every failing
    call returns before its first launch; the pointers below are never dereferenced)."""
    lib = _lib.load()
    info = _lib.PlanInfo()
    fake = (C.c_void_p * 2)(0x1000, 0x1000)
    fp = C.c_void_p(0x1000)

    def run(B=4, N=10, E=20, S=1, Vb=5, mode=2):
        return lib.impnn_encoder_run(2, fake, fp, 10, fp, Vb, fake, mode, fake, B, N, E, 32, 8, S, 1e-3, C.byref(info),
                                     fp, 1 << 30, None)

    assert run() == -1 and b"not filled by impnn_encoder_plan" in lib.impnn_last_error_string()
    # B = 0 plans nothing on the device but still records what it was planned for
    rc = lib.impnn_encoder_plan(2, fake, fake, fake, 0, 10, 20, 32, 8, 1, 10, 5, 2, 96, fp, 1 << 30, None, C.byref(info))
    assert rc == 0 and info.v[3] == 0 and info.v[8] == 96 and info.v[1] == 1
    for kw in ({"B": 4}, {"N": 11}, {"E": 21}, {"S": 2}, {"Vb": 6}, {"mode": 0}):
        args = {"B": 0}
        args.update(kw)
        assert run(**args) == -1, kw
        assert b"planned for another" in lib.impnn_last_error_string(), kw
    assert run(B=0) == 0  # matching (empty) batch: accepted


def test_cpu_tensors_fail_loudly():
    import torch
    from ionic_mpnn_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.embed_gather(torch.zeros(2, 3, dtype=torch.int32), torch.zeros(5, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.global_sum_pool(torch.zeros(1, 2, 4), torch.ones(1, 2, dtype=torch.int32))


def test_row_list_backward_reports_its_coverage():
    """impnn_gated_update_rows_bwd covers atom_dim 64 / 128 (the wide matrix-core kernel); other widths are refused
    before anything is launched, and the workspace query answers 0 for them."""
    lib = _lib.load()
    null = None
    assert lib.impnn_gated_update_rows_bwd_workspace_floats(1000, 128) > lib.impnn_gated_update_bwd_workspace_floats(1000, 128) > 0
    assert lib.impnn_gated_update_rows_bwd_workspace_floats(1000, 32) == 0
    rc = lib.impnn_gated_update_rows_bwd(*([null] * 9), 1e-3, *([null] * 5), 0, null, null, 1000, 32, 0, null)
    assert rc == -2 and b"row-list" in lib.impnn_last_error_string()
    rc = lib.impnn_gated_update_rows_bwd(*([null] * 9), 1e-3, *([null] * 5), 0, null, null, 1000, 128, 0, null)
    assert rc == -1  # null pointers
