"""ctypes binding of libimpnn.so (include/impnn.h).  There is no fallback: if the HIP library is
missing or a tensor is not on the GPU the call raises - nothing here computes on the CPU."""
from __future__ import annotations

import ctypes as C
import threading
from pathlib import Path

import torch

import os

# IMPNN_LIB: another build of the library (diagnostics: tools/ab_bench.py style A/B runs); default: the in-tree one
_LIB_PATH = Path(os.environ.get("IMPNN_LIB") or Path(__file__).resolve().parent / "csrc" / "libimpnn.so")
_lock = threading.Lock()
_lib = None

i32, i64, f32, vp, sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t
PP = C.POINTER(vp)

# name -> (restype, argtypes); mirrors include/impnn.h one to one
SIGNATURES = {
    "impnn_abi_version": (C.c_int, []),
    "impnn_last_error_string": (C.c_char_p, []),
    "impnn_target_arch": (C.c_char_p, []),
    "impnn_embed_gather": (C.c_int, [vp, vp, vp, i64, i32, i32, vp]),
    "impnn_bmm_message": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "impnn_bond_type_matrices": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "impnn_bmm_message_typed": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "impnn_reduce_scatter_add": (C.c_int, [vp, vp, i32, vp, i32, i32, i32, i32, vp]),
    "impnn_bmm_fused": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "impnn_gated_update": (C.c_int, [vp] * 10 + [f32, vp, i64, i32, vp]),
    "impnn_gated_update_rows": (C.c_int, [vp] * 10 + [f32, vp, vp, vp, i64, i32, vp]),
    "impnn_kept_rows": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "impnn_row_index_fill": (C.c_int, [vp, vp, vp, vp, i32, i32, vp]),
    "impnn_global_sum_pool": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "impnn_encoder_step_floats": (i64, [i32, i32]),
    "impnn_encoder_plan_overflow_offset": (sz, []),
    "impnn_encoder_workspace_bytes": (C.c_int, [i32] * 10 + [C.POINTER(sz)]),
    "impnn_encoder_fused": (C.c_int, [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp, i32, vp, i32,
                                      C.POINTER(vp), i32, C.POINTER(vp), i32, i32, i32, i32, i32, i32, f32, i32, vp,
                                      sz, vp]),
    "impnn_encoder_prepared_bytes": (sz, [i32, i32, i32, i32]),
    "impnn_encoder_prepare_weights": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp, sz, vp]),
    "impnn_encoder_fused_prepared": (C.c_int, [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp, i32, vp, i32,
                                               C.POINTER(vp), i32, C.POINTER(vp), i32, i32, i32, i32, i32, i32, f32,
                                               i32, vp, sz, vp]),
    "impnn_encoder_plan": (C.c_int, [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, i32, i32, i32, i32, i32,
                                     i32, i32, i32, i32, vp, sz, vp, vp]),
    "impnn_encoder_run": (C.c_int, [i32, C.POINTER(vp), vp, i32, vp, i32, C.POINTER(vp), i32, C.POINTER(vp), i32, i32,
                                    i32, i32, i32, i32, f32, vp, vp, sz, vp]),
    "impnn_model_head_floats": (i64, [i32, i32, i32, i32]),
    "impnn_model_head": (C.c_int, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "impnn_model_head_tensors": (C.c_int, [i32, vp, vp, vp, PP, vp, i32, i32, i32, i32, vp]),
    "impnn_model_head_bwd": (C.c_int, [i32, vp, vp, vp, PP, vp, vp, vp, PP, i32, i32, i32, i32, vp]),
    "impnn_model_head_loss_workspace_floats": (i64, [i32]),
    "impnn_model_head_loss": (C.c_int, [i32, vp, vp, vp, PP, C.POINTER(f32), vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]),
    "impnn_model_head_loss_bwd": (C.c_int, [i32, vp, vp, vp, PP, C.POINTER(f32), vp, vp, vp, vp, PP, i32, i32, i32, i32,
                                            vp]),
    "impnn_embed_gather_bwd": (C.c_int, [vp, vp, vp, i64, i32, i32, vp]),
    "impnn_reduce_scatter_bwd": (C.c_int, [vp, vp, i32, vp, i32, i32, i32, i32, vp]),
    "impnn_global_sum_pool_bwd": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "impnn_bmm_message_typed_bwd_workspace_bytes": (i64, [i32, i32, i32]),
    "impnn_bmm_message_typed_sorted": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp]),
    "impnn_bmm_message_typed_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp]),
    "impnn_message_reduce_typed_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp]),
    "impnn_message_reduce_typed_bwd_scratch": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32, i32,
                                                         vp]),
    "impnn_bond_type_matrices_bwd": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "impnn_bond_type_matrices_multi": (C.c_int, [vp, PP, PP, i32, i32, i32, i32, vp]),
    "impnn_bond_type_matrices_multi_bwd": (C.c_int, [vp, PP, PP, PP, vp, i32, i32, i32, i32, i32, vp]),
    "impnn_bond_type_matrices_multi_bwd_workspace_floats": (i64, [i32, i32, i32, i32]),
    "impnn_bond_type_matrices_multi_bwd_ws": (C.c_int, [vp, PP, PP, PP, vp, i32, i32, i32, i32, i32, vp, i64, vp]),
    "impnn_gated_update_param_floats": (i64, [i32]),
    "impnn_gated_update_bwd_workspace_floats": (i64, [i64, i32]),
    "impnn_gated_update_bwd": (C.c_int, [vp] * 9 + [f32] + [vp] * 5 + [i64, i64, i32, i32, vp]),
    "impnn_gated_update_rows_bwd_workspace_floats": (i64, [i64, i32]),
    "impnn_gated_update_rows_bwd": (C.c_int, [vp] * 9 + [f32] + [vp] * 5 + [i64, vp, vp, i64, i32, i32, vp]),
    "impnn_gated_update_rows_saved_floats": (i64, [i64, i32]),
    "impnn_gated_update_rows_train": (C.c_int, [vp] * 10 + [f32, vp, vp, vp, i64, i32, vp, vp]),
    "impnn_gated_update_rows_bwd_saved": (C.c_int, [vp] * 9 + [f32] + [vp] * 5 + [i64, vp, vp, i64, i32, i32, vp, vp]),
    "impnn_adam_clipnorm_step": (C.c_int, [vp, vp, i32, i64, f32, f32, f32, f32, f32, vp]),
    "impnn_adam_clipnorm_step_counted": (C.c_int, [vp, vp, i32, vp, f32, f32, f32, f32, f32, vp]),
    "impnn_batch_assemble": (C.c_int, [i32, vp, i32, i32, PP, PP, PP, PP, PP, i32, i32, i32, PP, PP, PP, vp, vp, vp]),
    "impnn_gather_rows": (C.c_int, [i32, PP, PP, C.POINTER(i64), vp, i32, vp]),
    "impnn_validate_indices": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "impnn_profile_enable": (C.c_int, [i32]),
    "impnn_profile_collect": (C.c_int, [C.POINTER(C.c_float), i32, C.POINTER(i32)]),
    "impnn_profile_disable": (C.c_int, []),
    "impnn_debug_set_stamp_buffer": (C.c_int, [vp, sz]),
}

ABI_VERSION = 3
IMPNN_E_UNSUPPORTED = -2


class PlanInfo(C.Structure):
    """impnn_encoder_plan_info: what impnn_encoder_plan planned for (host-side plain data)."""
    _fields_ = [("v", i32 * 12)]


class ImpnnError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"libimpnn error {code}: {text}")
        self.code = code


def lib_path():
    return _LIB_PATH


def load():
    """Loads libimpnn.so and checks every declared symbol.  Raises if the library is absent."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not _LIB_PATH.exists():
            raise RuntimeError(
                f"{_LIB_PATH} is missing: build it with `python -m ionic_mpnn_amd.build` "
                "(ionic_mpnn_amd has no CPU fallback)")
        lib = C.CDLL(str(_LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if lib.impnn_abi_version() != ABI_VERSION:
            raise RuntimeError("libimpnn.so ABI version mismatch; rebuild it")
        _lib = lib
        return lib


def check(code):
    if code != 0:
        raise ImpnnError(code, load().impnn_last_error_string().decode())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"expected torch.Tensor, got {type(t).__name__}")
        if not t.is_cuda:
            raise RuntimeError("ionic_mpnn_amd runs on the MI355X HIP path only; got a CPU tensor "
                               "(there is no CPU fallback)")


def f32c(t):
    """contiguous float32 view/copy on the same device"""
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t.contiguous()


def i32c(t):
    if t.dtype != torch.int32:
        if t.dtype in (torch.int64, torch.int16, torch.int8, torch.uint8):
            t = t.to(torch.int32)
        else:
            raise TypeError(f"index tensor must be an integer tensor, got {t.dtype}")
    return t.contiguous()


def ptr(t):
    return C.c_void_p(t.data_ptr() if t is not None and t.numel() > 0 else (t.data_ptr() if t is not None else 0))
