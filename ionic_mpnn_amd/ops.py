"""Functional wrappers: torch CUDA tensors in, torch CUDA tensors out, every one a libimpnn call on
the tensor's device and torch's current stream.  Shapes/dtypes follow the reference tensors
(int32 ids and connectivity, float32 state)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check, f32c, i32c, ptr, require_gpu, stream_ptr

LN_EPS = 1e-3  # keras LayerNormalization default (models/layers.py:139)
DEBUG_VALIDATE = False  # True: raise like TF-CPU on out-of-range indices (costs a device sync)


def _wants_grad(*tensors):
    """Training: an input requires grad and grad mode is on -> go through ionic_mpnn_amd.autograd."""
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


class NoBackward(NotImplementedError):
    """Raised when a forward-only entry (dense bond_state message, fused encoder) is asked for gradients."""


def embed_gather(ids, table):
    """Embedding(mask_zero=False) lookup, train_viscosity.py:171-172."""
    if _wants_grad(table):
        from . import autograd
        return autograd.EmbedGather.apply(ids, table)
    require_gpu(ids, table)
    ids = i32c(ids)
    table = f32c(table)
    out = torch.empty(*ids.shape, table.shape[1], dtype=torch.float32, device=table.device)
    with torch.cuda.device(table.device):
        check(_lib.load().impnn_embed_gather(ptr(ids), ptr(table), ptr(out), ids.numel(), table.shape[0],
                                             table.shape[1], stream_ptr()))
    return out


def _check_bmm_shapes(h, conn):
    if h.dim() != 3:
        raise ValueError(f"atom_state must be (B,N,D), got {tuple(h.shape)}")
    if conn.dim() != 3 or conn.shape[-1] != 2 or conn.shape[0] != h.shape[0]:
        raise ValueError(f"connectivity must be (B,E,2) with the batch of atom_state, got {tuple(conn.shape)}")


def validate_indices(conn=None, atom_ids=None, bond_ids=None, N=None, Va=1 << 30, Vb=1 << 30):
    """Raises ValueError where tf-CPU's gather/scatter_nd would (models/layers.py:106,78-82)."""
    ref = conn if conn is not None else (atom_ids if atom_ids is not None else bond_ids)
    require_gpu(ref)
    B = ref.shape[0]
    E = conn.shape[1] if conn is not None else (bond_ids.shape[1] if bond_ids is not None else 0)
    Nn = N if N is not None else (atom_ids.shape[1] if atom_ids is not None else 1)
    counts = torch.zeros(3, dtype=torch.int32, device=ref.device)
    with torch.cuda.device(ref.device):
        check(_lib.load().impnn_validate_indices(
            ptr(i32c(conn)) if conn is not None else None,
            ptr(i32c(atom_ids)) if atom_ids is not None else None,
            ptr(i32c(bond_ids)) if bond_ids is not None else None,
            ptr(counts), B, Nn, E, Va, Vb, stream_ptr()))
    c = counts.tolist()
    if any(c):
        raise ValueError(f"out-of-range indices: connectivity={c[0]} atom_ids={c[1]} bond_ids={c[2]}")


def batch_assemble(sample_idx, ions, max_atoms, slots, id_shift=1, t_flat=None):
    """train_viscosity.py:291-314 build_inputs(idx) on the GPU (impnn_batch_assemble).

    ions: per ion a dict of resident int32 device tensors atom_flat, atom_off (M+1), edge_flat (n,2),
    bond_flat, edge_off (M+1).  Returns per ion (atom_ids (B,N), bond_ids (B,slots), conn (B,slots,2))
    and the gathered t (B,1) or None."""
    require_gpu(sample_idx, *[t for ion in ions for t in ion.values()])
    sample_idx = i32c(sample_idx)
    dev = sample_idx.device
    B, n = int(sample_idx.numel()), len(ions)
    M = int(ions[0]["atom_off"].numel()) - 1
    outs = [(torch.empty(B, max_atoms, dtype=torch.int32, device=dev),
             torch.empty(B, slots, dtype=torch.int32, device=dev),
             torch.empty(B, slots, 2, dtype=torch.int32, device=dev)) for _ in range(n)]
    t_out = torch.empty(B, 1, dtype=torch.float32, device=dev) if t_flat is not None else None
    arr = C.c_void_p * n
    mk = lambda ts: arr(*[t.data_ptr() for t in ts])
    for ion in ions:
        for k in ("atom_flat", "atom_off", "edge_flat", "bond_flat", "edge_off"):
            if ion[k].dtype != torch.int32 or not ion[k].is_contiguous():
                raise ValueError(f"{k} must be a contiguous int32 tensor")
        if ion["atom_off"].numel() != M + 1 or ion["edge_off"].numel() != M + 1:
            raise ValueError("offset tables must have M+1 entries")
    with torch.cuda.device(dev):
        check(_lib.load().impnn_batch_assemble(
            n, ptr(sample_idx), B, M, mk([i["atom_flat"] for i in ions]), mk([i["atom_off"] for i in ions]),
            mk([i["edge_flat"] for i in ions]), mk([i["bond_flat"] for i in ions]), mk([i["edge_off"] for i in ions]),
            int(id_shift), int(max_atoms), int(slots), mk([o[0] for o in outs]), mk([o[1] for o in outs]),
            mk([o[2] for o in outs]), ptr(f32c(t_flat)) if t_flat is not None else None,
            ptr(t_out) if t_out is not None else None, stream_ptr()))
    return outs, t_out


def bmm_message(h, bond_state, conn, W):
    """BondMatrixMessage.call, models/layers.py:100-117 -> messages (B,E,D)."""
    if _wants_grad(h, bond_state, W):
        raise NoBackward("gradients of BondMatrixMessage are implemented for bond states that come from an "
                         "Embedding(lazy=True) (per-bond-type schedule), as the reference wires it")
    require_gpu(h, bond_state, conn, W)
    _check_bmm_shapes(h, conn)
    h, bond_state, W, conn = f32c(h), f32c(bond_state), f32c(W), i32c(conn)
    B, N, D = h.shape
    E, K = conn.shape[1], W.shape[0]
    if tuple(bond_state.shape) != (B, E, K) or tuple(W.shape) != (K, D, D):
        raise ValueError(f"bond_state {tuple(bond_state.shape)} / bond_transform {tuple(W.shape)} do not match "
                         f"(B,E,K)=({B},{E},{K}), (K,D,D)=({K},{D},{D})")
    if DEBUG_VALIDATE:
        validate_indices(conn=conn, N=N)
    m = torch.empty(B, E, D, dtype=torch.float32, device=h.device)
    with torch.cuda.device(h.device):
        check(_lib.load().impnn_bmm_message(ptr(h), ptr(bond_state), ptr(conn), ptr(W), ptr(m), B, N, E, D, K,
                                            stream_ptr()))
    return m


def bond_type_matrices(bond_table, W):
    """A[v] = sum_k bond_table[v,k] W[k]  (models/layers.py:108 once per vocabulary entry)."""
    if _wants_grad(bond_table, W):
        from . import autograd
        return autograd.BondTypeMatrices.apply(bond_table, W)
    require_gpu(bond_table, W)
    bond_table, W = f32c(bond_table), f32c(W)
    Vb, K = bond_table.shape
    D = W.shape[-1]
    if W.numel() != K * D * D:
        raise ValueError("bond_transform does not match bond_table")
    out = torch.empty(Vb, D, D, dtype=torch.float32, device=W.device)
    with torch.cuda.device(W.device):
        check(_lib.load().impnn_bond_type_matrices(ptr(bond_table), ptr(W), ptr(out), Vb, K, D, stream_ptr()))
    return out


def message_scratch(conn, bond_ids, B, E, D):
    """(buffer, reuse) - the (B,E,D) message buffer of this (conn, bond_ids) inside an ``autograd.training_pass``: the
    layers of an ion write their messages into the same buffer one after the other (each is consumed by the Reduce right
    behind it), so the zero rows of masked edges are written by the first layer only.  Outside a pass: (None, False)."""
    from . import autograd
    pass_id = autograd.current_pass()
    if pass_id is None:
        return None, False
    key = (pass_id, conn._version, bond_ids.data_ptr(), bond_ids._version, B, E, D)
    cached = getattr(conn, "_impnn_msg_scratch", None)
    if cached is not None and cached[0] == key:
        return cached[1], True
    m = torch.empty(B, E, D, dtype=torch.float32, device=conn.device)
    conn._impnn_msg_scratch = (key, m)
    return m, False


def bmm_message_typed(h, bond_ids, conn, type_mats, out=None, out_reused=False):
    """``out`` / ``out_reused``: ops.message_scratch's pair (model-internal: the training loop's message buffer)."""
    if _wants_grad(h, type_mats):
        from . import autograd
        return autograd.BmmMessageTyped.apply(h, bond_ids, conn, type_mats)
    require_gpu(h, bond_ids, conn, type_mats)
    _check_bmm_shapes(h, conn)
    h, type_mats, conn, bond_ids = f32c(h), f32c(type_mats), i32c(conn), i32c(bond_ids)
    B, N, D = h.shape
    E, Vb = conn.shape[1], type_mats.shape[0]
    if DEBUG_VALIDATE:
        validate_indices(conn=conn, bond_ids=bond_ids, N=N, Vb=Vb)
    m = out if out is not None else torch.empty(B, E, D, dtype=torch.float32, device=h.device)
    lib = _lib.load()
    with torch.cuda.device(h.device):
        if D != 32 and D <= 128 and E > 0 and B > 0:
            # any other width: type-sorted segments with A[type] in LDS (the D = 32 kernel sorts inside its workgroups)
            ws, ready = edge_sort_workspace(conn, bond_ids, B, E, Vb)
            flags = (1 if ready else 0) | (2 if out is not None and out_reused else 0)
            check(lib.impnn_bmm_message_typed_sorted(ptr(h), ptr(bond_ids), ptr(conn), ptr(type_mats), ptr(m), ptr(ws),
                                                     ws.numel(), B, N, E, D, Vb, flags, stream_ptr()))
        else:
            check(lib.impnn_bmm_message_typed(ptr(h), ptr(bond_ids), ptr(conn), ptr(type_mats), ptr(m), B, N,
                                              E, D, Vb, stream_ptr()))
    return m


def edge_sort_workspace(conn, bond_ids, B, E, Vb):
    """Workspace of the by-bond-type edge sort for (conn, bond_ids) and whether it already holds that sort.
    Inside an ``autograd.training_pass`` the workspace is kept on the connectivity tensor object, so the forward and
    backward message kernels of all layers of one ion sort once; outside, every call gets a fresh workspace."""
    from . import autograd
    lib = _lib.load()
    wsb = int(lib.impnn_bmm_message_typed_bwd_workspace_bytes(B, E, Vb))
    pass_id = autograd.current_pass()
    if pass_id is None:
        return torch.empty(max(wsb, 4), dtype=torch.uint8, device=conn.device), False
    key = (pass_id, conn._version, bond_ids.data_ptr(), bond_ids._version, Vb, wsb)
    cached = getattr(conn, "_impnn_edge_sort", None)
    if cached is not None and cached[0] == key:
        return cached[1], True
    ws = torch.empty(max(wsb, 4), dtype=torch.uint8, device=conn.device)
    conn._impnn_edge_sort = (key, ws)
    return ws, False


def reduce_scatter_add(messages, tgt_idx, num_atoms):
    """Reduce.call, models/layers.py:57-83.  `tgt_idx` may be the strided view conn[:, :, 1]
    (train_viscosity.py:182); it is then read in place."""
    if _wants_grad(messages):
        from . import autograd
        return autograd.ReduceScatterAdd.apply(messages, tgt_idx, num_atoms)
    require_gpu(messages, tgt_idx)
    messages = f32c(messages)
    B, E, D = messages.shape
    if tgt_idx.dtype != torch.int32:
        tgt_idx = i32c(tgt_idx)
    if tuple(tgt_idx.shape) != (B, E):
        raise ValueError(f"tgt_idx must be (B,E)=({B},{E}), got {tuple(tgt_idx.shape)}")
    stride = 1
    if not tgt_idx.is_contiguous():
        if B * E > 0 and tgt_idx.stride(1) == 2 and (tgt_idx.stride(0) == 2 * E or B == 1):
            stride = 2  # a conn[:, :, 1] view
        else:
            tgt_idx = tgt_idx.contiguous()
    if DEBUG_VALIDATE:
        validate_indices(conn=tgt_idx.contiguous().unsqueeze(-1).expand(B, E, 2), N=num_atoms)
    agg = torch.empty(B, num_atoms, D, dtype=torch.float32, device=messages.device)
    with torch.cuda.device(messages.device):
        check(_lib.load().impnn_reduce_scatter_add(ptr(messages), ptr(tgt_idx), stride, ptr(agg), B, num_atoms,
                                                   E, D, stream_ptr()))
    return agg


def bmm_fused(h, bond_state, conn, W):
    """Orphan models/bond_matrix_message.py:37-65 signature: -> aggregated (B,N,D)."""
    if _wants_grad(h, bond_state, W):
        raise NoBackward("the fused message+reduce entry is forward-only; train with fused=False")
    require_gpu(h, bond_state, conn, W)
    _check_bmm_shapes(h, conn)
    h, bond_state, W, conn = f32c(h), f32c(bond_state), f32c(W), i32c(conn)
    B, N, D = h.shape
    E, K = conn.shape[1], bond_state.shape[-1]
    if W.numel() != K * D * D:
        raise ValueError("bond_transform must hold K*D*D values")
    agg = torch.empty(B, N, D, dtype=torch.float32, device=h.device)
    if E == 0:
        return agg.zero_()
    with torch.cuda.device(h.device):
        check(_lib.load().impnn_bmm_fused(ptr(h), ptr(bond_state), ptr(conn), ptr(W), ptr(agg), B, N, E, D, K,
                                          stream_ptr()))
    return agg


def kept_row_index(atom_ids, bond_ids, conn, Vb):
    """(row_index (B*N,) int32, n_rows (1,) int32) of the rows an encode() loop has to carry: molecule b's rows
    [0, r_b) as flat indices b*N + n (impnn_kept_rows + impnn_row_index_fill; the prefix sum in between is
    torch.cumsum).  Everything stays on the device."""
    require_gpu(atom_ids, bond_ids, conn)
    atom_ids, bond_ids, conn = i32c(atom_ids), i32c(bond_ids), i32c(conn)
    B, N = atom_ids.shape
    E = conn.shape[1]
    dev = atom_ids.device
    r = torch.empty(B, dtype=torch.int32, device=dev)
    idx = torch.empty(max(B * N, 1), dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        check(lib.impnn_kept_rows(ptr(atom_ids), ptr(bond_ids), ptr(conn), ptr(r), B, N, E, int(Vb), stream_ptr()))
        incl = torch.cumsum(r, 0, dtype=torch.int32)
        check(lib.impnn_row_index_fill(ptr(r), ptr(incl), ptr(idx), ptr(cnt), B, N, stream_ptr()))
    return idx, cnt


def gated_update(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, beta, eps=LN_EPS, rows=None, save=False):
    """GatedUpdate.call, models/layers.py:142-156.  ``rows`` = (row_index, n_rows) of kept_row_index: only those rows
    of the output are computed (model-internal use: padding atoms; the rest of ``out`` is undefined).
    ``save`` (atom_dim 32 / 64 / 128; the training forward): returns (out, saved) - the gates, the candidate
    and r * h of the listed rows for impnn_gated_update_rows_bwd_saved."""
    if _wants_grad(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, beta):
        from . import autograd
        return autograd.GatedUpdate.apply(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, beta, eps)
    require_gpu(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, beta)
    if h.shape != agg.shape:
        raise ValueError(f"atom_state {tuple(h.shape)} and agg {tuple(agg.shape)} differ")
    D = h.shape[-1]
    for name, w in (("dense_z", Wz), ("dense_r", Wr), ("dense_h", Wh)):
        if tuple(w.shape) != (2 * D, D):
            raise ValueError(f"{name} kernel must be (2D,D)=({2 * D},{D}), got {tuple(w.shape)}")
    ts = [f32c(t) for t in (h, agg, Wz, bz, Wr, br, Wh, bh, gamma, beta)]
    out = torch.empty_like(ts[0])
    nrows = ts[0].numel() // D
    with torch.cuda.device(h.device):
        if save:
            lib = _lib.load()
            saved = torch.empty(int(lib.impnn_gated_update_rows_saved_floats(nrows, D)), dtype=torch.float32,
                                device=h.device)
            ri, rn = (ptr(rows[0]), ptr(rows[1])) if rows is not None else (None, None)
            check(lib.impnn_gated_update_rows_train(*[ptr(t) for t in ts], float(eps), ptr(out), ri, rn, nrows, D,
                                                    ptr(saved), stream_ptr()))
            return out, saved
        if rows is not None and D in (32, 64, 128):
            check(_lib.load().impnn_gated_update_rows(*[ptr(t) for t in ts], float(eps), ptr(out), ptr(rows[0]),
                                                      ptr(rows[1]), nrows, D, stream_ptr()))
        else:
            check(_lib.load().impnn_gated_update(*[ptr(t) for t in ts], float(eps), ptr(out), nrows, D, stream_ptr()))
    return out


def global_sum_pool(h, atom_ids):
    """GlobalSumPool.call, models/layers.py:161-164."""
    if _wants_grad(h):
        from . import autograd
        return autograd.GlobalSumPool.apply(h, atom_ids)
    require_gpu(h, atom_ids)
    h, atom_ids = f32c(h), i32c(atom_ids)
    B, N, D = h.shape
    if tuple(atom_ids.shape) != (B, N):
        raise ValueError(f"atom_ids must be (B,N)=({B},{N}), got {tuple(atom_ids.shape)}")
    out = torch.empty(B, D, dtype=torch.float32, device=h.device)
    with torch.cuda.device(h.device):
        check(_lib.load().impnn_global_sum_pool(ptr(h), ptr(atom_ids), ptr(out), B, N, D, stream_ptr()))
    return out


def encoder_step_floats(D, K):
    return int(_lib.load().impnn_encoder_step_floats(D, K))


def pack_step_weights(steps):
    """steps: list of dicts with bond_transform,Wz,bz,Wr,br,Wh,bh,gamma,beta (torch tensors) ->
    one float32 tensor in the canonical layout of include/impnn.h."""
    parts = []
    for s in steps:
        for k in ("bond_transform", "Wz", "bz", "Wr", "br", "Wh", "bh", "gamma", "beta"):
            parts.append(f32c(s[k]).reshape(-1))
    return torch.cat(parts) if parts else None


_workspaces = {}


def _workspace(device, nbytes):
    """Encoder workspace of the CURRENT stream of `device`: calls on one stream reuse it in stream order; calls in
    flight on different streams (a caller overlapping consecutive batches) must not share one."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


ENCODER_MODES = {"f32": 0, "f16x2": 1, "f32t": 2, "f32x3": 3}
FP16_MAX = 65504.0
SPLIT_SX, SPLIT_SW = 16.0, 256.0  # kSX / kSW of encoder_fused.hip


def encoder_fused_supported(N, E, D, K, S, Vb, mode="f32t"):
    """Does the fused encoder cover this shape in this mode (impnn_encoder_workspace_bytes says so)?"""
    out = C.c_size_t(0)
    rc = _lib.load().impnn_encoder_workspace_bytes(1, 1, N, E, D, K, S, Vb, ENCODER_MODES[mode], 0, C.byref(out))
    return rc == 0


def encoder_overflow_possible(N, E, D, mode):
    """Can a batch of this padded shape hold a molecule that does not fit one chunk of the typed D = 32 encoder (more than
    256 kept rows, more than 512 valid edges, an in-degree above 255)?  Only then is the plan's overflow word read back
    (one 4-byte device-to-host copy per call)."""
    return D == 32 and mode in ("f32t", "f32x3") and (N > 256 or E > 255)


def _raise_on_overflow(ws):
    off = int(_lib.load().impnn_encoder_plan_overflow_offset())
    if int(ws[off:off + 4].view(torch.int32).item()) != 0:
        raise EncoderOverflow("a molecule of this batch does not fit one chunk of the fused encoder "
                              "(> 256 kept rows, > 512 valid edges or an in-degree > 255)")


class EncoderUnsupported(RuntimeError):
    pass


class EncoderOverflow(EncoderUnsupported):
    """The shape is covered, this BATCH is not (PlanHeader::overflow): the caller takes the layer-at-a-time path."""


def split_mode_degree_limit(atom_table, bond_table, steps, D):
    """Largest in-degree for which the encoder's "f16x2" mode provably stays inside fp16 range.

    With LayerNorm, |h_s| <= Hmax = max|atom_table| + S*(sqrt(D-1)*max|gamma| + max|beta|)
    (models/layers.py:154-155), |G| <= deg*max|bond_table|*Hmax and |agg_i| <= L1max(W)*|G|max, where
    L1max = max_i sum_{k,j}|W[k,i,j]|.  Mode 1 needs 16*max(|h|,|G|,|agg|) < 65504 and
    256*max|weights| < 65504.  Returns 0.0 when no degree is safe.  (One device sync: call when
    weights change, not per batch.)"""
    if not steps:
        return float("inf")
    gmax = max(float(s["gamma"].abs().max()) for s in steps)
    bmax = max(float(s["beta"].abs().max()) for s in steps)
    wmax = max(float(s[k].abs().max()) for s in steps for k in ("bond_transform", "Wz", "Wr", "Wh"))
    l1 = max(float(s["bond_transform"].abs().sum(dim=(0, 2)).max()) for s in steps)
    hmax = float(atom_table.abs().max()) + len(steps) * ((D - 1) ** 0.5 * gmax + bmax)
    cmax = float(bond_table.abs().max())
    if wmax * SPLIT_SW >= FP16_MAX or hmax * SPLIT_SX >= FP16_MAX:
        return 0.0
    per_deg = SPLIT_SX * cmax * hmax * max(1.0, l1)
    return float("inf") if per_deg == 0.0 else 0.999 * FP16_MAX / per_deg


def prepare_encoder_weights(packed, bond_table, D, K, num_steps, mode="f32t"):
    """Builds the encoder's kernel-side weight image once (impnn_encoder_prepare_weights); pass the
    result as `prepared` to encoder_fused while the weights stay unchanged.  Mode "f32t" folds the bond
    embedding table into the image (the per-bond-type matrices), so it must be rebuilt when that table changes."""
    require_gpu(packed, bond_table)
    packed, bond_table = f32c(packed), f32c(bond_table)
    S, Vb = int(num_steps), int(bond_table.shape[0])
    lib = _lib.load()
    nbytes = int(lib.impnn_encoder_prepared_bytes(D, S, Vb, ENCODER_MODES[mode]))
    out = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=packed.device)
    with torch.cuda.device(packed.device):
        rc = lib.impnn_encoder_prepare_weights(ptr(packed), ptr(bond_table), D, K, S, Vb, ENCODER_MODES[mode],
                                               ptr(out), nbytes, stream_ptr())
    if rc == _lib.IMPNN_E_UNSUPPORTED:
        raise EncoderUnsupported(lib.impnn_last_error_string().decode())
    check(rc)
    return out


def encoder_fused(ions, atom_table, bond_table, packed_weights, num_steps, eps=LN_EPS, mode="f32t", prepared=None,
                  workgroups=0):
    """encode() up to GlobalSumPool for 1 or 2 ion branches in one launch.

    ions: list of (atom_ids (B,N), bond_ids (B,E), conn (B,E,2)); packed_weights: list of packed
    step-weight tensors (pack_step_weights), or None when `prepared` (list of
    prepare_encoder_weights outputs built for the same `mode`) is given.  Returns a list of pooled
    (B,D) tensors.
    mode: "f32t" (per-bond-type messages, exact f32 MFMA, any bond_dim), "f32x3" (the same with the GatedUpdate GEMMs
    as exact three-term bf16 products on the bf16 matrix pipe; opt-in), "f32" (pull form, exact f32 MFMA,
    bond_dim <= 8) or "f16x2" (pull form, split-fp16 MFMA, f32 accumulate; the caller vouches for
    the range condition of include/impnn.h - ionic_mpnn_amd.model does via split_mode_degree_limit).
    workgroups: persistent workgroups of the launch (0: library default, one per CU).
    """
    n = len(ions)
    if n not in (1, 2):
        raise ValueError("1 or 2 ion branches")
    if mode not in ENCODER_MODES:
        raise ValueError(f"mode must be one of {sorted(ENCODER_MODES)}")
    atom_table, bond_table = f32c(atom_table), f32c(bond_table)
    require_gpu(atom_table, bond_table)
    dev = atom_table.device
    prepared_in = []
    for (a, b, c) in ions:
        require_gpu(a, b, c)
        prepared_in.append((i32c(a), i32c(b), i32c(c)))
    B, N = prepared_in[0][0].shape
    E = prepared_in[0][1].shape[1]
    for (a, b, c) in prepared_in:
        if tuple(a.shape) != (B, N) or tuple(b.shape) != (B, E) or tuple(c.shape) != (B, E, 2):
            raise ValueError("ion branches must share (B,N,E)")
    Va, D = atom_table.shape
    Vb, K = bond_table.shape
    S = int(num_steps)
    mode_i, wgs = ENCODER_MODES[mode], int(workgroups)
    lib = _lib.load()
    need = C.c_size_t(0)
    rc = lib.impnn_encoder_workspace_bytes(n, B, N, E, D, K, S, Vb, mode_i, wgs, C.byref(need))
    if rc == _lib.IMPNN_E_UNSUPPORTED:
        raise EncoderUnsupported(lib.impnn_last_error_string().decode())
    check(rc)
    if DEBUG_VALIDATE:
        for (a, b, c) in prepared_in:
            validate_indices(conn=c, atom_ids=a, bond_ids=b, N=N, Va=Va, Vb=Vb)
    ws = _workspace(dev, need.value)
    pooled = [torch.empty(B, D, dtype=torch.float32, device=dev) for _ in range(n)]
    arr = C.c_void_p * n
    mk = lambda ts: arr(*[t.data_ptr() if t is not None else 0 for t in ts])
    common = (mk([p[0] for p in prepared_in]), mk([p[1] for p in prepared_in]), mk([p[2] for p in prepared_in]),
              ptr(atom_table), Va, ptr(bond_table), Vb)
    with torch.cuda.device(dev):
        if prepared is not None:
            if len(prepared) != n or any(int(t.numel()) < int(lib.impnn_encoder_prepared_bytes(D, S, Vb, mode_i))
                                         for t in prepared):
                raise ValueError("prepared weight images do not match (n_ions, num_steps, bond vocabulary, mode)")
            check(lib.impnn_encoder_fused_prepared(n, *common, mk(prepared), mode_i, mk(pooled), B, N, E,
                                                   D, K, S, float(eps), wgs, ptr(ws), ws.numel(), stream_ptr()))
        else:
            ws_w = [f32c(w) if w is not None else None for w in packed_weights]
            step_f = encoder_step_floats(D, K)
            for w in ws_w:
                if S > 0 and (w is None or w.numel() != S * step_f):
                    raise ValueError(f"packed step weights must hold S*{step_f} floats")
            check(lib.impnn_encoder_fused(n, *common, mk(ws_w), mode_i, mk(pooled), B, N, E, D, K, S, float(eps), wgs,
                                          ptr(ws), ws.numel(), stream_ptr()))
    if B > 0 and encoder_overflow_possible(N, E, D, mode):
        _raise_on_overflow(ws)
    return pooled


# ---------------------------------------------------------------------------------------------
# pipelined use: plan of batch i+1 on a side stream while batch i is being encoded
# ---------------------------------------------------------------------------------------------
class EncoderPlan:
    """A planned batch: its workspace, the event that marks the plan complete, and the shapes."""
    __slots__ = ("slot", "ready", "ions", "shape", "n_ions", "info", "mode")


class EncoderPipeline:
    """Two (or more) plan workspaces and a side stream.  ``plan()`` enqueues the graph-only plan
    kernels of a batch on the side stream; ``run()`` enqueues the encoder kernel on torch's current
    stream after the plan's event.  A workspace is reused only after the run that read it is done
    (event-ordered, no host synchronisation)."""

    def __init__(self, device, depth=2):
        self.device = torch.device(device)
        self.side = torch.cuda.Stream(device=self.device)
        self.slots = [{"ws": None, "done": None} for _ in range(depth)]
        self.next = 0

    def plan(self, ions, D, K, S, Va, Vb, mode="f32t", workgroups=0):
        """mode: the mode the batch will be run in (modes "f32"/"f16x2" share one record kind, "f32t" has its own)."""
        n = len(ions)
        prep = []
        for (a, b, c) in ions:
            require_gpu(a, b, c)
            prep.append((i32c(a), i32c(b), i32c(c)))
        B, N = prep[0][0].shape
        E = prep[0][1].shape[1]
        lib = _lib.load()
        need = C.c_size_t(0)
        mode_i, wgs = ENCODER_MODES[mode], int(workgroups)
        rc = lib.impnn_encoder_workspace_bytes(n, B, N, E, D, K, S, Vb, mode_i, wgs, C.byref(need))
        if rc == _lib.IMPNN_E_UNSUPPORTED:
            raise EncoderUnsupported(lib.impnn_last_error_string().decode())
        check(rc)
        slot = self.slots[self.next]
        self.next = (self.next + 1) % len(self.slots)
        if slot["ws"] is None or slot["ws"].numel() < need.value:
            slot["ws"] = torch.empty(max(need.value, 1 << 20), dtype=torch.uint8, device=self.device)
        arr = C.c_void_p * n
        mk = lambda ts: arr(*[t.data_ptr() for t in ts])
        cur = torch.cuda.current_stream(self.device)
        self.side.wait_stream(cur)  # the inputs were produced on the current stream
        if slot["done"] is not None:
            self.side.wait_event(slot["done"])  # the encoder that last read this workspace
        with torch.cuda.device(self.device), torch.cuda.stream(self.side):
            info = _lib.PlanInfo()
            check(lib.impnn_encoder_plan(n, mk([p[0] for p in prep]), mk([p[1] for p in prep]), mk([p[2] for p in prep]),
                                         B, N, E, D, K, S, Va, Vb, mode_i, wgs, ptr(slot["ws"]), slot["ws"].numel(),
                                         C.c_void_p(self.side.cuda_stream), C.byref(info)))
            ready = torch.cuda.Event()
            ready.record(self.side)
        for trio in prep:
            for t in trio:
                t.record_stream(self.side)
        h = EncoderPlan()
        h.slot, h.ready, h.ions, h.shape, h.n_ions = slot, ready, prep, (B, N, E, D, K, S, Vb), n
        h.info, h.mode = info, mode
        return h

    def run(self, plan, atom_table, bond_table, prepared, mode=None, eps=LN_EPS):
        mode = plan.mode if mode is None else mode
        B, N, E, D, K, S, Vb = plan.shape
        n = plan.n_ions
        atom_table, bond_table = f32c(atom_table), f32c(bond_table)
        lib = _lib.load()
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(plan.ready)
        pooled = [torch.empty(B, D, dtype=torch.float32, device=self.device) for _ in range(n)]
        arr = C.c_void_p * n
        mk = lambda ts: arr(*[t.data_ptr() for t in ts])
        ws = plan.slot["ws"]
        with torch.cuda.device(self.device):
            check(lib.impnn_encoder_run(n, mk([p[0] for p in plan.ions]), ptr(atom_table), atom_table.shape[0],
                                        ptr(bond_table), Vb, mk(prepared), ENCODER_MODES[mode], mk(pooled), B, N, E, D,
                                        K, S, float(eps), C.byref(plan.info), ptr(ws), ws.numel(), stream_ptr()))
            done = torch.cuda.Event()
            done.record(cur)
        plan.slot["done"] = done
        if B > 0 and encoder_overflow_possible(N, E, D, mode):
            _raise_on_overflow(ws)
        return pooled


def gather_rows(srcs, dsts, rows):
    """dsts[t][r] = srcs[t][rows[r]] for up to 8 tensors in one launch (impnn_gather_rows): the mini-batch gather
    of model.fit from a device-resident data set.  rows: device int64; tensors: contiguous, 4-byte elements."""
    require_gpu(rows, *srcs, *dsts)
    if rows.dtype != torch.int64 or not rows.is_contiguous():
        raise TypeError("rows must be a contiguous int64 tensor")
    n_rows = rows.numel()
    rb = []
    for sx, dx in zip(srcs, dsts):
        if not (sx.is_contiguous() and dx.is_contiguous()) or sx.dtype != dx.dtype or sx.element_size() != 4:
            raise TypeError("gather_rows needs contiguous tensors of one 4-byte dtype per pair")
        if tuple(sx.shape[1:]) != tuple(dx.shape[1:]) or dx.shape[0] != n_rows:
            raise ValueError(f"gather_rows: shapes {tuple(sx.shape)} -> {tuple(dx.shape)} for {n_rows} rows")
        rb.append(sx[0].numel() * 4 if sx.shape[0] else 4)
    n = len(rb)
    st = (C.c_void_p * n)(*[t.data_ptr() for t in srcs])
    dt = (C.c_void_p * n)(*[t.data_ptr() for t in dsts])
    bt = (C.c_int64 * n)(*rb)
    with torch.cuda.device(rows.device):
        check(_lib.load().impnn_gather_rows(n, st, dt, bt, ptr(rows), n_rows, stream_ptr()))


def model_head(kind, pooled_cat, pooled_an, temperature, head_weights, fp_size, mixing_size):
    """Everything after GlobalSumPool in one launch (impnn_model_head): kind "viscosity" or "melting_point"."""
    require_gpu(pooled_cat, pooled_an, head_weights)
    pooled_cat, pooled_an, head_weights = f32c(pooled_cat), f32c(pooled_an), f32c(head_weights)
    B, D = pooled_cat.shape
    k = {"viscosity": 0, "melting_point": 1}[kind]
    lib = _lib.load()
    if head_weights.numel() != lib.impnn_model_head_floats(k, D, fp_size, mixing_size):
        raise ValueError("packed head weights have the wrong length")
    T = None
    if k == 0:
        require_gpu(temperature)
        T = f32c(temperature).reshape(-1)
        if T.numel() != B:
            raise ValueError("temperature must hold one value per sample")
    out = torch.empty(B, 1, dtype=torch.float32, device=pooled_cat.device)
    with torch.cuda.device(pooled_cat.device):
        check(lib.impnn_model_head(k, ptr(pooled_cat), ptr(pooled_an), ptr(T) if T is not None else None,
                                   ptr(head_weights), ptr(out), B, D, fp_size, mixing_size, stream_ptr()))
    return out
