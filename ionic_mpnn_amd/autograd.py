"""Autograd nodes of the layer-at-a-time path: each forward is the libimpnn entry of
the reference layer, each backward the matching ``impnn_*_bwd`` entry (csrc/train_kernels.hip).
``ops.*`` routes through these nodes whenever an input requires grad and grad mode is on; with no grad
the calls are the plain forward entries.  torch.autograd only keeps the graph - no torch op computes here.
A model's training pass uses the larger nodes at the end of this file (a batch-32 step is bound by the number of
launches): BondTypeMatricesAll (every layer's type matrices), MessagePassingStep (message -> Reduce -> GatedUpdate)
and ModelHeadLoss (head + mse + l2 penalties); the per-layer nodes serve the drop-in layers called one by one."""
from __future__ import annotations

import ctypes as C
import itertools
import threading

import torch

from . import _lib, ops
from ._lib import check, f32c, i32c, ptr, stream_ptr


class _PassState(threading.local):  # one scope stack per host thread
    def __init__(self):
        self.d = {"id": None}

    def __getitem__(self, k):
        return self.d[k]

    def __setitem__(self, k, v):
        self.d[k] = v


_pass = _PassState()
_pass_ids = itertools.count(1)  # ids are unique across threads (next() is atomic under the GIL)


class training_pass:
    """Scope of ONE differentiable forward pass (MPNNModel.__call__(training=True) opens it).  Work that depends on
    the batch's graph tensors only - the edge sort by bond type of the message backward - is shared by the nodes
    created inside one scope and never across scopes (the tensors may be refilled in place by kernels that torch's
    version counters do not see)."""

    def __enter__(self):
        self.prev = _pass["id"]
        _pass["id"] = next(_pass_ids)
        return self

    def __exit__(self, *exc):
        _pass["id"] = self.prev
        return False


def current_pass():
    return _pass["id"]


def _sink(param):
    """The existing gradient buffer of a leaf parameter, when a backward kernel may add into it directly
    (float32, contiguous, same device): ionic_mpnn_amd.train.Adam keeps every .grad as a view of one flat buffer.
    Returning None for that input afterwards tells autograd there is nothing left to accumulate."""
    g = getattr(param, "grad", None)
    if g is None or not param.is_leaf or g.dtype != torch.float32 or not g.is_contiguous() or g.device != param.device:
        return None
    return g


def _lib_call(device, fn, *args):
    with torch.cuda.device(device):
        check(fn(*args, stream_ptr()))


class EmbedGather(torch.autograd.Function):
    """Embedding lookup (train_viscosity.py:171-172)."""

    @staticmethod
    def forward(ctx, ids, table):
        ids = i32c(ids)
        ctx.save_for_backward(ids, table)
        return ops.embed_gather(ids, table)

    @staticmethod
    def backward(ctx, dout):
        ids, table = ctx.saved_tensors
        V, dim = table.shape
        sink = _sink(table)  # the kernel accumulates (atomics): add straight into the gradient buffer
        dtable = sink if sink is not None else torch.zeros(V, dim, dtype=torch.float32, device=dout.device)
        dout = f32c(dout)
        _lib_call(dout.device, _lib.load().impnn_embed_gather_bwd, ptr(ids), ptr(dout), ptr(dtable), ids.numel(), V, dim)
        return None, (None if sink is not None else dtable)


class BondTypeMatrices(torch.autograd.Function):
    """A[v] = sum_k Tb[v,k] W[k] (models/layers.py:108, once per vocabulary entry)."""

    @staticmethod
    def forward(ctx, bond_table, W):
        bond_table, W = f32c(bond_table), f32c(W)
        ctx.save_for_backward(bond_table, W)
        return ops.bond_type_matrices(bond_table, W)

    @staticmethod
    def backward(ctx, dmats):
        bond_table, W = ctx.saved_tensors
        Vb, K = bond_table.shape
        D = W.shape[-1]
        dmats = f32c(dmats)
        sw, st = _sink(W), _sink(bond_table)
        if sw is not None and st is not None and K < 64:
            _lib_call(W.device, _lib.load().impnn_bond_type_matrices_bwd, ptr(bond_table), ptr(W), ptr(dmats), ptr(sw),
                      ptr(st), Vb, K, D, 1)
            return None, None
        dW, dtb = torch.empty_like(W), torch.empty_like(bond_table)
        _lib_call(W.device, _lib.load().impnn_bond_type_matrices_bwd, ptr(bond_table), ptr(W), ptr(dmats), ptr(dW),
                  ptr(dtb), Vb, K, D, 0)
        return dtb, dW


class BondTypeMatricesAll(torch.autograd.Function):
    """The type matrices of all message layers (both ions, every step) as one node: one forward launch, two backward
    launches (impnn_bond_type_matrices_multi[_bwd]) instead of three per layer."""

    @staticmethod
    def forward(ctx, bond_table, *Ws):
        bond_table = f32c(bond_table)
        Ws = tuple(f32c(W) for W in Ws)
        Vb, K = bond_table.shape
        D = Ws[0].shape[-1]
        outs = tuple(torch.empty(Vb, D, D, dtype=torch.float32, device=bond_table.device) for _ in Ws)
        wt = (C.c_void_p * len(Ws))(*[W.data_ptr() for W in Ws])
        ot = (C.c_void_p * len(Ws))(*[o.data_ptr() for o in outs])
        _lib_call(bond_table.device, _lib.load().impnn_bond_type_matrices_multi, ptr(bond_table), wt, ot, len(Ws), Vb, K, D)
        ctx.save_for_backward(bond_table, *Ws)
        if any(W.requires_grad for W in Ws) or bond_table.requires_grad:
            pool = torch.zeros(len(Ws), Vb, D, D, dtype=torch.float32, device=bond_table.device)  # one fill per step
            for p, o in enumerate(outs):
                o._impnn_dmats = pool[p]
        return outs

    @staticmethod
    def backward(ctx, *dmats):
        bond_table, *Ws = ctx.saved_tensors
        Vb, K = bond_table.shape
        D = Ws[0].shape[-1]
        dmats = [f32c(d) if d is not None else torch.zeros(Vb, D, D, dtype=torch.float32, device=bond_table.device)
                 for d in dmats]
        sinks = [_sink(W) for W in Ws]
        st = _sink(bond_table)
        use_sinks = st is not None and all(sk is not None for sk in sinks)
        dWs = sinks if use_sinks else [torch.empty_like(W) for W in Ws]
        dtb = st if use_sinks else torch.empty_like(bond_table)
        n = len(Ws)
        wt = (C.c_void_p * n)(*[W.data_ptr() for W in Ws])
        dt = (C.c_void_p * n)(*[d.data_ptr() for d in dmats])
        gt = (C.c_void_p * n)(*[g.data_ptr() for g in dWs])
        lib = _lib.load()
        wsn = int(lib.impnn_bond_type_matrices_multi_bwd_workspace_floats(n, Vb, K, D))
        ws = torch.empty(max(wsn, 1), dtype=torch.float32, device=bond_table.device)
        _lib_call(bond_table.device, lib.impnn_bond_type_matrices_multi_bwd_ws, ptr(bond_table), wt, dt, gt, ptr(dtb),
                  n, Vb, K, D, 1 if use_sinks else 0, ptr(ws), wsn)
        if use_sinks:
            return (None,) * (n + 1)
        return (dtb, *dWs)


class BmmMessageTyped(torch.autograd.Function):
    """BondMatrixMessage.call in the per-bond-type schedule (models/layers.py:100-117)."""

    @staticmethod
    def forward(ctx, h, bond_ids, conn, type_mats):
        h, type_mats, bond_ids, conn = f32c(h), f32c(type_mats), i32c(bond_ids), i32c(conn)
        ctx.save_for_backward(h, bond_ids, conn, type_mats)
        ctx.graph_key = (conn, bond_ids, _pass["id"])  # the tensor OBJECTS (the S layers of one ion share them)
        return ops.bmm_message_typed(h, bond_ids, conn, type_mats)

    @staticmethod
    def backward(ctx, dm):
        h, bond_ids, conn, mats = ctx.saved_tensors
        B, N, D = h.shape
        E, Vb = conn.shape[1], mats.shape[0]
        dm = f32c(dm)
        both = torch.zeros(h.numel() + mats.numel(), dtype=torch.float32, device=h.device)  # one fill for both sums
        dh, dmats = both[:h.numel()].view_as(h), both[h.numel():].view_as(mats)
        lib = _lib.load()
        # the edge sort by bond type depends on (conn, bond_ids) only: shared inside a training pass
        holder, bond_obj, pass_id = ctx.graph_key  # the objects the forward saw (saved tensors may be re-wrapped)
        prev = _pass["id"]
        _pass["id"] = pass_id                       # backward runs outside the with-block: re-enter its pass
        try:
            ws, ready = ops.edge_sort_workspace(holder, bond_obj, B, E, Vb)
        finally:
            _pass["id"] = prev
        _lib_call(h.device, lib.impnn_bmm_message_typed_bwd, ptr(h), ptr(bond_ids), ptr(conn), ptr(mats),
                  ptr(dm), ptr(dh), ptr(dmats), ptr(ws), ws.numel(), B, N, E, D, Vb, 1 if ready else 0)
        return dh, None, None, dmats


class MessageReduceTyped(torch.autograd.Function):
    """Reduce o BondMatrixMessage as one node (models/layers.py:100-117 then :57-83): the forward runs the two entries
    and keeps no messages; the backward reads the aggregate's gradient at every edge's target row
    (impnn_message_reduce_typed_bwd), so neither the (B,E,D) messages nor their gradient outlive the forward."""

    @staticmethod
    def forward(ctx, h, bond_ids, conn, type_mats):
        h, type_mats, bond_ids, conn = f32c(h), f32c(type_mats), i32c(bond_ids), i32c(conn)
        ctx.save_for_backward(h, bond_ids, conn, type_mats)
        ctx.graph_key = (conn, bond_ids, _pass["id"])
        m = ops.bmm_message_typed(h, bond_ids, conn, type_mats)
        return ops.reduce_scatter_add(m, conn[:, :, 1], h.shape[1])

    @staticmethod
    def backward(ctx, dagg):
        h, bond_ids, conn, mats = ctx.saved_tensors
        B, N, D = h.shape
        E, Vb = conn.shape[1], mats.shape[0]
        dagg = f32c(dagg)
        both = torch.zeros(h.numel() + mats.numel(), dtype=torch.float32, device=h.device)
        dh, dmats = both[:h.numel()].view_as(h), both[h.numel():].view_as(mats)
        holder, bond_obj, pass_id = ctx.graph_key
        prev = _pass["id"]
        _pass["id"] = pass_id
        try:
            ws, ready = ops.edge_sort_workspace(holder, bond_obj, B, E, Vb)
        finally:
            _pass["id"] = prev
        _lib_call(h.device, _lib.load().impnn_message_reduce_typed_bwd, ptr(h), ptr(bond_ids), ptr(conn), ptr(mats),
                  ptr(dagg), ptr(dh), ptr(dmats), ptr(ws), ws.numel(), B, N, E, D, Vb, 1 if ready else 0)
        return dh, None, None, dmats


class ReduceScatterAdd(torch.autograd.Function):
    """Reduce.call (models/layers.py:57-83)."""

    @staticmethod
    def forward(ctx, messages, tgt_idx, num_atoms):
        tgt = i32c(tgt_idx)
        ctx.save_for_backward(tgt)
        ctx.num_atoms = int(num_atoms)
        return ops.reduce_scatter_add(messages, tgt, num_atoms)

    @staticmethod
    def backward(ctx, dagg):
        (tgt,) = ctx.saved_tensors
        B, E = tgt.shape
        dagg = f32c(dagg)
        D = dagg.shape[-1]
        dm = torch.empty(B, E, D, dtype=torch.float32, device=dagg.device)
        _lib_call(dagg.device, _lib.load().impnn_reduce_scatter_bwd, ptr(dagg), ptr(tgt), 1, ptr(dm), B, ctx.num_atoms, E, D)
        return dm, None, None


class GatedUpdate(torch.autograd.Function):
    """GatedUpdate.call (models/layers.py:142-156)."""

    @staticmethod
    def forward(ctx, h, agg, Wz, bz, Wr, br, Wh, bh, gamma, beta, eps):
        ts = [f32c(t) for t in (h, agg, Wz, bz, Wr, br, Wh, bh, gamma)]
        ctx.save_for_backward(*ts, beta)
        ctx.eps = float(eps)
        return ops.gated_update(*ts, beta, eps)

    @staticmethod
    def backward(ctx, dout):
        return _gated_update_backward(ctx.saved_tensors, ctx.eps, dout)


def _gated_update_backward(saved, eps, dout, row_list=None, kept=None, unlisted_undefined=False):
    """(dh, dagg, 8 parameter gradients or None where the kernel added into the sink, None for eps).
    row_list = (row_index, n_rows) of ops.kept_row_index: gradients of those rows only (impnn_gated_update_rows_bwd);
    dh is zero elsewhere (padding atoms carry no gradient), dagg is undefined there and never read.
    kept: the buffer the training forward filled (ops.gated_update(.., save=True)) - the backward then skips its
    recompute passes (impnn_gated_update_rows_bwd_saved) and overwrites the buffer."""
    h, agg, Wz, bz, Wr, br, Wh, bh, gamma, beta = saved
    D = h.shape[-1]
    rows = h.numel() // D
    lib = _lib.load()
    dout = f32c(dout)
    dh = torch.zeros_like(h) if row_list is not None and not unlisted_undefined else torch.empty_like(h)
    dagg = torch.empty_like(agg)
    P = int(lib.impnn_gated_update_param_floats(D))
    if row_list is not None or (kept is not None and D != 32):
        wsn = int(lib.impnn_gated_update_rows_bwd_workspace_floats(rows, D))
    else:
        wsn = int(lib.impnn_gated_update_bwd_workspace_floats(rows, D))
    ws = torch.empty(max(wsn, 1), dtype=torch.float32, device=h.device)

    def call(dparams, accumulate):
        common = (ptr(h), ptr(agg), ptr(Wz), ptr(bz), ptr(Wr), ptr(br), ptr(Wh), ptr(bh), ptr(gamma), eps, ptr(dout),
                  ptr(dh), ptr(dagg), ptr(dparams), ptr(ws), wsn)
        if kept is not None:
            ri, rn = (ptr(row_list[0]), ptr(row_list[1])) if row_list is not None else (None, None)
            _lib_call(h.device, lib.impnn_gated_update_rows_bwd_saved, *common, ri, rn, rows, D, accumulate, ptr(kept))
        elif row_list is not None:
            _lib_call(h.device, lib.impnn_gated_update_rows_bwd, *common, ptr(row_list[0]), ptr(row_list[1]), rows, D,
                      accumulate)
        else:
            _lib_call(h.device, lib.impnn_gated_update_bwd, *common, rows, D, accumulate)
    # the eight parameter gradients leave the kernel as one block in the canonical order; when the existing
    # .grad buffers form exactly that block (train.Adam's flat buffer does), the kernel adds into it directly
    params = (Wz, bz, Wr, br, Wh, bh, gamma, beta)
    sinks = [_sink(t) for t in params]
    direct = all(g is not None for g in sinks)
    if direct:
        base, off = sinks[0].data_ptr(), 0
        for t, g in zip(params, sinks):
            direct = direct and g.data_ptr() == base + 4 * off and g.numel() == t.numel()
            off += t.numel()
    if direct:
        call(sinks[0], 1)
        return (dh, dagg, *([None] * 8), None)
    dparams = torch.empty(P, dtype=torch.float32, device=h.device)
    call(dparams, 0)
    n_w, o = 2 * D * D, 0
    grads = []
    for _ in range(3):
        grads.append(dparams[o:o + n_w].view(2 * D, D))
        grads.append(dparams[o + n_w:o + n_w + D])
        o += n_w + D
    grads.append(dparams[o:o + D])
    grads.append(dparams[o + D:o + 2 * D])
    return (dh, dagg, *grads, None)


# edge slots per ion from which the message adjoint writes per-edge vectors and sums them in slot order instead of adding
# into dh with float atomics (impnn_message_reduce_typed_bwd_scratch): 15.5 -> 14.4 ms per step at batch 4096, 2.25 -> 1.89
# ms at batch 256; at the reference's batch 32 the second launch costs more than the atomics (1.13 -> 1.19 ms)
MESSAGE_BWD_EDGE_BUFFER_MIN_SLOTS = 8192


class MessagePassingStep(torch.autograd.Function):
    """One message-passing step as one node (train_viscosity.py:179-186: BondMatrixMessage -> Reduce -> GatedUpdate).
    Backward: impnn_gated_update_bwd writes dh and dagg, then impnn_message_reduce_typed_bwd ADDS the message path's
    share into the same dh - no separate sum of the two consumers of h, no zero fill of dh; the type-matrix gradient
    goes into a buffer BondTypeMatricesAll zeroed for all layers at once (``type_mats._impnn_dmats``) when there is one."""

    @staticmethod
    def forward(ctx, h, bond_ids, conn, type_mats, Wz, bz, Wr, br, Wh, bh, gamma, beta, eps, row_index=None,
                n_rows=None, inner=False):
        """row_index / n_rows (ops.kept_row_index; atom_dim 64 / 128): GatedUpdate forward and backward on the kept rows
        only - padding atoms reach neither a message nor the pool, so their rows of the output are left undefined
        and their gradient is zero (include/impnn.h, impnn_gated_update_rows[_bwd])."""
        h, type_mats, bond_ids, conn = f32c(h), f32c(type_mats), i32c(bond_ids), i32c(conn)
        gu = [f32c(t) for t in (Wz, bz, Wr, br, Wh, bh, gamma)]
        # inner (with a row list): this step's input is the output of another MessagePassingStep on the SAME list - that
        # node reads its incoming gradient at the listed rows only, so the rows outside the list need no zero fill here
        ctx.inner = bool(inner) and row_index is not None
        if h.shape[-1] != 32:
            # one message buffer per ion and pass (the Reduce behind each layer consumes it at once; at atom_dim 64 / 128
            # the message adjoint writes its per-edge vectors there)
            buf, reused = ops.message_scratch(conn, bond_ids, h.shape[0], conn.shape[1], h.shape[-1])
            m = ops.bmm_message_typed(h, bond_ids, conn, type_mats, out=buf, out_reused=reused)
            del buf
        else:
            m = ops.bmm_message_typed(h, bond_ids, conn, type_mats)
        agg = ops.reduce_scatter_add(m, conn[:, :, 1], h.shape[1])
        del m
        ctx.row_list = (row_index, n_rows) if row_index is not None else None
        ctx.kept = None
        if h.shape[-1] in (64, 128) or (h.shape[-1] == 32 and ctx.row_list is None):
            # the gates, the candidate and r * h of the (kept) rows stay for the backward (4 D floats per row and step)
            # instead of being recomputed there with half of its matrix work
            out, ctx.kept = ops.gated_update(h, agg, *gu, beta, eps, rows=ctx.row_list, save=True)
        else:
            out = ops.gated_update(h, agg, *gu, beta, eps, rows=ctx.row_list)
        ctx.save_for_backward(h, agg, *gu, beta, bond_ids, conn, type_mats)
        ctx.eps = float(eps)
        ctx.graph_key = (conn, bond_ids, _pass["id"])
        ctx.dmats_buf = getattr(type_mats, "_impnn_dmats", None)
        return out

    @staticmethod
    def backward(ctx, dout):
        saved = ctx.saved_tensors
        h, bond_ids, conn, mats = saved[0], saved[10], saved[11], saved[12]
        kept = ctx.kept
        if kept is not None:
            if kept is False:
                raise RuntimeError("MessagePassingStep: the kept activations were consumed by an earlier backward "
                                   "(run the forward again instead of retain_graph)")
            ctx.kept = False
        dh, dagg, *dparams = _gated_update_backward(saved[:10], ctx.eps, dout, ctx.row_list, kept, ctx.inner)
        del kept
        B, N, D = h.shape
        E, Vb = conn.shape[1], mats.shape[0]
        dmats = ctx.dmats_buf if ctx.dmats_buf is not None else torch.zeros_like(mats)
        holder, bond_obj, pass_id = ctx.graph_key
        prev = _pass["id"]
        _pass["id"] = pass_id
        scratch, kept_zero = None, False
        try:
            ws, ready = ops.edge_sort_workspace(holder, bond_obj, B, E, Vb)
            # (atom_dim 32: measured slower that way - 1.71 -> 1.88 ms per step at batch 4096 - the rows are a quarter as
            #  long, the atomics a quarter as many, and the buffer's round trip through HBM costs the same launch)
            if D in (64, 128) and B * E >= MESSAGE_BWD_EDGE_BUFFER_MIN_SLOTS:
                # the forward's message buffer of this ion and pass: zero rows at masked edges, free since the Reduce
                scratch, kept_zero = ops.message_scratch(holder, bond_obj, B, E, D)
        finally:
            _pass["id"] = prev
        if scratch is not None and kept_zero:
            _lib_call(h.device, _lib.load().impnn_message_reduce_typed_bwd_scratch, ptr(h), ptr(bond_ids), ptr(conn),
                      ptr(mats), ptr(dagg), ptr(dh), ptr(dmats), ptr(ws), ws.numel(), ptr(scratch), B, N, E, D, Vb,
                      1 if ready else 0)
        else:
            _lib_call(h.device, _lib.load().impnn_message_reduce_typed_bwd, ptr(h), ptr(bond_ids), ptr(conn), ptr(mats),
                      ptr(dagg), ptr(dh), ptr(dmats), ptr(ws), ws.numel(), B, N, E, D, Vb, 1 if ready else 0)
        return (dh, None, None, dmats, *dparams, None, None, None)


class GlobalSumPool(torch.autograd.Function):
    """GlobalSumPool.call (models/layers.py:161-164)."""

    @staticmethod
    def forward(ctx, h, atom_ids):
        ids = i32c(atom_ids)
        ctx.save_for_backward(ids)
        ctx.D = int(h.shape[-1])
        return ops.global_sum_pool(h, ids)

    @staticmethod
    def backward(ctx, dp):
        (ids,) = ctx.saved_tensors
        B, N = ids.shape
        dp = f32c(dp)
        dh = torch.empty(B, N, ctx.D, dtype=torch.float32, device=dp.device)
        _lib_call(dp.device, _lib.load().impnn_global_sum_pool_bwd, ptr(dp), ptr(ids), ptr(dh), B, N, ctx.D)
        return dh, None


class ModelHead(torch.autograd.Function):
    """Everything after GlobalSumPool as one node (SURVEY.md 8 f1 + f4): forward impnn_model_head_tensors, backward
    impnn_model_head_bwd, both reading the individual Dense kernels/biases (train_viscosity.py:189-214 /
    train_melting_point.py:173-198)."""

    @staticmethod
    def forward(ctx, kind, fp_size, mixing_size, pooled_cat, pooled_an, temperature, *weights):
        pooled_cat, pooled_an = f32c(pooled_cat), f32c(pooled_an)
        weights = tuple(f32c(w) for w in weights)
        B, D = pooled_cat.shape
        T = f32c(temperature).reshape(-1) if kind == 0 else None
        if T is not None and T.numel() != B:
            raise ValueError("temperature must hold one value per sample")
        out = torch.empty(B, 1, dtype=torch.float32, device=pooled_cat.device)
        table = (C.c_void_p * len(weights))(*[w.data_ptr() for w in weights])
        _lib_call(pooled_cat.device, _lib.load().impnn_model_head_tensors, kind, ptr(pooled_cat), ptr(pooled_an),
                  ptr(T) if T is not None else None, table, ptr(out), B, D, fp_size, mixing_size)
        ctx.save_for_backward(pooled_cat, pooled_an, *(() if T is None else (T,)), *weights)
        ctx.meta = (kind, fp_size, mixing_size, weights)
        return out

    @staticmethod
    def backward(ctx, dout):
        kind, F, Mx, params = ctx.meta
        saved = ctx.saved_tensors
        pooled_cat, pooled_an = saved[0], saved[1]
        T = saved[2] if kind == 0 else None
        weights = saved[3 if kind == 0 else 2:]
        B, D = pooled_cat.shape
        dout = f32c(dout).reshape(-1)
        sinks = [_sink(p) for p in params]
        grads = [s if s is not None else torch.zeros_like(w) for s, w in zip(sinks, weights)]
        dpc, dpa = torch.empty_like(pooled_cat), torch.empty_like(pooled_an)
        wt = (C.c_void_p * len(weights))(*[w.data_ptr() for w in weights])
        gt = (C.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
        _lib_call(dout.device, _lib.load().impnn_model_head_bwd, kind, ptr(pooled_cat), ptr(pooled_an),
                  ptr(T) if T is not None else None, wt, ptr(dout), ptr(dpc), ptr(dpa), gt, B, D, F, Mx)
        return (None, None, None, dpc, dpa, None) + tuple(None if s is not None else g for s, g in zip(sinks, grads))


class ModelHeadLoss(torch.autograd.Function):
    """Head + keras "mse" + l2 penalties as one node -> the scalar loss (impnn_model_head_loss[_bwd]).  ``l2``: one
    lambda per weight tensor; ``workspace``: a persistent float tensor whose first word is zero (the kernel's arrival
    counter; see include/impnn.h)."""

    @staticmethod
    def forward(ctx, kind, fp_size, mixing_size, l2, workspace, pooled_cat, pooled_an, temperature, y, *weights):
        pooled_cat, pooled_an, y = f32c(pooled_cat), f32c(pooled_an), f32c(y).reshape(-1)
        weights = tuple(f32c(w) for w in weights)
        B, D = pooled_cat.shape
        T = f32c(temperature).reshape(-1) if kind == 0 else None
        if y.numel() != B or (T is not None and T.numel() != B):
            raise ValueError("y / temperature must hold one value per sample")
        lib = _lib.load()
        if workspace.numel() < lib.impnn_model_head_loss_workspace_floats(B):
            raise ValueError("loss workspace too small")
        loss = torch.empty((), dtype=torch.float32, device=pooled_cat.device)
        lam = (C.c_float * len(weights))(*[float(v) for v in l2])
        table = (C.c_void_p * len(weights))(*[w.data_ptr() for w in weights])
        _lib_call(pooled_cat.device, lib.impnn_model_head_loss, kind, ptr(pooled_cat), ptr(pooled_an),
                  ptr(T) if T is not None else None, table, lam, ptr(y), None, ptr(loss), ptr(workspace),
                  workspace.numel(), B, D, fp_size, mixing_size)
        ctx.save_for_backward(pooled_cat, pooled_an, y, *(() if T is None else (T,)), *weights)
        ctx.meta = (kind, fp_size, mixing_size, tuple(float(v) for v in l2), weights)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        kind, F, Mx, l2, params = ctx.meta
        saved = ctx.saved_tensors
        pooled_cat, pooled_an, y = saved[0], saved[1], saved[2]
        T = saved[3] if kind == 0 else None
        weights = saved[4 if kind == 0 else 3:]
        B, D = pooled_cat.shape
        dloss = f32c(dloss).reshape(1)
        sinks = [_sink(p) for p in params]
        grads = [s if s is not None else torch.zeros_like(w) for s, w in zip(sinks, weights)]
        dpc, dpa = torch.empty_like(pooled_cat), torch.empty_like(pooled_an)
        lam = (C.c_float * len(weights))(*l2)
        wt = (C.c_void_p * len(weights))(*[w.data_ptr() for w in weights])
        gt = (C.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
        _lib_call(dloss.device, _lib.load().impnn_model_head_loss_bwd, kind, ptr(pooled_cat), ptr(pooled_an),
                  ptr(T) if T is not None else None, wt, lam, ptr(y), ptr(dloss), ptr(dpc), ptr(dpa), gt, B, D, F, Mx)
        return (None, None, None, None, None, dpc, dpa, None, None) + tuple(
            None if s is not None else g for s, g in zip(sinks, grads))
