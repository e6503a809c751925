"""build_model() of the reference's trainers, wired from ionic_mpnn_amd.layers.

``build_model`` (viscosity, train_viscosity.py:139-231) and ``build_melting_point_model``
(train_melting_point.py:137-215) keep the reference's signatures/defaults and create the layers in
the reference's order, so Keras-style auto names (gated_update_2, dense_3, ...) line up.

Two execution schedules, both on the HIP library:
  * ``fused=True`` (default when the shape is covered): one impnn_encoder_fused launch computes
    both ions' encode() up to GlobalSumPool;
  * ``fused=False``: layer at a time through the drop-in layers (a1..a8), tensor boundaries
    identical to the reference's; the two ions' chains run on two HIP streams (batches up to 4096).
Everything after GlobalSumPool is one launch (impnn_model_head); training goes through the larger
autograd nodes of ionic_mpnn_amd.autograd (f4), the torch-op head remains for traces and widths the kernels do not cover.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import layers as L
from . import ops

INPUT_NAMES = ["cat_atom", "cat_bond", "cat_connectivity", "an_atom", "an_bond", "an_connectivity", "temperature"]


# largest batch whose two ion chains run on two HIP streams in the layer-at-a-time path (MPNNModel._encode_two_streams).
# Measured on MI355X (training step, D=32): faster at every batch size, 0.47 -> 0.37 ms at 32 and 2.19 -> 1.88 ms at
# 4096; layered inference 7.7 -> 8.5 M pairs/s at 4096 and D=128 forward 9.8 -> 9.4 ms, but 1.12 -> 1.18 ms at batch
# 8192 (config 3), where every kernel already fills the chip.  IMPNN_TWO_STREAM_MAX_BATCH=0 switches it off.
TWO_STREAM_MAX_BATCH = 4096
TRAIN_ROW_LIST_MIN_ROWS = int(os.environ.get("IMPNN_TRAIN_ROW_LIST_MIN_ROWS", 4096))


class MPNNModel:
    def __init__(self, kind, atom_vocab_size, bond_vocab_size, atom_dim, bond_dim, fp_size, mixing_size,
                 num_steps, fp_l2, device=None, name=None):
        self.kind = kind
        self.name = name or ("MeltingPoint_MPNN" if kind == "melting_point" else "model")
        self.atom_vocab_size, self.bond_vocab_size = int(atom_vocab_size), int(bond_vocab_size)
        self.atom_dim, self.bond_dim = int(atom_dim), int(bond_dim)
        self.fp_size, self.mixing_size, self.num_steps = int(fp_size), int(mixing_size), int(num_steps)
        self.fp_l2 = float(fp_l2)
        self.device = device or L.default_device()
        dev = dict(device=self.device)
        D, K, S = self.atom_dim, self.bond_dim, self.num_steps
        # shared embeddings (train_viscosity.py:163-164)
        self.atom_emb = L.Embedding(atom_vocab_size, D, mask_zero=False, **dev)
        self.bond_emb = L.Embedding(bond_vocab_size, K, mask_zero=False, lazy=True, **dev)
        self.branches = {}
        for p in ("cat", "an"):  # encode(), train_viscosity.py:166-190, called at :193-194
            br = {"bmm": [], "reduce": [], "update": []}
            for i in range(S):
                br["bmm"].append(L.BondMatrixMessage(D, K, name=f"{p}_bmm_{i}", **dev))
                br["reduce"].append(L.Reduce(name=f"{p}_reduce_{i}", **dev))
                br["update"].append(L.GatedUpdate(D, **dev))
            br["pool"] = L.GlobalSumPool(**dev)
            br["fp"] = L.Dense(fp_size, activation="relu", kernel_regularizer=("l2", fp_l2), **dev)
            self.branches[p] = br
        self.cat_proj = L.Dense(mixing_size, activation="relu", **dev)  # :197
        self.an_proj = L.Dense(mixing_size, activation="relu", **dev)   # :198
        if kind == "viscosity":
            self.mix = L.AddTwoTensors(name="mix_cat_an", **dev)        # :201
            self.visc_params = L.Dense(3, **dev)                        # :204
            self.param_A = L.SliceParamA(name="param_A", **dev)
            self.param_B = L.SliceParamB(name="param_B", **dev)
            self.param_C = L.SliceParamC(name="param_C", **dev)
            self.scale_T = L.ScaleTemperature(name="scale_T", **dev)
            self.log_eta = L.ComputeLogEta(name="log_eta", **dev)
        else:
            self.mix = L.AddTwoTensors(name="add", **dev)               # keras Add(), train_melting_point.py:191
            self.mp_hidden = L.Dense(fp_size, activation="relu", kernel_regularizer=("l2", fp_l2), **dev)  # :197
            self.mp_out = L.Dense(1, **dev)                             # :198
        self._build_all()
        self._packed = None
        self._prepared = {}
        self._split_deg_limit = None
        self.encoder_mode = "auto"  # "auto" | "f32t" | "f32x3" | "f32" | "f16x2" (ops.encoder_fused)
        self.encoder_workgroups = 0  # persistent workgroups per encoder launch (0: library default, one per CU)

    # ------------------------------------------------------------------ construction
    def _build_all(self):
        D, K = self.atom_dim, self.bond_dim
        for lyr, shape in ((self.atom_emb, (None, None)), (self.bond_emb, (None, None))):
            lyr.build(shape); lyr.built = True
        for p in ("cat", "an"):
            br = self.branches[p]
            for i in range(self.num_steps):
                br["bmm"][i].build([(None, None, D), (None, None, K), (None, None, 2)]); br["bmm"][i].built = True
                br["update"][i].build([(None, None, D), (None, None, D)]); br["update"][i].built = True
                br["reduce"][i].built = True
            br["pool"].built = True
            br["fp"].build((None, D)); br["fp"].built = True
        for lyr in (self.cat_proj, self.an_proj):
            lyr.build((None, self.fp_size)); lyr.built = True
        if self.kind == "viscosity":
            self.visc_params.build((None, self.mixing_size)); self.visc_params.built = True
        else:
            self.mp_hidden.build((None, self.mixing_size)); self.mp_hidden.built = True
            self.mp_out.build((None, self.fp_size)); self.mp_out.built = True

    @property
    def layers(self):
        out = [self.atom_emb, self.bond_emb]
        for p in ("cat", "an"):
            br = self.branches[p]
            for i in range(self.num_steps):
                out += [br["bmm"][i], br["reduce"][i], br["update"][i]]
            out += [br["pool"], br["fp"]]
        out += [self.cat_proj, self.an_proj, self.mix]
        if self.kind == "viscosity":
            out += [self.visc_params, self.param_A, self.param_B, self.param_C, self.scale_T, self.log_eta]
        else:
            out += [self.mp_hidden, self.mp_out]
        return out

    def get_layer(self, name):
        for lyr in self.layers:
            if lyr.name == name:
                return lyr
        raise ValueError(f"No such layer: {name}")

    # ------------------------------------------------------------------ weights
    def _named_tensors(self):
        t = {"atom_embedding": self.atom_emb.embeddings, "bond_embedding": self.bond_emb.embeddings}
        for p in ("cat", "an"):
            br = self.branches[p]
            for i in range(self.num_steps):
                t[f"{p}_bmm_{i}/bond_transform"] = br["bmm"][i].bond_transform
                for wn, w in br["update"][i]._weights.items():
                    t[f"{p}_gu_{i}/{wn}"] = w
            t[f"{p}_fp/kernel"], t[f"{p}_fp/bias"] = br["fp"].kernel, br["fp"].bias
        t["cat_proj/kernel"], t["cat_proj/bias"] = self.cat_proj.kernel, self.cat_proj.bias
        t["an_proj/kernel"], t["an_proj/bias"] = self.an_proj.kernel, self.an_proj.bias
        if self.kind == "viscosity":
            t["visc_params/kernel"], t["visc_params/bias"] = self.visc_params.kernel, self.visc_params.bias
        else:
            t["mp_hidden/kernel"], t["mp_hidden/bias"] = self.mp_hidden.kernel, self.mp_hidden.bias
            t["mp_out/kernel"], t["mp_out/bias"] = self.mp_out.kernel, self.mp_out.bias
        return t

    def state_dict(self):
        return {k: v.detach().cpu().numpy().copy() for k, v in self._named_tensors().items()}

    def load_weights(self, weights):
        """weights: dict name -> array in the naming of ionic_mpnn_amd.weights, or the path of a save_weights file."""
        if isinstance(weights, (str, bytes)) or hasattr(weights, "__fspath__"):
            weights = self.load_weight_file(weights)[1]
        named = self._named_tensors()
        missing = sorted(set(named) - set(weights))
        if missing:
            raise KeyError(f"missing weights: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        for k, t in named.items():
            a = np.ascontiguousarray(np.asarray(weights[k], dtype=np.float32))
            if tuple(a.shape) != tuple(t.shape):
                raise ValueError(f"{k}: shape {a.shape} != {tuple(t.shape)}")
            with torch.no_grad():
                t.copy_(torch.from_numpy(a))
        self.invalidate_packed_weights()

    # ---- f3: configuration and weight files (the .keras archive itself needs an HDF5 reader: not here)
    def get_config(self):
        """Constructor arguments + per-layer configs (keras Model.get_config analogue, models/layers.py:119-125)."""
        return {"name": self.name, "kind": self.kind, "atom_vocab_size": self.atom_vocab_size,
                "bond_vocab_size": self.bond_vocab_size, "atom_dim": self.atom_dim, "bond_dim": self.bond_dim,
                "fp_size": self.fp_size, "mixing_size": self.mixing_size, "num_steps": self.num_steps,
                "fp_l2": self.fp_l2,
                "layers": [{"class_name": type(l).__name__, "config": l.get_config()} for l in self.layers]}

    @classmethod
    def from_config(cls, config, device=None):
        L.reset_uids()  # keras auto-names (gated_update_3, ...) restart with a new model graph
        return cls(config["kind"], config["atom_vocab_size"], config["bond_vocab_size"], config["atom_dim"],
                   config["bond_dim"], config["fp_size"], config["mixing_size"], config["num_steps"],
                   config.get("fp_l2", 1e-4), device=device, name=config.get("name"))

    def save_weights(self, path):
        """All variables under their Keras-style names (ionic_mpnn_amd.weights) plus the config, as one .npz."""
        import json
        arrays = self.state_dict()
        cfg = {k: v for k, v in self.get_config().items() if k != "layers"}
        # through a file handle: np.savez(path, ...) appends ".npz" to any other suffix, and the reference flow saves
        # to "models/viscosity_final.keras" and loads that very name (train_melting_point_transfer.py:78)
        with open(path, "wb") as f:
            np.savez(f, __config__=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8), **arrays)

    def save(self, path):
        """model.save("models/viscosity_final.keras") (train_viscosity.py:354) - here the npz container of save_weights
        (config + variables) under exactly the name given; the Keras zip/HDF5 container is not written."""
        self.save_weights(path)

    @staticmethod
    def load_weight_file(path):
        """-> (config dict, weights dict) from a file written by save_weights (no pickle is involved)."""
        import json
        import os as _os
        if not _os.path.exists(path) and _os.path.exists(str(path) + ".npz"):
            path = str(path) + ".npz"  # files written before round 2 (np.savez appended the suffix then)
        with np.load(path, allow_pickle=False) as z:
            cfg = json.loads(bytes(z["__config__"].tobytes()).decode())
            return cfg, {k: z[k] for k in z.files if k != "__config__"}

    def invalidate_packed_weights(self):
        """Call after mutating layer weights in place; the fused encoder caches a packed copy."""
        self._packed = None
        self._prepared = {}
        self._split_deg_limit = None
        self._head_packed = None
        for p in ("cat", "an"):
            for lyr in self.branches[p]["bmm"]:
                lyr.invalidate_cache()

    def _head_tensors(self):
        """Head weight tensors in the order of impnn_model_head's packed layout (include/impnn.h)."""
        parts = []
        for p in ("cat", "an"):
            parts += [self.branches[p]["fp"].kernel, self.branches[p]["fp"].bias]
        parts += [self.cat_proj.kernel, self.cat_proj.bias, self.an_proj.kernel, self.an_proj.bias]
        if self.kind == "viscosity":
            parts += [self.visc_params.kernel, self.visc_params.bias]
        else:
            parts += [self.mp_hidden.kernel, self.mp_hidden.bias, self.mp_out.kernel, self.mp_out.bias]
        return parts

    def _packed_head(self):
        """Head weights in the layout of impnn_model_head (include/impnn.h), cached per weight version."""
        if getattr(self, "_head_packed", None) is None:
            with torch.no_grad():
                self._head_packed = torch.cat([t.reshape(-1) for t in self._head_tensors()]).contiguous()
        return self._head_packed

    def _prepared_weights(self, mode):
        """Kernel-side weight images (one per ion), built once per weight version and mode."""
        if mode not in self._prepared:
            self._prepared[mode] = [ops.prepare_encoder_weights(pk, self.bond_emb.embeddings, self.atom_dim,
                                                                self.bond_dim, self.num_steps, mode)
                                    if pk is not None else None for pk in self._packed_weights()]
        return self._prepared[mode]

    def _packed_weights(self):
        if self._packed is None:
            packed = []
            for p in ("cat", "an"):
                br = self.branches[p]
                steps = []
                for i in range(self.num_steps):
                    w = br["update"][i]._weights
                    steps.append({"bond_transform": br["bmm"][i].bond_transform,
                                  "Wz": w["dense_z/kernel"], "bz": w["dense_z/bias"],
                                  "Wr": w["dense_r/kernel"], "br": w["dense_r/bias"],
                                  "Wh": w["dense_h/kernel"], "bh": w["dense_h/bias"],
                                  "gamma": w["layernorm/gamma"], "beta": w["layernorm/beta"]})
                packed.append(ops.pack_step_weights(steps))
                lim = ops.split_mode_degree_limit(self.atom_emb.embeddings, self.bond_emb.embeddings, steps,
                                                  self.atom_dim)
                self._split_deg_limit = lim if self._split_deg_limit is None else min(self._split_deg_limit, lim)
            self._packed = packed
        return self._packed

    def resolve_encoder_mode(self, N, E=None):
        """The fused encoder mode for (N, E)-shaped ion inputs, or None when no mode covers them.
        "auto" picks exact-f32 arithmetic only: "f32t" (per-bond-type messages, any bond_dim) first, then "f32"
        (pull form, bond_dim <= 8).  "f16x2" (split-fp16 products, narrower than f32) is used on request only, and
        then only while its static range bound holds for every possible in-degree (<= E edge slots); otherwise the
        request falls back to the exact modes.  "f32x3" (f32 as bf16x9: as exact as f32, another summation order) is used
        on request where the shape allows it and falls back to "f32t" where it does not."""
        if E is None:  # (older call sites passed E alone)
            N, E = 1, N
        sup = lambda m: ops.encoder_fused_supported(N, E, self.atom_dim, self.bond_dim, self.num_steps,
                                                    self.bond_vocab_size, m)
        want = self.encoder_mode
        if want == "f16x2":
            self._packed_weights()
            if sup("f16x2") and (self._split_deg_limit is None or E <= self._split_deg_limit):
                return "f16x2"
            want = "auto"
        if want == "f32x3" and not sup("f32x3") and sup("f32t"):
            return "f32t"  # a shape only the exact-f32 form of the same path takes
        if want in ("f32t", "f32", "f32x3"):
            return want if sup(want) else None
        for m in ("f32t", "f32"):
            if sup(m):
                return m
        return None

    # ------------------------------------------------------------------ forward
    def fused_supported(self, N, E):
        return self.resolve_encoder_mode(N, E) is not None

    def _builds_graph(self):
        """True when this call is differentiated: grad mode on and some variable asks for a gradient."""
        return torch.is_grad_enabled() and any(t.requires_grad for _, t in self.trainable_variables())

    def _all_type_matrices(self):
        """Training: the type matrices of every message layer from one node (3 launches per step instead of 3 per
        layer); None where the per-layer entries are the better fit (bond_dim >= 64 is GEMM-shaped)."""
        if self.bond_dim >= 64 or self.num_steps == 0 or not self._builds_graph():
            return None
        from . import autograd
        keys = [(p, i) for p in ("cat", "an") for i in range(self.num_steps)]
        mats = autograd.BondTypeMatricesAll.apply(self.bond_emb.embeddings,
                                                  *[self.branches[p]["bmm"][i].bond_transform for p, i in keys])
        return dict(zip(keys, mats))

    def encode_layered(self, prefix, atom_ids, bond_ids, conn, trace=None, typed=True, type_mats=None):
        """encode() layer at a time (train_viscosity.py:171-187) -> pooled (B,D)."""
        br = self.branches[prefix]
        h = self.atom_emb(atom_ids)
        bond = self.bond_emb(bond_ids)
        if not typed:
            bond = bond.dense()
        one_node = typed and trace is None and self._builds_graph()  # training: a whole step as one autograd node
        # inference through the layered path (what serves atom_dim 64 / 128): GatedUpdate only on the rows an
        # encode() loop has to carry - padding atoms can reach neither a message nor the pool (include/impnn.h,
        # impnn_gated_update_rows); their rows of h are left undefined and are never read
        rows = None
        # (atom_dim 32 has the entry too, but there the three small launches that build the list cost what the skipped
        #  rows save: measured 8.1 vs 8.5 M pairs/s at batch 4096)
        if typed and trace is None and self.atom_dim in (64, 128) and self.num_steps > 0 and (
                not self._builds_graph() or atom_ids.numel() >= TRAIN_ROW_LIST_MIN_ROWS):
            # (training too, from TRAIN_ROW_LIST_MIN_ROWS atom rows per ion: the one-node step runs GatedUpdate forward
            #  AND backward on the list, impnn_gated_update_rows_bwd - padding atoms carry no gradient; below that a
            #  step is bound by its launch count and the list's own launches cost more than the skipped rows save)
            rows = ops.kept_row_index(atom_ids, bond_ids, conn, self.bond_vocab_size)
        for i in range(self.num_steps):
            if one_node:
                from . import autograd
                mats = type_mats[(prefix, i)] if type_mats else br["bmm"][i]._type_matrices(bond.table)
                u, w = br["update"][i], br["update"][i]._weights
                h = autograd.MessagePassingStep.apply(
                    h, bond.ids, conn, mats, w["dense_z/kernel"], w["dense_z/bias"], w["dense_r/kernel"],
                    w["dense_r/bias"], w["dense_h/kernel"], w["dense_h/bias"], u.gamma, u.beta, u.epsilon,
                    *(rows if rows is not None else (None, None)), i > 0)
                continue
            m = br["bmm"][i]([h, bond, conn])
            agg = br["reduce"][i]([m, conn[:, :, 1], h])
            if rows is not None:
                u, w = br["update"][i], br["update"][i]._weights
                h = ops.gated_update(h, agg, w["dense_z/kernel"], w["dense_z/bias"], w["dense_r/kernel"],
                                     w["dense_r/bias"], w["dense_h/kernel"], w["dense_h/bias"], u.gamma, u.beta,
                                     u.epsilon, rows=rows)
            else:
                h = br["update"][i]([h, agg])
            if trace is not None:
                trace[f"{prefix}/m{i}"], trace[f"{prefix}/agg{i}"], trace[f"{prefix}/h{i + 1}"] = m, agg, h
        pooled = br["pool"]([h, atom_ids])
        if trace is not None:
            trace[f"{prefix}/pooled"] = pooled
        return pooled

    def plan_batch(self, inputs):
        """Enqueues the graph-only plan of a batch on a side stream (ops.EncoderPipeline) and returns
        a handle for ``encode_pooled(inputs, plan=handle)``.  Call it for batch i+1 before encoding
        batch i: the plan then runs underneath the encoder of batch i."""
        if getattr(self, "_pipeline", None) is None:
            self._pipeline = ops.EncoderPipeline(self.device)
        ions = [(inputs["cat_atom"], inputs["cat_bond"], inputs["cat_connectivity"]),
                (inputs["an_atom"], inputs["an_bond"], inputs["an_connectivity"])]
        mode = self.resolve_encoder_mode(inputs["cat_atom"].shape[1], inputs["cat_bond"].shape[1])
        if mode is None:
            raise ops.EncoderUnsupported("no fused encoder mode covers this model / batch shape")
        return self._pipeline.plan(ions, self.atom_dim, self.bond_dim, self.num_steps, self.atom_vocab_size,
                                   self.bond_vocab_size, mode=mode, workgroups=self.encoder_workgroups)

    def encode_pooled(self, inputs, fused=None, trace=None, plan=None):
        """Both ions' GlobalSumPool outputs: the hot path."""
        if plan is not None:
            mode = plan.mode
            pc, pa = self._pipeline.run(plan, self.atom_emb.embeddings, self.bond_emb.embeddings,
                                        self._prepared_weights(mode), mode=mode)
            if trace is not None:
                trace["cat/pooled"], trace["an/pooled"] = pc, pa
            return pc, pa
        ca, cb, cc = inputs["cat_atom"], inputs["cat_bond"], inputs["cat_connectivity"]
        aa, ab, ac = inputs["an_atom"], inputs["an_bond"], inputs["an_connectivity"]
        if fused is None:
            fused = (tuple(ca.shape) == tuple(aa.shape) and tuple(cb.shape) == tuple(ab.shape)
                     and self.fused_supported(ca.shape[1], cb.shape[1]))
        if fused:
            mode = self.resolve_encoder_mode(ca.shape[1], cb.shape[1])
            if mode is None:
                raise ops.EncoderUnsupported("no fused encoder mode covers this model / batch shape")
            prepared = self._prepared_weights(mode) if self.num_steps > 0 else None
            try:
                pc, pa = ops.encoder_fused([(ca, cb, cc), (aa, ab, ac)], self.atom_emb.embeddings,
                                           self.bond_emb.embeddings, None if prepared else self._packed_weights(),
                                           self.num_steps, mode=mode, prepared=prepared,
                                           workgroups=self.encoder_workgroups)
            except ops.EncoderOverflow:
                # a molecule of THIS batch exceeds a chunk (possible only for N > 256 or E > 255): layer at a time
                self.overflow_fallbacks = getattr(self, "overflow_fallbacks", 0) + 1
                return self.encode_pooled(inputs, fused=False, trace=trace)
            if trace is not None:
                trace["cat/pooled"], trace["an/pooled"] = pc, pa
            return pc, pa
        tm = self._all_type_matrices() if trace is None else None
        if trace is None and getattr(self, "two_streams", True) and ca.is_cuda \
                and ca.shape[0] <= int(os.environ.get("IMPNN_TWO_STREAM_MAX_BATCH", TWO_STREAM_MAX_BATCH)):
            return self._encode_two_streams((ca, cb, cc), (aa, ab, ac), tm)
        return (self.encode_layered("cat", ca, cb, cc, trace, type_mats=tm),
                self.encode_layered("an", aa, ab, ac, trace, type_mats=tm))

    def _encode_two_streams(self, cat, an, tm):
        """Layer-at-a-time path: the two ions' chains are independent until the head, and most of their kernels
        leave part of the chip idle (a batch-32 kernel nearly all of it) - the anion chain runs on a second HIP stream
        (in training: a parallel branch of the captured hipGraph; autograd replays each node's backward on the stream
        of its forward).  Tensors allocated on
        the current stream and read by the side stream are recorded there, so the caching allocator does not reuse
        them before the side stream is done.  ``join_training_streams`` after backward() orders the gradient sinks
        written on the side stream before the optimizer step."""
        cur = torch.cuda.current_stream(self.device)
        side = getattr(self, "_side_stream", None)
        if side is None:
            side = self._side_stream = torch.cuda.Stream(device=self.device)
            # leaves reached from the side chain through AccumulateGrad (bond_dim >= 64 keeps per-layer type-matrix
            # nodes) see a producer on another stream than their own: intended here, torch syncs the two
            quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if quiet is not None:
                quiet(False)
        for (p, _), mats in (tm or {}).items():
            if p == "an":
                mats.record_stream(side)
                pool = getattr(mats, "_impnn_dmats", None)
                if pool is not None:
                    pool.record_stream(side)
        for t in an:
            t.record_stream(side)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            pa = self.encode_layered("an", *an, None, type_mats=tm)
        pc = self.encode_layered("cat", *cat, None, type_mats=tm)
        cur.wait_stream(side)
        pa.record_stream(cur)
        self._side_stream_used = True
        if pa.requires_grad:
            # whoever calls backward(): when the gradient reaches the anion chain, queue the join for the end of that
            # backward pass (runs on the calling thread, so "current stream" is the caller's)
            def _queue_join(grad):
                torch.autograd.Variable._execution_engine.queue_callback(self.join_training_streams)
                return grad
            pa.register_hook(_queue_join)
        return pc, pa

    def join_training_streams(self):
        """After loss.backward(): the current stream waits for the side stream of _encode_two_streams (its backward
        kernels add into the gradient buffers in place, which torch's own end-of-backward sync does not see)."""
        if getattr(self, "_side_stream_used", False):
            torch.cuda.current_stream(self.device).wait_stream(self._side_stream)
            self._side_stream_used = False

    def head(self, pooled_cat, pooled_an, temperature=None, trace=None, differentiable=False):
        if trace is None and not differentiable and self.atom_dim <= 128 and max(self.fp_size, self.mixing_size) <= 64:
            return ops.model_head(self.kind, pooled_cat, pooled_an, temperature, self._packed_head(), self.fp_size,
                                  self.mixing_size)  # one launch
        if trace is None and differentiable and self.atom_dim <= 128 and max(self.fp_size, self.mixing_size) <= 64 \
                and torch.is_grad_enabled():
            from . import autograd
            T = temperature if self.kind == "viscosity" else None
            return autograd.ModelHead.apply({"viscosity": 0, "melting_point": 1}[self.kind], self.fp_size,
                                            self.mixing_size, pooled_cat, pooled_an, T, *self._head_tensors())
        fp_cat = self.branches["cat"]["fp"](pooled_cat)   # Dense(fp_size, relu), :189
        fp_an = self.branches["an"]["fp"](pooled_an)
        mixed = self.mix([self.cat_proj(fp_cat), self.an_proj(fp_an)])
        if trace is not None:
            trace["cat/fp"], trace["an/fp"], trace["mixed"] = fp_cat, fp_an, mixed
        if self.kind == "viscosity":
            vp = self.visc_params(mixed)
            if trace is not None:
                trace["visc_params"] = vp
            T = self.scale_T(temperature.to(torch.float32).reshape(-1, 1))
            return self.log_eta([self.param_A(vp), self.param_B(vp), T, self.param_C(vp)])
        return self.mp_out(self.mp_hidden(mixed))

    def __call__(self, inputs, fused=None, trace=None, training=False):
        """Inference by default (no graph is kept, the fused encoder and the head kernel run).  With
        ``training=True`` the call is differentiable: layer-at-a-time path (ionic_mpnn_amd.autograd) and the
        head as one autograd node over impnn_model_head_tensors / impnn_model_head_bwd."""
        inputs = self._to_device(inputs)
        if training:
            from . import autograd
            with autograd.training_pass():
                pc, pa = self.encode_pooled(inputs, fused=False)
                return self.head(pc, pa, inputs.get("temperature"), differentiable=True)
        from . import autograd
        with torch.no_grad(), autograd.training_pass():  # (the scope also lets the layers of an ion share graph work)
            pc, pa = self.encode_pooled(inputs, fused=fused, trace=trace)
            return self.head(pc, pa, inputs.get("temperature"), trace=trace)

    # ------------------------------------------------------------------ training
    def trainable_variables(self):
        """[(name, tensor)] in a fixed order; every variable of the reference model is trainable."""
        return list(self._named_tensors().items())

    def compile(self, optimizer=None, loss="mse"):
        """model.compile(optimizer=Adam(1e-3, clipnorm=1.0), loss="mse") (train_viscosity.py:227-230)."""
        from . import train
        if loss != "mse":
            raise ValueError("the reference trainers use loss='mse'")
        self.optimizer = optimizer if optimizer is not None else train.Adam(1e-3, clipnorm=1.0)
        for _, t in self.trainable_variables():
            t.requires_grad_(True)
        self.optimizer.build([t for _, t in self.trainable_variables()])
        return self

    def regularization_loss(self):
        """keras l2(fp_l2) on the fingerprint Dense kernels (train_viscosity.py:189) and, for the melting-point
        model, on the hidden Dense (train_melting_point.py:173,197): fp_l2 * sum(w^2)."""
        ks = [self.branches["cat"]["fp"].kernel, self.branches["an"]["fp"].kernel]
        if self.kind != "viscosity":
            ks.append(self.mp_hidden.kernel)
        return self.fp_l2 * sum((k * k).sum() for k in ks)

    def _head_l2(self):
        """keras l2 lambda per head tensor, in the order of _head_tensors() (see regularization_loss)."""
        lam = [self.fp_l2, 0.0, self.fp_l2, 0.0, 0.0, 0.0, 0.0, 0.0]
        return lam + ([0.0, 0.0] if self.kind == "viscosity" else [self.fp_l2, 0.0, 0.0, 0.0])

    def _loss(self, inputs, y, training):
        from . import train
        y = torch.as_tensor(y, dtype=torch.float32).to(self.device).reshape(-1, 1)
        if training and torch.is_grad_enabled() and y.shape[0] > 0 \
                and self.atom_dim <= 128 and max(self.fp_size, self.mixing_size) <= 64:
            # head, mse and the l2 penalties as ONE node (impnn_model_head_loss): ~25 launches fewer per step
            from . import autograd
            inputs = self._to_device(inputs)
            need = int(ops._lib.load().impnn_model_head_loss_workspace_floats(int(y.shape[0])))
            ws = getattr(self, "_loss_ws", None)
            if ws is None or ws.numel() < need:
                ws = self._loss_ws = torch.zeros(max(need, 1024), dtype=torch.float32, device=self.device)
            with autograd.training_pass():
                pc, pa = self.encode_pooled(inputs, fused=False)
                T = inputs.get("temperature") if self.kind == "viscosity" else None
                return autograd.ModelHeadLoss.apply({"viscosity": 0, "melting_point": 1}[self.kind], self.fp_size,
                                                    self.mixing_size, self._head_l2(), ws, pc, pa, T, y,
                                                    *self._head_tensors())
        pred = self(inputs, training=True) if training else self(inputs)
        if training:
            return train.mse(y, pred) + self.regularization_loss()
        with torch.no_grad():
            return train.mse(y, pred) + self.regularization_loss()

    def train_on_batch(self, inputs, y, group=None, n_global=None):
        """One optimizer step on one mini-batch -> the batch loss (MSE + penalties) as a 0-d tensor.
        With torch.distributed initialised (or ``group`` given) every rank calls this with its shard of
        the global mini-batch: the gradients are averaged over ranks (weighted by shard size) with one
        all-reduce of the flat gradient buffer, then every rank applies the same step.  ``n_global``: size of the
        global mini-batch when the caller knows it (fit does) - saves the blocking all-reduce that counts it."""
        from . import dist as idist
        if getattr(self, "optimizer", None) is None:
            self.compile()
        opt = self.optimizer
        n_local = len(inputs["cat_atom"])
        loss = self._loss(inputs, y, training=True) if n_local else None
        if idist.is_distributed():
            if n_global is not None:
                weight = float(n_local) / float(n_global) if n_global > 0 else 0.0
            else:
                weight, _ = idist.shard_loss_weight(n_local, opt.flat_grad.device, group)
            if loss is not None:
                (loss * weight).backward()
            self.join_training_streams()
            idist.all_reduce_flat_gradients_(opt.flat_grad, group)
        elif loss is not None:
            loss.backward()
            self.join_training_streams()
        opt.apply_gradients()   # clips, updates, and leaves the gradients in place ...
        opt.zero_grad()         # ... so clear them for the next accumulation
        self.invalidate_packed_weights()
        return loss.detach() if loss is not None else torch.zeros((), device=opt.flat_grad.device)

    def evaluate(self, inputs, y, batch_size=32):
        """model.evaluate: sample-weighted mean of the batch losses (MSE + penalties)."""
        n = len(inputs["cat_atom"])
        tot = torch.zeros((), dtype=torch.float64, device=self.device)
        for lo in range(0, n, batch_size):
            sl = slice(lo, min(n, lo + batch_size))
            loss = self._loss({k: v[sl] for k, v in inputs.items()}, y[sl], training=False)
            tot.add_(loss, alpha=sl.stop - sl.start)  # summed on the device: one host sync per evaluate()
        return float(tot) / max(n, 1)

    def fit(self, x, y, validation_data=None, epochs=1, batch_size=32, callbacks=None, shuffle=True, verbose=0,
            seed=None, graph=True):
        """model.fit as the trainers call it (train_viscosity.py:328-338): per epoch a fresh shuffle,
        mini-batches of ``batch_size``, `loss` = sample-weighted mean of the batch losses, `val_loss` from
        evaluate(); callbacks see on_train_begin / on_epoch_end / on_train_end.  Returns a History.
        ``graph=True`` (single process): full-size mini-batches replay one captured hipGraph of the whole step
        (train.GraphedTrainStep); the last, smaller batch of an epoch runs eagerly.
        Under torch.distributed every rank passes the FULL (x, y): the shuffle seed is rank 0's (broadcast once), every
        global mini-batch is cut into contiguous shards (data.shard_bounds), so all ranks run the same number of steps
        with the same sample order, and the reported loss is the global sample-weighted mean."""
        from . import train
        if getattr(self, "optimizer", None) is None:
            self.compile()
        x = self._to_device(x)
        y = np.asarray(y, dtype=np.float32)
        n = len(y)
        if n:  # once per data set (one device sync), not per batch: raise where tf-CPU's gather / scatter_nd would
            for pfx in ("cat", "an"):
                ops.validate_indices(conn=x[f"{pfx}_connectivity"], atom_ids=x[f"{pfx}_atom"], bond_ids=x[f"{pfx}_bond"],
                                     N=x[f"{pfx}_atom"].shape[1], Va=self.atom_vocab_size, Vb=self.bond_vocab_size)
        from . import dist as idist
        distributed = idist.is_distributed()
        if distributed:  # one shared permutation stream: rank 0's seed (drawn if none was given)
            import torch.distributed as tdist
            sd = torch.tensor([int(seed) if seed is not None else int(np.random.SeedSequence().entropy % (1 << 62))],
                              dtype=torch.int64, device=self.device if tdist.get_backend() == "nccl" else "cpu")
            tdist.broadcast(sd, src=0)
            seed = int(sd.item())
            rank, world = tdist.get_rank(), tdist.get_world_size()
        rng = np.random.default_rng(seed)
        hist = train.History()
        callbacks = list(callbacks or [])
        self.stop_training = False
        for cb in callbacks:
            if hasattr(cb, "set_model"):
                cb.set_model(self)
            if hasattr(cb, "on_train_begin"):
                cb.on_train_begin({})
        graphed = None
        use_graph = bool(graph) and not distributed and n >= batch_size
        y_dev = torch.from_numpy(y).to(self.device).reshape(-1, 1)
        val_dev = None
        for epoch in range(int(epochs)):
            order = rng.permutation(n) if shuffle else np.arange(n)
            order_dev = torch.from_numpy(order).to(self.device)  # one upload per epoch
            tot = torch.zeros((), dtype=torch.float64, device=self.device)
            for lo in range(0, n, batch_size):
                idx = order[lo:lo + batch_size]
                tidx = order_dev[lo:lo + batch_size]
                if use_graph and len(idx) == batch_size:
                    if graphed is None:
                        x = {k: v.contiguous() for k, v in x.items()}
                        graphed = train.GraphedTrainStep(self, {k: v[tidx] for k, v in x.items()}, y[idx],
                                                         resident=(x, y_dev))
                    loss = graphed.step_on_rows(x, y_dev, tidx)
                elif distributed:
                    from .data import shard_bounds
                    s0, s1 = shard_bounds(len(idx), world, rank)
                    sidx = tidx[s0:s1]
                    loss = self.train_on_batch({k: v[sidx] for k, v in x.items()}, y_dev[sidx], n_global=len(idx))
                    tot.add_(loss, alpha=s1 - s0)  # shard means, weighted by shard size; summed over ranks below
                    continue
                else:
                    loss = self.train_on_batch({k: v[tidx] for k, v in x.items()}, y_dev[tidx])
                tot.add_(loss, alpha=len(idx))
            if distributed:  # every rank summed its shards' losses: one all-reduce per epoch for the logged mean
                if tdist.get_backend() == "nccl":
                    tdist.all_reduce(tot)
                else:
                    t_cpu = tot.cpu()
                    tdist.all_reduce(t_cpu)
                    tot = t_cpu.to(self.device)
            logs = {"loss": float(tot) / max(n, 1)}
            if validation_data is not None:
                if val_dev is None:  # uploaded once, not per epoch
                    val_dev = (self._to_device(validation_data[0]),
                               torch.from_numpy(np.asarray(validation_data[1], np.float32)).to(self.device))
                # the sample-weighted mean does not depend on how the set is cut: large forward-only batches
                logs["val_loss"] = self.evaluate(val_dev[0], val_dev[1], max(batch_size, 4096))
            hist._log(epoch, logs)
            if verbose:
                print(f"Epoch {epoch + 1}/{epochs} - " + " - ".join(f"{k}: {v:.4f}" for k, v in logs.items()), flush=True)
            for cb in callbacks:
                if hasattr(cb, "on_epoch_end"):
                    cb.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        for cb in callbacks:
            if hasattr(cb, "on_train_end"):
                cb.on_train_end({})
        self.history = hist
        return hist

    def predict(self, inputs, batch_size=None, fused=None):
        """model.predict(x) (train_viscosity.py:366): returns a numpy (n,1) array.  The reference's
        Keras default batch is 32; here the whole set is one batch unless batch_size is given."""
        n = len(inputs["cat_atom"])
        bs = n if not batch_size else int(batch_size)
        if n == 0:
            return np.zeros((0, 1), np.float32)
        if n <= bs:
            return self(inputs, fused=fused).detach().cpu().numpy()
        # several chunks: consecutive chunks on two HIP streams (their kernels overlap, see bench.py --streams), one
        # host synchronisation at the end instead of one per chunk
        cur = torch.cuda.current_stream(self.device)
        lanes = getattr(self, "_predict_lanes", None)
        if lanes is None:
            lanes = self._predict_lanes = [torch.cuda.Stream(device=self.device) for _ in range(2)]
        for ln in lanes:
            ln.wait_stream(cur)
        outs = []
        for i, lo in enumerate(range(0, n, bs)):
            chunk = {k: v[lo:lo + bs] for k, v in inputs.items()}
            with torch.cuda.stream(lanes[i % 2]):
                outs.append(self(chunk, fused=fused).detach())
        for ln in lanes:
            cur.wait_stream(ln)
        for o in outs:
            o.record_stream(cur)
        return torch.cat(outs, dim=0).cpu().numpy()

    def _to_device(self, inputs):
        out = {}
        for k, v in inputs.items():
            if isinstance(v, np.ndarray):
                v = torch.from_numpy(v)
            if not isinstance(v, torch.Tensor):
                raise TypeError(f"input {k!r} must be a numpy array or torch tensor")
            out[k] = v.to(self.device, non_blocking=True)
        if self.kind == "viscosity" and "temperature" not in out:
            raise KeyError("the viscosity model needs a 'temperature' input (train_viscosity.py:160)")
        return out


def build_model(atom_vocab_size, bond_vocab_size, atom_dim=32, bond_dim=8, fp_size=32, mixing_size=20,
                num_steps=4, device=None):
    """train_viscosity.py:139-231 (same positional/keyword signature and defaults)."""
    return MPNNModel("viscosity", atom_vocab_size, bond_vocab_size, atom_dim, bond_dim, fp_size, mixing_size,
                     num_steps, fp_l2=1e-4, device=device)


def build_melting_point_model(atom_vocab_size, bond_vocab_size, atom_dim=32, fp_size=32, mixing_size=20,
                              num_steps=4, device=None):
    """train_melting_point.py:137-215: bond embedding width = atom_dim**2 (:146)."""
    return MPNNModel("melting_point", atom_vocab_size, bond_vocab_size, atom_dim, atom_dim * atom_dim, fp_size,
                     mixing_size, num_steps, fp_l2=1e-5, device=device)


def load_model(path, custom_objects=None, device=None):
    """keras.models.load_model analogue for files written by MPNNModel.save / save_weights
    (train_melting_point_transfer.py:78-93 passes custom_objects: accepted and ignored - the layer classes are
    this package's own)."""
    cfg, w = MPNNModel.load_weight_file(path)
    m = MPNNModel.from_config(cfg, device=device)
    m.load_weights(w)
    return m
