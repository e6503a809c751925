"""Keras-layer surface of the reference's custom layers, on torch tensors + libimpnn (HIP).

Mirrors models/layers.py of the reference: same class names, constructor arguments, input-list
order, weight names/shapes, ``build()/call()/get_config()`` protocol and Keras-style auto names
(``gated_update``, ``gated_update_1`` ... - the transfer script addresses layers by those names,
train_melting_point_transfer.py:214-220).  Host tensors are ``torch.Tensor`` on an MI355X
(int32 ids/connectivity, float32 state) instead of ``tf.Tensor``; there is no CPU execution path.
"""
from __future__ import annotations

import collections
import re

import numpy as np
import torch

from . import ops
from .weights import embedding_uniform, glorot_uniform

_name_uids = collections.defaultdict(int)
_registry = {}
_init_rng = np.random.default_rng(0)


def reset_uids():
    """keras.backend.clear_session() analogue for auto-generated layer names."""
    _name_uids.clear()


def set_init_seed(seed):
    global _init_rng
    _init_rng = np.random.default_rng(seed)


def _to_snake_case(name):
    s = re.sub(r"(.)([A-Z][a-z]+)", r"\1_\2", name)
    return re.sub(r"([a-z0-9])([A-Z])", r"\1_\2", s).lower()


def register_keras_serializable(package="Custom"):
    """@register_keras_serializable() of models/layers.py:3,10,... : records the class so that
    ``custom_objects``-style lookup by name works (train_melting_point_transfer.py:78-93)."""
    def deco(cls):
        _registry[f"{package}>{cls.__name__}"] = cls
        _registry[cls.__name__] = cls
        return cls
    return deco


def get_registered(name):
    return _registry[name]


def default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("ionic_mpnn_amd needs an MI355X (torch.cuda is unavailable); no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


class Layer:
    """The slice of keras.layers.Layer the reference uses."""

    def __init__(self, name=None, trainable=True, dtype="float32", device=None, **kwargs):
        if kwargs:
            raise TypeError(f"unexpected keyword arguments {sorted(kwargs)}")
        if name is None:
            base = _to_snake_case(type(self).__name__)
            uid = _name_uids[base]
            _name_uids[base] += 1
            name = base if uid == 0 else f"{base}_{uid}"
        self.name = name
        self.trainable = trainable
        self.dtype = dtype
        self.built = False
        self._device = device
        self._weights = collections.OrderedDict()

    # -- keras protocol
    def build(self, input_shape):
        pass

    def call(self, inputs, **kwargs):
        raise NotImplementedError

    def __call__(self, inputs, **kwargs):
        if not self.built:
            self.build(_shapes_of(inputs))
            self.built = True
        return self.call(inputs, **kwargs)

    def get_config(self):
        return {"name": self.name, "trainable": self.trainable, "dtype": self.dtype}

    @classmethod
    def from_config(cls, config):
        return cls(**config)

    # -- variables
    @property
    def device(self):
        if self._device is None:
            self._device = default_device()
        return self._device

    def add_weight(self, shape, initializer="glorot_uniform", name=None, trainable=True):
        shape = tuple(int(s) for s in shape)
        if callable(initializer):
            arr = initializer(_init_rng, shape)
        elif initializer == "glorot_uniform":
            arr = glorot_uniform(_init_rng, shape)
        elif initializer == "uniform":
            arr = embedding_uniform(_init_rng, shape)
        elif initializer == "zeros":
            arr = np.zeros(shape, np.float32)
        elif initializer == "ones":
            arr = np.ones(shape, np.float32)
        else:
            raise ValueError(f"unknown initializer {initializer!r}")
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(self.device)
        self._weights[name] = t
        return t

    @property
    def weights(self):
        return list(self._weights.values())

    @property
    def trainable_weights(self):
        return self.weights if self.trainable else []

    def weight_names(self):
        return list(self._weights.keys())

    def get_weights(self):
        return [w.detach().cpu().numpy() for w in self.weights]

    def set_weights(self, arrays):
        names = self.weight_names()
        if len(arrays) != len(names):
            raise ValueError(f"layer {self.name} expects {len(names)} weights, got {len(arrays)}")
        for n, a in zip(names, arrays):
            a = np.asarray(a, dtype=np.float32)
            if tuple(a.shape) != tuple(self._weights[n].shape):
                raise ValueError(f"{self.name}/{n}: shape {a.shape} != {tuple(self._weights[n].shape)}")
            self._weights[n].copy_(torch.from_numpy(np.ascontiguousarray(a)))


def _shapes_of(inputs):
    if isinstance(inputs, (list, tuple)):
        return [tuple(getattr(t, "shape", ())) for t in inputs]
    return tuple(getattr(inputs, "shape", ()))


class EmbeddedLookup:
    """An Embedding output that still knows its ids and table.  BondMatrixMessage uses it to run
    the per-bond-type schedule (SURVEY.md 7 schedule A) instead of reading a (B,E,K) tensor -
    mandatory when K = D*D (train_melting_point.py:146).  ``dense()`` materialises the tensor the
    reference would have produced."""

    def __init__(self, ids, table):
        self.ids, self.table = ids, table
        self._dense = None

    @property
    def shape(self):
        return (*self.ids.shape, self.table.shape[1])

    def dense(self):
        if self._dense is None:
            self._dense = ops.embed_gather(self.ids, self.table)
        return self._dense


class Embedding(Layer):
    """keras.layers.Embedding(input_dim, output_dim, mask_zero=False) as used at
    train_viscosity.py:163-164.  ``lazy=True`` returns an EmbeddedLookup handle."""

    def __init__(self, input_dim, output_dim, mask_zero=False, lazy=False, **kwargs):
        super().__init__(**kwargs)
        if mask_zero:
            raise ValueError("the reference uses mask_zero=False; masking is not implemented")
        self.input_dim, self.output_dim, self.mask_zero, self.lazy = int(input_dim), int(output_dim), False, lazy

    def build(self, input_shape):
        self.embeddings = self.add_weight((self.input_dim, self.output_dim), "uniform", name="embeddings")

    def call(self, ids):
        if self.lazy:
            return EmbeddedLookup(ids, self.embeddings)
        return ops.embed_gather(ids, self.embeddings)

    def get_config(self):
        cfg = super().get_config()
        cfg.update({"input_dim": self.input_dim, "output_dim": self.output_dim, "mask_zero": False})
        return cfg


@register_keras_serializable()
class Reduce(Layer):
    """models/layers.py:52-83: aggregate messages to target atoms, ignoring tgt_idx == 0."""

    def call(self, inputs):
        messages, tgt_idx, atom_ref = inputs
        return ops.reduce_scatter_add(messages, tgt_idx, int(atom_ref.shape[1]))


@register_keras_serializable()
class BondMatrixMessage(Layer):
    """models/layers.py:86-125.  ``call([atom_state, bond_state, connectivity]) -> (B,E,D)``.

    ``fused=True`` gives the orphan models/bond_matrix_message.py:12-19 contract instead:
    the output is the aggregated (B,N,D) tensor (message + Reduce in one launch)."""

    def __init__(self, atom_dim, bond_dim, fused=False, **kwargs):
        super().__init__(**kwargs)
        self.atom_dim = int(atom_dim)
        self.bond_dim = int(bond_dim)
        self.fused = bool(fused)

    def build(self, input_shape):
        if getattr(self, "bond_transform", None) is None:
            self.bond_transform = self.add_weight(
                shape=(self.bond_dim, self.atom_dim, self.atom_dim),
                initializer="glorot_uniform", name="bond_transform")

    def call(self, inputs):
        atom_state, bond_state, connectivity = inputs
        if atom_state.shape[-1] != self.atom_dim:
            raise ValueError(f"atom_state last dim {atom_state.shape[-1]} != atom_dim {self.atom_dim}")
        if isinstance(bond_state, EmbeddedLookup):
            if bond_state.table.shape[1] != self.bond_dim:
                raise ValueError("bond embedding width != bond_dim")
            if self.fused:
                bond_state = bond_state.dense()
            else:
                return ops.bmm_message_typed(atom_state, bond_state.ids, connectivity,
                                             self._type_matrices(bond_state.table))
        if bond_state.shape[-1] != self.bond_dim:
            raise ValueError(f"bond_state last dim {bond_state.shape[-1]} != bond_dim {self.bond_dim}")
        if self.fused:
            return ops.bmm_fused(atom_state, bond_state, connectivity, self.bond_transform)
        return ops.bmm_message(atom_state, bond_state, connectivity, self.bond_transform)

    def _type_matrices(self, table):
        """A[v] = sum_k table[v,k] W[k] depends on the weights only: without grad it is kept until either
        tensor changes (torch version counters; ``invalidate_cache()`` for writers that bypass torch, i.e. the
        optimizer kernel).  With bond_dim = atom_dim**2 this GEMM is most of a layer-at-a-time step."""
        W = self.bond_transform
        if torch.is_grad_enabled() and (W.requires_grad or table.requires_grad):
            return ops.bond_type_matrices(table, W)
        key = (table.data_ptr(), table._version, W.data_ptr(), W._version)
        c = getattr(self, "_mats_cache", None)
        if c is None or c[0] != key:
            c = (key, ops.bond_type_matrices(table, W))
            self._mats_cache = c
        return c[1]

    def invalidate_cache(self):
        self._mats_cache = None

    def get_config(self):
        cfg = super().get_config()
        cfg.update({"atom_dim": self.atom_dim, "bond_dim": self.bond_dim})
        return cfg


class _DenseVars:
    """Holder that mimics the keras Dense sub-layer's .kernel/.bias (models/layers.py:136-138)."""

    def __init__(self, kernel, bias):
        self.kernel, self.bias = kernel, bias


@register_keras_serializable()
class GatedUpdate(Layer):
    """models/layers.py:128-156: GRU-style gate + LayerNormalization + residual + Dropout.

    The reference has no get_config() override here, so its atom_dim is lost on save
    (SURVEY.md 8b); this class serialises it."""

    def __init__(self, atom_dim, dropout_rate=0.0, **kwargs):
        super().__init__(**kwargs)
        self.atom_dim = int(atom_dim)
        self.dropout_rate = float(dropout_rate)
        self.epsilon = ops.LN_EPS

    def build(self, input_shape):
        D = self.atom_dim
        for g in ("z", "r", "h"):
            k = self.add_weight((2 * D, D), "glorot_uniform", name=f"dense_{g}/kernel")
            b = self.add_weight((D,), "zeros", name=f"dense_{g}/bias")
            setattr(self, f"dense_{g}", _DenseVars(k, b))
        self.gamma = self.add_weight((D,), "ones", name="layernorm/gamma")
        self.beta = self.add_weight((D,), "zeros", name="layernorm/beta")

    def call(self, inputs, training=None):
        atom_state, agg = inputs
        if training and self.dropout_rate > 0.0:
            raise NotImplementedError("Dropout with rate > 0 in training mode is outside the forward path; "
                                      "the reference always builds GatedUpdate with rate 0.0 "
                                      "(train_viscosity.py:184)")
        w = self._weights
        return ops.gated_update(atom_state, agg, w["dense_z/kernel"], w["dense_z/bias"], w["dense_r/kernel"],
                                w["dense_r/bias"], w["dense_h/kernel"], w["dense_h/bias"], self.gamma, self.beta,
                                self.epsilon)

    def get_config(self):
        cfg = super().get_config()
        cfg.update({"atom_dim": self.atom_dim, "dropout_rate": self.dropout_rate})
        return cfg


GRUUpdate = GatedUpdate  # README.md:38,199 call it GRUUpdate; the class in the code is GatedUpdate


@register_keras_serializable()
class GlobalSumPool(Layer):
    """models/layers.py:159-164."""

    def call(self, inputs):
        atom_features, atom_ids = inputs
        return ops.global_sum_pool(atom_features, atom_ids)


class Dense(Layer):
    """keras.layers.Dense for the tiny head layers (train_viscosity.py:189,197-198,204): torch addmm;
    not a custom kernel."""

    def __init__(self, units, activation=None, kernel_regularizer=None, **kwargs):
        super().__init__(**kwargs)
        self.units, self.activation, self.kernel_regularizer = int(units), activation, kernel_regularizer
        if activation not in (None, "relu"):
            raise ValueError("only None / 'relu' appear in the reference heads")

    def build(self, input_shape):
        self.kernel = self.add_weight((int(input_shape[-1]), self.units), "glorot_uniform", name="kernel")
        self.bias = self.add_weight((self.units,), "zeros", name="bias")

    def call(self, x):
        y = torch.addmm(self.bias, x, self.kernel)
        return torch.relu(y) if self.activation == "relu" else y

    def get_config(self):
        cfg = super().get_config()
        cfg.update({"units": self.units, "activation": self.activation})
        return cfg


# ---- viscosity head helpers (models/layers.py:10-49); elementwise torch, ~1e2 flop/sample
@register_keras_serializable()
class ComputeLogEta(Layer):
    def call(self, inputs):
        A, B, T, C = inputs
        return A + B / (T + C + 1e-6)


@register_keras_serializable()
class ScaleTemperature(Layer):
    def call(self, t):
        return t / 100.0


@register_keras_serializable()
class SliceParamA(Layer):
    def call(self, x):
        return x[:, 0:1]


def _softplus(x):
    """tf.nn.softplus: log(1 + exp(x)) in the overflow-free form max(x,0) + log1p(exp(-|x|)) - same value,
    and a finite gradient (sigmoid) for |x| > 88, where log1p(exp(x)) differentiates to inf/inf."""
    return torch.clamp(x, min=0.0) + torch.log1p(torch.exp(-torch.abs(x)))


@register_keras_serializable()
class SliceParamB(Layer):
    def call(self, x):
        return torch.clamp(_softplus(x[:, 1:2]), 0.0, 20.0)


@register_keras_serializable()
class SliceParamC(Layer):
    def call(self, x):
        return torch.clamp(_softplus(x[:, 2:3]), 0.1, 50.0)


@register_keras_serializable()
class AddTwoTensors(Layer):
    def call(self, inputs):
        a, b = inputs
        return a + b
