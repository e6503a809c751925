"""model.compile / fit / evaluate as the reference's trainers use them:
Adam(1e-3, clipnorm=1.0) + MSE (+ the l2(1e-4) penalty of the fingerprint Dense kernels), mini-batches of
32 in a fresh shuffle per epoch, EarlyStopping(monitor="val_loss", patience=50, restore_best_weights=True)
(train_viscosity.py:189,227-230,328-338; train_melting_point.py:173,205-208,300-311).

Forward and backward of the message-passing layers run in libimpnn (ionic_mpnn_amd.autograd); the
optimizer step of all variables is one launch (impnn_adam_clipnorm_step); the head after GlobalSumPool, the
mse and the l2 penalties are one node too (impnn_model_head_loss[_bwd]) - torch.autograd only keeps the graph, no
torch arithmetic kernel runs in a step.  Multi-GPU: one process per GPU, every rank takes
its contiguous shard of each mini-batch, and the flat gradient buffer is averaged with ONE all-reduce
per step (RCCL; gloo in the CPU tests) before the replicated optimizer step."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, stream_ptr


class Adam:
    """keras.optimizers.Adam(learning_rate, clipnorm): per-variable tf.clip_by_norm, then Adam with
    epsilon 1e-7 and the bias-corrected step size lr*sqrt(1-b2^t)/(1-b1^t)."""

    def __init__(self, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, clipnorm=None):
        self.learning_rate, self.beta_1, self.beta_2 = float(learning_rate), float(beta_1), float(beta_2)
        self.epsilon, self.clipnorm = float(epsilon), (None if clipnorm is None else float(clipnorm))
        self._vars = None
        self._step_dev = None

    @property
    def iterations(self):
        """Number of updates applied (the counter lives in device memory; reading it synchronises)."""
        return int(self._step_dev.item()) if self._step_dev is not None else 0

    def get_config(self):
        return {"name": "Adam", "learning_rate": self.learning_rate, "beta_1": self.beta_1, "beta_2": self.beta_2,
                "epsilon": self.epsilon, "clipnorm": self.clipnorm}

    def build(self, variables):
        """variables: list of leaf tensors on one GPU.  Gradients live in ONE flat buffer (views become the
        variables' .grad), so the data-parallel average is a single collective and the kernel's pointer table
        stays valid for the whole run (also inside a captured hipGraph)."""
        variables = list(variables)
        dev = variables[0].device
        sizes = [int(v.numel()) for v in variables]
        total = sum(sizes)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(total, dtype=torch.float32, device=dev)
        self._step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        table, off = [], 0
        for var, n in zip(variables, sizes):
            if var.dtype != torch.float32 or not var.is_contiguous():
                raise ValueError("optimizer variables must be contiguous float32 tensors")
            var.grad = self.flat_grad[off:off + n].view_as(var)
            table += [var.data_ptr(), self.flat_grad.data_ptr() + 4 * off, self.m.data_ptr() + 4 * off,
                      self.v.data_ptr() + 4 * off]
            off += n
        # data_ptr values are < 2^63: store them as int64 bit patterns
        self._table = torch.tensor(table, dtype=torch.int64, device=dev)
        self._sizes = torch.tensor(sizes, dtype=torch.int64, device=dev)
        self._vars = variables

    def zero_grad(self):
        self.flat_grad.zero_()

    def state(self):
        return {"m": self.m.clone(), "v": self.v.clone(), "step": self._step_dev.clone()}

    def load_state(self, st):
        self.m.copy_(st["m"]); self.v.copy_(st["v"]); self._step_dev.copy_(st["step"])

    def apply_gradients(self):
        """One launch: clip, moments, update for every variable (gradients are read from the flat buffer);
        the step counter is advanced on the device, so the call can sit inside a captured graph."""
        if self._vars is None:
            raise RuntimeError("Adam.build(variables) has not been called")
        lo, hi = self.flat_grad.data_ptr(), self.flat_grad.data_ptr() + 4 * self.flat_grad.numel()
        for var in self._vars:  # autograd swaps .grad for a fresh tensor if it was reset to None
            if var.grad is None or not (lo <= var.grad.data_ptr() < hi):
                raise RuntimeError("a variable's .grad no longer points into the optimizer's flat buffer; "
                                   "use optimizer.zero_grad() instead of setting .grad = None")
        dev = self.flat_grad.device
        with torch.cuda.device(dev):
            check(_lib.load().impnn_adam_clipnorm_step_counted(
                C.c_void_p(self._table.data_ptr()), C.c_void_p(self._sizes.data_ptr()), len(self._vars),
                C.c_void_p(self._step_dev.data_ptr()), self.learning_rate, self.beta_1, self.beta_2, self.epsilon,
                self.clipnorm if self.clipnorm else 0.0, stream_ptr()))


class GraphedTrainStep:
    """One training step (forward, backward, optimizer) of a fixed batch shape captured in a hipGraph
    (torch.cuda.CUDAGraph) and replayed per mini-batch: the reference trains with batches of 32
    (train_viscosity.py:332), where a step is ~150 small launches and the host, not the GPU, sets the pace.
    Inputs are copied into static buffers; every kernel argument that changes between steps lives in device
    memory (the Adam step counter)."""

    def __init__(self, model, inputs, y, resident=None):
        """``resident=(x, y_dev)``: the whole padded training set lives on the device; the captured step then starts
        with the gather of the rows in ``self.static_rows`` (impnn_gather_rows) and ``step_on_rows`` only refreshes
        those 8*batch bytes between replays."""
        self.model = model
        dev = model.device
        self.static_in = {k: v.clone() for k, v in model._to_device(inputs).items()}
        self.static_y = torch.as_tensor(np.asarray(y, np.float32)).to(dev).reshape(-1, 1).clone()
        self.batch = int(self.static_y.shape[0])
        self.resident = None
        if resident is not None:
            x, y_dev = resident
            keys = list(self.static_in)
            ok = len(keys) + 1 <= 8 and all(
                x[k].is_contiguous() and x[k].dtype == self.static_in[k].dtype and x[k].element_size() == 4
                for k in keys) and y_dev.is_contiguous() and y_dev.dtype == torch.float32
            if ok:
                self.resident = ([x[k] for k in keys] + [y_dev], [self.static_in[k] for k in keys] + [self.static_y])
                self.static_rows = torch.zeros(self.batch, dtype=torch.int64, device=dev)
        opt = model.optimizer
        saved_w = [t.detach().clone() for _, t in model.trainable_variables()]
        saved_o = opt.state()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # warm-up off the default stream, as graph capture requires
            for _ in range(3):
                self._step()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_loss = self._step()
        with torch.no_grad():  # the warm-up steps were real updates: undo them
            for (_, t), w0 in zip(model.trainable_variables(), saved_w):
                t.copy_(w0)
        opt.load_state(saved_o)
        opt.zero_grad()
        model.invalidate_packed_weights()

    def _step(self):
        m = self.model
        if self.resident is not None:
            from . import ops
            ops.gather_rows(self.resident[0], self.resident[1], self.static_rows)
        loss = m._loss(self.static_in, self.static_y, training=True)
        loss.backward()
        m.join_training_streams()
        m.optimizer.apply_gradients()
        m.optimizer.zero_grad()
        return loss.detach()

    def matches(self, inputs):
        return all(tuple(inputs[k].shape) == tuple(v.shape) for k, v in self.static_in.items())

    def step_on_rows(self, x, y_dev, rows):
        """One step on the mini-batch ``rows`` (device int64 tensor, len = batch) of a device-resident data set:
        the rows are gathered straight into the static buffers (one index_select per tensor, no staging copy, no
        host round trip)."""
        if self.resident is not None and x[next(iter(self.static_in))] is self.resident[0][0]:
            self.static_rows.copy_(rows)  # the gather is the first node of the graph
        else:
            for k, v in self.static_in.items():
                torch.index_select(x[k], 0, rows, out=v)
            torch.index_select(y_dev, 0, rows, out=self.static_y)
        self.graph.replay()
        self.model.invalidate_packed_weights()
        return self.static_loss

    def __call__(self, inputs, y):
        for k, v in self.static_in.items():
            v.copy_(inputs[k], non_blocking=True)
        self.static_y.copy_(torch.as_tensor(np.asarray(y, np.float32)).reshape(-1, 1), non_blocking=True)
        self.graph.replay()
        self.model.invalidate_packed_weights()
        return self.static_loss


class History:
    def __init__(self):
        self.history = {}
        self.epoch = []

    def _log(self, epoch, logs):
        self.epoch.append(epoch)
        for k, v in logs.items():
            self.history.setdefault(k, []).append(float(v))


class Callback:
    """keras.callbacks.Callback protocol as the trainers use it (train_viscosity.py:112-132 subclasses it):
    ``set_model`` is called once, then ``on_train_begin(logs)``, ``on_epoch_end(epoch, logs)`` per epoch and
    ``on_train_end(logs)``; a callback stops training by setting ``self.model.stop_training = True``."""

    def __init__(self):
        self.model = None

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass


class EarlyStopping(Callback):
    """keras.callbacks.EarlyStopping(monitor, patience, restore_best_weights) with mode "min"
    (train_viscosity.py:334)."""

    def __init__(self, monitor="val_loss", patience=0, restore_best_weights=False, min_delta=0.0):
        super().__init__()
        self.monitor, self.patience, self.restore_best_weights = monitor, int(patience), bool(restore_best_weights)
        self.min_delta = abs(float(min_delta))
        self.best, self.wait, self.best_weights, self.stopped_epoch, self.best_epoch = np.inf, 0, None, 0, 0

    def on_train_begin(self, logs=None):
        self.best, self.wait, self.best_weights, self.stopped_epoch, self.best_epoch = np.inf, 0, None, 0, 0

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        if self.restore_best_weights and self.best_weights is None:
            self.best_weights = self.model.state_dict()
        self.wait += 1
        if cur < self.best - self.min_delta:
            self.best, self.best_epoch, self.wait = cur, epoch, 0
            if self.restore_best_weights:
                self.best_weights = self.model.state_dict()
            return
        if self.wait >= self.patience and epoch > 0:
            self.stopped_epoch = epoch
            self.model.stop_training = True

    def on_train_end(self, logs=None):
        if self.restore_best_weights and self.best_weights is not None:
            self.model.load_weights(self.best_weights)


def mse(y_true, y_pred):
    """keras "mse": mean over the last axis, then over the batch."""
    return torch.mean((y_pred.reshape(y_true.shape[0], -1) - y_true.reshape(y_true.shape[0], -1)) ** 2)


def slice_inputs(inputs, idx):
    return {k: v[idx] for k, v in inputs.items()}
