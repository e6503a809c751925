"""ionic_mpnn_amd: MI355X-native (gfx950 HIP) message-passing forward path of goalheart/ionic-mpnn.

Python host code on PyTorch-ROCm (device memory, streams, torch.distributed) over libimpnn.so, a
C-ABI library of hand-written HIP kernels (include/impnn.h).  Importing the package does not need
a GPU; every compute call does and raises otherwise - there is no CPU path.
"""
from . import data, synthetic, weights  # noqa: F401  (host-only modules)
from .layers import (AddTwoTensors, BondMatrixMessage, ComputeLogEta, Dense, EmbeddedLookup, Embedding,  # noqa: F401
                     GatedUpdate, GlobalSumPool, GRUUpdate, Layer, Reduce, ScaleTemperature, SliceParamA,
                     SliceParamB, SliceParamC, register_keras_serializable, reset_uids)
from .model import MPNNModel, build_melting_point_model, build_model, load_model  # noqa: F401

__all__ = [
    "BondMatrixMessage", "Reduce", "GatedUpdate", "GRUUpdate", "GlobalSumPool", "Embedding", "Dense",
    "AddTwoTensors", "SliceParamA", "SliceParamB", "SliceParamC", "ScaleTemperature", "ComputeLogEta",
    "build_model", "build_melting_point_model", "MPNNModel", "load_model",
]
