"""CPU restatement of the reference's optimizer step: keras.optimizers.Adam(lr, clipnorm) as configured at
train_viscosity.py:227-230.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py); "parity unpinned" (Keras is
not importable here; the update rule follows the published Keras/TF 2.12 Adam: tf.clip_by_norm per variable,
m/v moments, step size lr*sqrt(1-b2^t)/(1-b1^t), epsilon 1e-7 outside the square root)."""
from __future__ import annotations

import numpy as np


def clip_by_norm(g, clipnorm):
    """tf.clip_by_norm: g * clipnorm / max(||g||_2, clipnorm)."""
    if clipnorm is None or clipnorm <= 0:
        return g
    n = np.sqrt(np.sum(np.asarray(g, np.float64) ** 2))
    return g * (clipnorm / max(n, clipnorm))


def adam_step(w, g, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7, clipnorm=None):
    """One update of one variable; t counts from 1.  Returns (w, m, v) in float64."""
    w, g, m, v = (np.asarray(a, np.float64) for a in (w, g, m, v))
    g = clip_by_norm(g, clipnorm)
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * g * g
    alpha = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    return w - alpha * m / (np.sqrt(v) + eps), m, v
