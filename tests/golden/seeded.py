"""Deterministic tensors from numpy's frozen legacy RandomState (stable across numpy/torch versions).
Shared by the fixture generator and the tests so that large weights need not be committed."""
import numpy as np
import torch


def randn(seed, *shape, scale=1.0):
    return torch.from_numpy((np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32))


def made_layers(seed, input_size, hidden_size, n_hidden, scale=None):
    """[(weight, bias)] for MaskedLinear(D,H), n_hidden x (H,H), (H,2D)."""
    dims = [(hidden_size, input_size)] + [(hidden_size, hidden_size)] * n_hidden + [(2 * input_size, hidden_size)]
    out = []
    for i, (o, k) in enumerate(dims):
        s = scale if scale is not None else 1.0 / np.sqrt(k)
        out.append((randn(seed + 2 * i, o, k, scale=s), randn(seed + 2 * i + 1, o, scale=0.1)))
    return out
