"""Probability helpers with the reference's names (kgvae/utils.py:323-428).

The hot uses -- reparameterisation of the (N, 2h) encoder output and the KL to the mixture
prior -- go through the fused HIP kernels (``ops.reparam``, ``ops.kl_to_mixture``).  The small
generic forms below (mixture parameters of ``z_pre``: k x h values) are device-side tensor glue.
"""
import math

import torch
import torch.nn.functional as F


def gaussian_parameters(h, dim=-1):
    m, raw = torch.split(h, h.size(dim) // 2, dim=dim)
    return m, F.softplus(raw) + 1e-8


def sample_gaussian(m, v, repeat=1, eps=None):
    if repeat > 1:
        m, v = m.squeeze(), v.squeeze()
        sd = torch.cat([torch.sqrt(v)] * repeat, dim=0)
        m = torch.cat([m] * repeat, dim=0)
    else:
        sd = torch.sqrt(v)
    if eps is None:
        eps = torch.randn_like(sd)
    return m + eps * sd


def log_normal(x, m, v):
    return torch.sum(-(x - m).pow(2) / (2 * v) - v.sqrt().log() - math.log(math.sqrt(2 * math.pi)), dim=-1)


def log_sum_exp(x, dim=0):
    mx = torch.max(x, dim)[0]
    return mx + (x - mx.unsqueeze(dim).expand_as(x)).exp().sum(dim).log()


def log_mean_exp(x, dim):
    return log_sum_exp(x, dim) - math.log(x.size(dim))


def log_normal_mixture(z, m, v):
    return log_mean_exp(log_normal(z.unsqueeze(1), m, v), dim=-1)
