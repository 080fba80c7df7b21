"""Oracle restatement of the probability helpers  --  TEST INFRASTRUCTURE (pinned by golden vectors).

Follows /root/reference/kgvae/utils.py:
  gaussian_parameters :323-339   split in two halves along ``dim``; v = softplus(.) + 1e-8
  sample_gaussian     :342-361   m + eps * sqrt(v); ``repeat`` tiles squeezed m, sqrt(v) along dim 0
  log_normal_mixture  :364-378
  log_normal          :381-397   constant is log(sqrt(2*pi)); "- log(sqrt(v))"
  log_sum_exp         :400-413
  log_mean_exp        :416-428
RNG is never drawn here: ``eps`` is an explicit input (the reference calls
``torch.randn_like(sqrt_v)``; fixtures record that draw).
"""
import math

import torch
import torch.nn.functional as F

LOG_SQRT_2PI = math.log(math.sqrt(2.0 * math.pi))


def gaussian_parameters(h, dim=-1):
    half = h.size(dim) // 2
    m, raw = torch.split(h, half, dim=dim)
    return m, F.softplus(raw) + 1e-8


def tile_for_sampling(m, v, repeat=1):
    """The (m, sqrt_v) pair that ``sample_gaussian`` multiplies eps into."""
    if repeat > 1:
        m, v = m.squeeze(), v.squeeze()
        return torch.cat([m] * repeat, dim=0), torch.cat([torch.sqrt(v)] * repeat, dim=0)
    return m, torch.sqrt(v)


def sample_gaussian(m, v, eps, repeat=1):
    mm, sd = tile_for_sampling(m, v, repeat)
    return mm + eps * sd


def log_normal(x, m, v):
    elem = -(x - m).pow(2) / (2 * v) - v.sqrt().log() - LOG_SQRT_2PI
    return elem.sum(dim=-1)


def log_sum_exp(x, dim=0):
    mx = torch.max(x, dim)[0]
    return mx + (x - mx.unsqueeze(dim).expand_as(x)).exp().sum(dim).log()


def log_mean_exp(x, dim):
    return log_sum_exp(x, dim) - math.log(x.size(dim))


def log_normal_mixture(z, m, v):
    return log_mean_exp(log_normal(z.unsqueeze(1), m, v), dim=-1)
