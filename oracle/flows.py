"""Oracle restatement of the MADE / IAF blocks  --  TEST INFRASTRUCTURE (pinned by golden vectors).

Follows /root/reference/kgvae/flow_network.py:
  MaskedLinear.forward   :14-15   F.linear(x, mask * weight, bias)
  PermuteLayer           :18-34   column reversal, log_det = zeros(N, 1)
  MADE.__init__          :44-63   D->H, [H->H] x n_hidden, H->2D, ReLU between
  MADE.create_masks      :65-83   degrees + masks (last mask repeated twice along rows)
  MADE.forward           :85-98   len(self.m) = n_hidden+3 sequential passes (NOT textbook IAF)
  MADE.inverse           :100-112 single pass (x - mu) * exp(-alpha)

A MADE block is described here by ``(input_size, hidden_size, n_hidden)`` plus a list of
``(weight, bias)`` pairs (``net.{0,2,4,...}.{weight,bias}`` in the reference state_dict).
"""
import torch
import torch.nn.functional as F

from . import bf16


def made_degrees(input_size, hidden_size, n_hidden):
    """flow_network.py:69-77 -- list of n_hidden+3 int64 degree vectors (``MADE.m``)."""
    deg = [torch.arange(input_size)]
    for _ in range(n_hidden + 1):
        deg.append(torch.arange(hidden_size) % (input_size - 1))
    deg.append(torch.arange(input_size) % input_size - 1)
    return deg


def made_masks(input_size, hidden_size, n_hidden):
    """flow_network.py:79-83 and :61-62 -- one float mask per linear layer (n_hidden+2 of them)."""
    deg = made_degrees(input_size, hidden_size, n_hidden)
    masks = [(hi.unsqueeze(-1) >= lo.unsqueeze(0)).float() for lo, hi in zip(deg[:-1], deg[1:])]
    masks[-1] = masks[-1].repeat(2, 1)
    return masks


def column_multiplicity(idx, input_size):
    """How many times each of the D columns occurs in an index set of ``MADE.m`` (int64 (D,))."""
    return torch.bincount(idx % input_size, minlength=input_size)


def made_net(x, layers, masks):
    """The masked MLP: ReLU after every layer but the last (flow_network.py:53-63)."""
    h = x
    last = len(layers) - 1
    for li, ((w, b), m) in enumerate(zip(layers, masks)):
        h = bf16.linear(h, m * w, b)
        if li != last:
            h = torch.relu(h)
    return h


def made_forward(z, layers, input_size, hidden_size, n_hidden):
    """flow_network.py:85-98.  Returns (x, log_det[N])."""
    masks = made_masks(input_size, hidden_size, n_hidden)
    x = torch.zeros_like(z)
    alpha = None
    for idx in made_degrees(input_size, hidden_size, n_hidden):
        mu, alpha = torch.chunk(made_net(x, layers, masks), 2, dim=1)
        # ``x[:, i] = z[:, i] * exp(alpha[:, i] + mu[:, i])`` with an index VECTOR i that may repeat a
        # column (passes 1..n_hidden+1 use [0..D-2, 0] when H == D) and may hold -1 (= last column).
        # Forward: duplicates write equal values.  Backward: autograd hands the column's gradient to
        # EVERY duplicate, so a column listed c times receives c x its gradient -- the golden
        # vectors pin that, and ``column_multiplicity`` exposes c for the HIP kernel.
        val = z * torch.exp(alpha + mu)
        x = x.clone()
        x[:, idx] = val[:, idx]
    return x, alpha.sum(dim=-1)


def made_inverse(x, layers, input_size, hidden_size, n_hidden):
    """flow_network.py:100-112.  Returns (z, log_det[N])."""
    masks = made_masks(input_size, hidden_size, n_hidden)
    mu, alpha = made_net(x, layers, masks).chunk(2, dim=-1)
    return (x - mu) * torch.exp(-alpha), (-alpha).sum(dim=-1)


def permute(x):
    """PermuteLayer.forward / .inverse (flow_network.py:28-34): reverse the columns."""
    return x.flip(1), torch.zeros(x.shape[0], 1, dtype=x.dtype)


def made_layers_from_state(state, prefix, n_hidden):
    """Pull ``(weight, bias)`` of ``<prefix>net.{0,2,..}`` out of a reference state_dict."""
    return [(state[f'{prefix}net.{2 * i}.weight'], state[f'{prefix}net.{2 * i}.bias'])
            for i in range(n_hidden + 2)]


def flow_chain_forward(z, state, prefix, n_flows, dim):
    """KGVAE's ``nf`` Sequential = [MADE(dim, dim, n_flows), PermuteLayer(dim)] * n_flows
    (kgvae/model.py:44-50) applied as in kgvae/model.py:116-123.  Returns (z, log_det_sum[N])."""
    total = torch.zeros(z.shape[0], dtype=z.dtype)
    for f in range(n_flows):
        layers = made_layers_from_state(state, f'{prefix}{2 * f}.', n_flows)
        z, ld = made_forward(z, layers, dim, dim, n_flows)
        total = total + ld
        z, _ = permute(z)
    return z, total


def flow_chain_inverse(x, state, prefix, n_flows, dim):
    """``for flow in self.nf[::-1]: x, _ = flow.inverse(x)`` (kgvae/model.py:66-68)."""
    for f in reversed(range(n_flows)):
        x, _ = permute(x)
        layers = made_layers_from_state(state, f'{prefix}{2 * f}.', n_flows)
        x, _ = made_inverse(x, layers, dim, dim, n_flows)
    return x
