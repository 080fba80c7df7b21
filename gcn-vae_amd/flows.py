"""MADE / IAF blocks on the gfx950 kernels (masked linears = f32 MFMA GEMM with fused bias+ReLU,
IAF update = gv_iaf_update_*).  Class names, constructor arguments, ``forward``/``inverse``
results and state_dict keys (``net.{0,2,..}.{weight,bias,mask}``) follow
/root/reference/kgvae/flow_network.py:7-112 -- including its six-pass ``forward`` (one pass per
entry of ``self.m``) and the column multiplicity of the ``x[:, i] = ...`` gradient."""
import numpy as np
import torch
import torch.nn as nn

from . import ops


class MaskedLinear(ops.StayOnDevice, nn.Linear):
    def __init__(self, input_size, output_size, mask):
        super().__init__(input_size, output_size)
        self.register_buffer('mask', mask)

    def masked_weight(self):
        return ops.masked_weight(self.mask, self.weight)

    def forward(self, x, act=ops.ACT_NONE, weight=None):
        return ops.linear(x, self.masked_weight() if weight is None else weight, self.bias, act)


class PermuteLayer(nn.Module):
    def __init__(self, num_inputs):
        super().__init__()
        self.perm = np.array(np.arange(0, num_inputs)[::-1])

    def forward(self, inputs):
        # log-det of a permutation: zeros (kgvae/flow_network.py:28-30) -- one constant tensor per shape, not a fill per call
        key = (inputs.size(0), inputs.device)
        zero = getattr(self, '_zero', None)
        if zero is None or zero[0] != key:
            zero = self._zero = (key, torch.zeros(inputs.size(0), 1, device=inputs.device))
        return ops.reverse_cols(inputs), zero[1]

    def inverse(self, inputs):
        return self.forward(inputs)


class MADE(ops.StayOnDevice, nn.Module):
    def __init__(self, input_size, hidden_size, n_hidden):
        super().__init__()
        self.input_size, self.hidden_size, self.n_hidden = input_size, hidden_size, n_hidden
        masks = self.create_masks()
        layers = [MaskedLinear(input_size, hidden_size, masks[0]), nn.ReLU(inplace=True)]
        for i in range(n_hidden):
            layers += [MaskedLinear(hidden_size, hidden_size, masks[i + 1]), nn.ReLU(inplace=True)]
        layers += [MaskedLinear(hidden_size, input_size * 2, masks[-1].repeat(2, 1))]
        self.net = nn.Sequential(*layers)
        # per pass: how often each column occurs in the index set (0 = column untouched)
        counts = torch.stack([torch.bincount(idx % input_size, minlength=input_size) for idx in self.m])
        if bool((counts[0] == 0).any()):     # the fused node evaluates pass 0 on one broadcast row and relies on this
            raise ValueError('MADE: the first index set must cover every column (arange(D) in the reference)')
        self.register_buffer('_colcount', counts.to(torch.int32), persistent=False)

    def create_masks(self):
        d, h = self.input_size, self.hidden_size
        degrees = [torch.arange(d)] + [torch.arange(h) % (d - 1) for _ in range(self.n_hidden + 1)]
        degrees.append(torch.arange(d) % d - 1)
        self.m = degrees
        return [(hi.unsqueeze(-1) >= lo.unsqueeze(0)).float() for lo, hi in zip(degrees[:-1], degrees[1:])]

    def _linears(self):
        return [m for m in self.net if isinstance(m, MaskedLinear)]

    def _run_net(self, x, weights):
        lin = self._linears()
        for i, (layer, w) in enumerate(zip(lin, weights)):
            x = layer(x, ops.ACT_RELU if i + 1 < len(lin) else ops.ACT_NONE, w)
        return x

    def call_arguments(self):
        """(colcount, weights, biases, masks) of the node forward() runs: what ops.made_prepare needs to do the parameter-only part
        of the call ahead of time."""
        lin = self._linears()
        return self._colcount, [l.weight for l in lin], [l.bias for l in lin], [l.mask for l in lin]

    def forward(self, z, reverse_out=False):
        """``reverse_out``: x comes back with its columns reversed -- the PermuteLayer that follows the block in an IAF stack
        (kgvae/model.py:60-66) folded into the node's last launch (the caller then skips that layer)."""
        colcount, weights, biases, masks = self.call_arguments()
        # masks folded once per call, not per pass (the bf16 node folds all layers in one launch, forward and backward)
        return ops.made_forward(z, colcount, weights, biases, masks=masks, reverse_out=reverse_out)

    def forward_unfused(self, z):
        """The same computation as a chain of per-op autograd nodes (kept for cross-checking the fused node)."""
        d = self.input_size
        weights = [l.masked_weight() for l in self._linears()]
        x = torch.zeros_like(z)
        net_out = None
        for p in range(len(self.m)):
            net_out = self._run_net(x, weights)
            x = ops.iaf_update(z, net_out, x, self._colcount[p])
        return x, ops.rowsum_cols(net_out, d, d)

    def inverse(self, x):
        d = self.input_size
        out = self._run_net(x, [l.masked_weight() for l in self._linears()])
        mu, alpha = out[:, :d], out[:, d:]
        return (x - mu) * torch.exp(-alpha), -ops.rowsum_cols(out, d, d)
