"""Encoders with the reference's class names and signatures (kgvae/model.py:13-211):
``KGVAE`` (embedding -> 2 x RelGraphConv(bdd) -> Gaussian parameters -> reparameterise -> IAF),
``BaseRGCN`` / ``RGCN`` and the small wrapper layers.  state_dict keys are the reference's."""
import random

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, prob
from .flows import MADE, PermuteLayer
from .layers import RelGraphConv


class EmbeddingLayer(ops.StayOnDevice, nn.Module):
    def __init__(self, num_nodes, h_dim):
        super().__init__()
        self.embedding = nn.Embedding(num_nodes, h_dim)

    def forward(self, g, h, r, norm):
        h = ops.to_module_device(self.embedding.weight, h)
        # the lookup opens every forward pass: the device RNG's tick advances on this launch
        return ops.embedding(self.embedding.weight, h.squeeze(), ops.device_rng(self.embedding.weight.device),
                             sole_consumer=True)


class DistLayer(ops.StayOnDevice, nn.Module):
    """Present in the reference (kgvae/model.py:194-200) but never instantiated."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.linear = nn.Linear(in_dim, out_dim)

    def forward(self, g, h, r, norm):
        return ops.linear(h.squeeze(), self.linear.weight, self.linear.bias)


class KGVAE(ops.StayOnDevice, nn.Module):
    def __init__(self, num_nodes, h_dim, out_dim, num_rels, num_bases, num_hidden_layers=1, dropout=0,
                 use_self_loop=False, use_cuda=True, k=10, n_flows=0, verbose=False):
        super().__init__()
        self.num_nodes, self.h_dim, self.out_dim, self.num_rels = num_nodes, h_dim, out_dim, num_rels
        self.num_bases = None if num_bases < 0 else num_bases
        self.num_hidden_layers, self.dropout = num_hidden_layers, dropout
        self.use_self_loop, self.use_cuda = use_self_loop, use_cuda
        self.k = k
        if verbose:
            print("use cuda", self.use_cuda)
            print(f"Mixture of {self.k} Gaussians.")
        self.flow_log_prob = None
        self.build_encoder()
        self.z_pre = nn.Parameter(torch.randn(1, 2 * self.k, self.h_dim) / np.sqrt(self.k * self.h_dim))
        self.pi = nn.Parameter(torch.ones(k) / k, requires_grad=False)
        self.n_flows = n_flows
        if self.n_flows > 0:
            if verbose:
                print(f"Sequence of {self.n_flows} Flow transforms.")
            self.build_iaf()
        # parity / graph-capture hooks (None = draw on the device / host as the reference does)
        self.eps_override = None          # (N, h) standard-normal draw for the reparameterisation
        self.mmd_eps_override = None      # (num_sample, h) draw for the prior samples of get_mmd
        self.mmd_index_override = None    # int64 (num_sample,) posterior row pick of get_mmd
        self.rng_stream_eps, self.rng_stream_prior = ops.new_rng_stream(), ops.new_rng_stream()
        self._prior_eps_next = None       # prior noise drawn by forward()'s fused RNG launch
        # get_mmd pushes 200 prior samples through the same flow stack as the N node rows.  The flows act row-wise,
        # so when the task head announces that MMD will be evaluated (LinkPredict sets this for mmd_param > 0) the
        # prior rows ride along in forward() as extra rows of the same GEMMs instead of ~150 tiny launches of their own.
        self.batch_mmd_prior_with_forward = False
        self.rows_dev = None              # device int32 (1,): how many node rows of a padded static-shape batch exist (graph_step)
        self.fuse_kl_with_reparam = False # set by the task head when it will evaluate get_kl on forward()'s z (LinkPredict, kl_param > 0)
        self.grad_reducer = None          # distributed.BucketedArenaReduce: multi-GPU gradient exchange started under backward
        self._z_pri_flowed = None
        # multi-GPU destination-row partition (distributed.RowPartition): this rank computes its own row block only;
        # forward() then takes the rank's distributed.RowBlockGraph and the node ids of ALL positions, and returns the
        # all-gathered z of all positions (padding rows zero).  None = the whole graph on this device.
        self.row_part = None

    def build_iaf(self):
        blocks = []
        for _ in range(self.n_flows):
            blocks.append(MADE(self.h_dim, self.h_dim, self.n_flows))
            blocks.append(PermuteLayer(self.h_dim))
        self.nf = nn.Sequential(*blocks)

    def build_encoder(self):
        self.input_layer = EmbeddingLayer(self.num_nodes, self.h_dim)
        self.rconv_layer_1 = RelGraphConv(self.h_dim, self.h_dim, self.num_rels, "bdd", self.num_bases,
                                          activation=nn.ReLU(), self_loop=True, dropout=self.dropout)
        self.rconv_layer_2 = RelGraphConv(self.h_dim, self.h_dim * 2, self.num_rels, "bdd", self.num_bases,
                                          activation=None, self_loop=True, dropout=self.dropout)

    def sample_z(self, batch):
        m, v = prob.gaussian_parameters(self.z_pre.squeeze(0), dim=0)
        idx = torch.distributions.categorical.Categorical(self.pi).sample((batch,))
        x = prob.sample_gaussian(m[idx], v[idx])
        if self.n_flows > 0:
            for flow in self.nf[::-1]:
                x, _ = flow.inverse(x)
        return x

    def compute_kernel(self, x, y):
        dim = x.size(1)
        d2 = (x.unsqueeze(1) - y.unsqueeze(0)).pow(2).mean(2) / float(dim)
        return torch.exp(-d2)

    def get_kl(self, z):
        # the reference adds ``self.flow_log_prob`` unconditionally and crashes when n_flows == 0
        # (None + tensor, kgvae/model.py:86); None is read as "no flow term".
        return ops.kl_to_mixture(z, self.z_mean, self.z_sigma, self.z_pre.squeeze(0), self.flow_log_prob)

    def _prior_draw(self, device, dtype):
        num_sample = 200
        rows = (num_sample // self.k) * self.k if num_sample // self.k > 1 else self.k
        if self.mmd_eps_override is not None:
            eps = self.mmd_eps_override
        elif self._prior_eps_next is not None:
            eps, self._prior_eps_next = self._prior_eps_next, None
        else:
            eps = torch.empty(rows, self.h_dim, device=device, dtype=dtype)
            ops.device_rng(device).fill([(eps, ops.RNG_NORMAL, 0.0, self.rng_stream_prior)])
        return ops.prior_sample(self.z_pre.squeeze(0), eps)       # sample_gaussian(m_mix, s_mix, repeat)

    def _prior_rows(self):
        num_sample = 200
        return (num_sample // self.k) * self.k if num_sample // self.k > 1 else self.k

    def _draw_noise(self, n, device):
        """All random draws of one forward pass in ONE launch: both layers' dropout masks, the reparameterisation
        noise and (when the head announced get_mmd) the prior noise.  Overrides (parity mode) are left alone."""
        jobs, eps = [], self.eps_override
        for layer in (self.rconv_layer_1, self.rconv_layer_2):
            if layer.wants_keep_mask():
                layer._keep_next = torch.empty(n, layer.out_feat, dtype=torch.uint8, device=device)
                jobs.append(layer.keep_job(layer._keep_next))
        if eps is None:
            eps = torch.empty(n, self.h_dim, dtype=torch.float32, device=device)
            jobs.append((eps, ops.RNG_NORMAL, 0.0, self.rng_stream_eps))
        if self.batch_mmd_prior_with_forward and self.training and self.mmd_eps_override is None:
            self._prior_eps_next = torch.empty(self._prior_rows(), self.h_dim, dtype=torch.float32, device=device)
            jobs.append((self._prior_eps_next, ops.RNG_NORMAL, 0.0, self.rng_stream_prior))
        if jobs:
            ops.device_rng(device).fill(jobs)
        return eps

    def mmd_inputs(self, z):
        """The two sample sets of get_mmd: prior draws (through the flows) and the posterior row pick."""
        num_sample = 200
        if self._z_pri_flowed is not None:        # already carried through the flows by forward()
            z_pri, self._z_pri_flowed = self._z_pri_flowed, None
        else:
            z_pri = self._prior_draw(z.device, z.dtype)
            if self.n_flows > 0:
                for flow in self.nf:
                    z_pri, _ = flow.forward(z_pri)
        if self.mmd_index_override is not None:
            pick = self.mmd_index_override
        elif self.row_part is not None and getattr(self.row_part, 'real_positions', None) is not None:
            real = self.row_part.real_positions        # row partition: z holds all positions, some are padding
            pick = torch.as_tensor(real[random.sample(range(len(real)), num_sample)], device=z.device)
        else:   # (Monte Carlo) posterior rows, python RNG as in the reference
            pick = torch.tensor(random.sample(range(z.shape[0]), num_sample), device=z.device)
        return z_pri, pick

    def get_mmd(self, z):
        z_pri, pick = self.mmd_inputs(z)
        return ops.mmd(z_pri, ops.embedding(z, pick))

    def get_flow_log_prob(self):
        return self.flow_log_prob

    def _forward_rows(self, g, h, r, norm):
        """forward() under the destination-row partition: embedding lookup of all positions (replicated), layer 1 on the
        rank's rows, all-gather, layer 2 on the rank's rows, reparameterisation of the rank's rows, all-gather of z."""
        from .distributed import AllGatherRows
        from .distributed import AllReduceSum
        part = self.row_part
        c = part.own_rows
        x0 = self.input_layer(g, h, r, norm)                                   # (total_rows, h)
        eps = self._draw_noise(c, x0.device)
        if eps.shape[0] != c:          # parity overrides are given for all positions: take the rank's rows
            eps = eps[part.row0:part.row0 + c].contiguous()
        if getattr(part, 'chunks', 1) > 1:
            # pipelined exchange (GV_DIST_ROW_CHUNKS): layer 1 gathers its rows block by block under its own aggregation; layer 2's
            # backward reduce-scatters dL/dh1 block by block under its own K1^T
            h1 = self.rconv_layer_1.forward_rows(g, x0, r, norm, part, gather_input=False, pad_output=True, gather_output=True)
            h2 = self.rconv_layer_2.forward_rows(g, h1, r, norm, part, gather_input=False, pad_output=False, x_gathered=True)
        else:
            h1 = self.rconv_layer_1.forward_rows(g, x0, r, norm, part, gather_input=False, pad_output=True)
            h2 = self.rconv_layer_2.forward_rows(g, h1, r, norm, part, gather_input=True, pad_output=False)
        z, self.z_mean, self.z_sigma = ops.reparam(h2, eps)
        self.flow_log_prob = None
        self._z_pri_flowed = None
        if self.n_flows > 0:
            # the flows are row-wise: they run on the rank's rows; flow_log_prob = mean over ALL rows of the row sums
            z, log_det_sum = self._apply_flows(z)
            self.flow_log_prob = (AllReduceSum.apply(log_det_sum.sum().reshape(1), part.group) / part.real_rows).reshape(())
        self.z_own = z
        return AllGatherRows.apply(ops.pad_rows(z, part.slot_rows), part)

    def _apply_flows(self, z, want_mean=False):
        """The IAF stack on the rows of z (kgvae/model.py:116-123); get_mmd's prior rows ride along when announced.
        Returns (z after the flows, per-row sum of the MADE log-determinants)."""
        n = z.shape[0]
        ride_along = self.batch_mmd_prior_with_forward and self.training
        if ride_along:
            z = ops.cat_rows(z, self._prior_draw(z.device, z.dtype))      # (kernel copies: no memcpy node in a captured step)
        log_dets, flows, i = [], list(self.nf), 0
        # (the bf16 MADE nodes' row blocks stay forked from the first block to the last: ops.made.keep_row_blocks_forked)
        with ops.made.keep_row_blocks_forked(z.shape[0]):
            while i < len(flows):
                flow = flows[i]
                if isinstance(flow, MADE):            # PermuteLayer contributes zeros
                    # (a PermuteLayer behind the block rides in the block's last launch: the columns come out reversed)
                    fold = i + 1 < len(flows) and type(flows[i + 1]) is PermuteLayer
                    z, log_det = flow.forward(z, reverse_out=fold)
                    log_dets.append(log_det)
                    i += 2 if fold else 1
                else:
                    ops.made.fork_sync()
                    z, log_det = flow.forward(z)
                    i += 1
        if ride_along:      # (ops.split_rows: the slices' backward without a memcpy node in a captured step)
            z, self._z_pri_flowed = ops.split_rows(z, n)
        if want_mean and 1 <= len(log_dets) <= 8:     # flow_log_prob in ONE launch (and one in backward) instead of adds + mask + sum + divide
            return z, ops.mean_rows_multi(log_dets, n=n, rows_dev=self.rows_dev)
        log_det_sum = None
        for log_det in log_dets:
            log_det_sum = log_det if log_det_sum is None else log_det_sum + log_det
        if ride_along:
            log_det_sum = ops.split_rows(log_det_sum, n)[0]
        return z, log_det_sum

    def forward(self, g, h, r, norm):
        h, r, norm = ops.to_module_device(self.z_pre, h, r, norm)      # host tensors (the reference's validation block) come to the device
        self.node_id = h.squeeze()
        if self.row_part is not None:
            return self._forward_rows(g, h, r, norm)
        if self.n_flows > 0:      # the flows' parameter-only work (mask folds, packed weights, pass 0's row) beside the encoder's layers
            ops.made_prepare([f.call_arguments() for f in self.nf if isinstance(f, MADE)])
        h = self.input_layer(g, h, r, norm)
        eps = self._draw_noise(h.shape[0], h.device)
        h = self.rconv_layer_1(g, h, r, norm)
        if self.grad_reducer is not None:     # layer 2 onwards is final once dL/dh1 exists: its arena suffix reduces under layer 1's backward
            h = self.grad_reducer.milestone(h, next(self.rconv_layer_2.parameters()))
        h = self.rconv_layer_2(g, h, r, norm)
        # no flow between z and the KL term: the KL forward pass rides on the reparameterisation's sweep over the rows, and
        # its backward is chained through the reparameterisation's (ops.reparam); with flows the two stay separate
        fuse_kl = self.training and self.n_flows == 0 and self.fuse_kl_with_reparam
        z, self.z_mean, self.z_sigma = ops.reparam(h, eps, self.z_pre.squeeze(0) if fuse_kl else None)
        self._z_pri_flowed = None
        if self.n_flows > 0:
            # flow_log_prob = mean over the rows that exist (a static-shape batch: the device count) of the blocks' summed log-dets
            z, self.flow_log_prob = self._apply_flows(z, want_mean=True)
            ops.made_prepare_finish()
            if self.flow_log_prob.dim() != 0:       # (more than 8 blocks: the per-row sum came back)
                log_det_sum = self.flow_log_prob
                if self.rows_dev is None:
                    self.flow_log_prob = torch.mean(log_det_sum.view(-1, 1))
                else:
                    rows = self.rows_dev.reshape(()).to(log_det_sum.dtype)
                    live = torch.arange(log_det_sum.numel(), device=log_det_sum.device) < self.rows_dev.reshape(())
                    self.flow_log_prob = (log_det_sum.reshape(-1) * live).sum() / rows
        return z


class BaseRGCN(ops.StayOnDevice, nn.Module):
    def __init__(self, num_nodes, h_dim, out_dim, num_rels, num_bases, num_hidden_layers=1, dropout=0,
                 use_self_loop=False, use_cuda=False, **unused):
        # **unused: LinkPredict passes k= / n_flows= to every encoder class; the reference's BaseRGCN
        # does not accept them and ``--model-class RGCN`` dies with a TypeError (kgvae/link_predict.py:35-46).
        super().__init__()
        self.num_nodes, self.h_dim, self.out_dim, self.num_rels = num_nodes, h_dim, out_dim, num_rels
        self.num_bases = None if num_bases < 0 else num_bases
        self.num_hidden_layers, self.dropout = num_hidden_layers, dropout
        self.use_self_loop, self.use_cuda = use_self_loop, use_cuda
        self.build_model()

    def build_model(self):
        self.layers = nn.ModuleList()
        i2h = self.build_input_layer()
        if i2h is not None:
            self.layers.append(i2h)
        for idx in range(self.num_hidden_layers):
            self.layers.append(self.build_hidden_layer(idx))
        h2o = self.build_output_layer()
        if h2o is not None:
            self.layers.append(h2o)

    def build_input_layer(self):
        return None

    def build_hidden_layer(self, idx):
        raise NotImplementedError

    def build_output_layer(self):
        return None

    def forward(self, g, h, r, norm):
        for layer in self.layers:
            h = layer(g, h, r, norm)
        return h

    def get_kl(self, z):
        return torch.zeros(1, device=z.device)

    def get_flow_log_prob(self):
        return None


class RGCN(BaseRGCN):
    def build_input_layer(self):
        return EmbeddingLayer(self.num_nodes, self.h_dim)

    def build_hidden_layer(self, idx):
        act = F.relu if idx < self.num_hidden_layers - 1 else None
        return RelGraphConv(self.h_dim, self.h_dim, self.num_rels, "bdd", self.num_bases, activation=act,
                            self_loop=True, dropout=self.dropout)
