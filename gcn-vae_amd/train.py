"""Link-prediction task head and driver with the reference's names, flags and control flow
(kgvae/link_predict.py:30-326), running on the gfx950 kernels.

    python -m gcn_vae_amd.train -d FB15k-237-synthetic --n-hidden 200 --n-bases 100 --n-flows 3 \
        --mmd-param 1 --mog-k 10 --gpu 0

Differences from the reference, all forced by its own crashes or by the device:
  * defaults that crash there (``--n-flows 0`` with ``--kl-param > 0``; ``--model-class RGCN``;
    the periodic validation with ``flow_log_prob=None``) run here;
  * validation stays on the GPU (the reference flips the model to the CPU because its scorer
    materialises a (h, Eb, V) tensor; ours is one GEMM);
  * the wall-clock spans synchronise the device before reading the clock;
  * ``--gpu`` must name a ROCm device: there is no CPU path.
"""
import argparse
import time

import numpy as np
import torch
import torch.nn as nn

from . import ops, ranking, sampling
from .optim import FlatAdam
from .data import load_data
from .encoders import KGVAE, RGCN
from .sampling import node_norm_to_edge_norm


class LinkPredict(ops.StayOnDevice, nn.Module):
    def __init__(self, model_class, in_dim, h_dim, num_rels, num_bases=-1, num_hidden_layers=1, dropout=0,
                 use_cuda=True, reg_param=0, kl_param=0, mmd_param=0, k=1, n_flows=0):
        super().__init__()
        self.encoder = model_class(num_nodes=in_dim, h_dim=h_dim, out_dim=h_dim, num_rels=num_rels * 2,
                                   num_bases=num_bases, num_hidden_layers=num_hidden_layers, dropout=dropout,
                                   use_self_loop=use_cuda, use_cuda=use_cuda, k=k, n_flows=n_flows)
        self.reg_param, self.kl_param, self.mmd_param = reg_param, kl_param, mmd_param
        if kl_param > 0 and hasattr(self.encoder, 'fuse_kl_with_reparam'):
            self.encoder.fuse_kl_with_reparam = True             # get_loss will ask for KL(z): its forward pass rides on the reparameterisation
        if mmd_param > 0 and hasattr(self.encoder, 'batch_mmd_prior_with_forward'):
            self.encoder.batch_mmd_prior_with_forward = True     # get_loss will ask for MMD: batch its flow passes
        self.w_relation = nn.Parameter(torch.Tensor(num_rels, h_dim))
        self.use_cuda, self.k, self.n_flows = use_cuda, k, n_flows
        nn.init.xavier_uniform_(self.w_relation, gain=nn.init.calculate_gain('relu'))
        self._tidx_key, self._tidx = None, None

    def triplet_index(self, embedding, triplets):
        """Index of a triplet batch for the DistMult backward; cached per (storage, version)."""
        key = (triplets.data_ptr(), triplets._version, tuple(triplets.shape), embedding.shape[0])
        if key != self._tidx_key:
            # a batch that is scored once (mini-batch training) gets the sync-free index; one that is reused every step
            # (full-graph training: set ``static_batch``) gets the exact, locality-ordered one
            self._tidx = ops.TripletIndex(triplets.to(embedding.device), embedding.shape[0], self.w_relation.shape[0],
                                          sync_free=not getattr(self, 'static_batch', False))
            self._tidx_key = key
            self._tidx_keepalive = triplets
        return self._tidx

    def calc_score(self, embedding, triplets):
        return ops.distmult_score(embedding, self.w_relation, self.triplet_index(embedding, triplets))

    def forward(self, g, h, r, norm):
        return self.encoder.forward(g, h, r, norm)

    def regularization_loss(self, embedding):
        return ops.mean_sq(embedding) + ops.mean_sq(self.w_relation)

    def get_loss(self, g, embed, triplets, labels):
        """(loss, predict_loss, kl, mmd) -- the four terms of kgvae/link_predict.py:71-92 from one fused
        autograd node (ops.loss_head); calc_score / regularization_loss / get_kl / get_mmd stay available
        as separate differentiable ops with the reference's signatures."""
        enc = self.encoder
        tidx = self.triplet_index(embed, triplets)
        vae = isinstance(enc, KGVAE)
        flp = enc.get_flow_log_prob() if vae else None
        kl_w = self.kl_param if vae else 0.0
        mmd_w = self.mmd_param if vae else 0.0
        z_pri, pick = enc.mmd_inputs(embed) if mmd_w > 0 else (None, None)
        part = getattr(enc, 'row_part', None) if vae else None
        if part is not None:
            return self._get_loss_rows(part, embed, tidx, flp, kl_w, mmd_w, z_pri, pick, labels)
        loss, predict_loss, kl, mmd = ops.loss_head(
            embed, enc.z_mean if kl_w > 0 else None, enc.z_sigma if kl_w > 0 else None, self.w_relation,
            enc.z_pre.squeeze(0) if kl_w > 0 else None, flp, z_pri, pick, labels, tidx, self.reg_param, kl_w, mmd_w,
            score_bias=self.n_flows > 0, rows_dev=getattr(self, 'rows_dev', None))
        # shapes as the reference returns them: a disabled term is ``zeros(1)`` there and broadcasts the loss to (1,)
        kl = kl.reshape(()) if kl_w > 0 else kl.reshape(1)
        mmd = mmd.reshape(()) if mmd_w > 0 else mmd.reshape(1)
        if kl_w <= 0 or mmd_w <= 0:
            loss = loss.reshape(1)
        return loss, predict_loss, kl, mmd


def _get_loss_rows(self, part, embed, tidx, flp, kl_w, mmd_w, z_pri, pick, labels):
    """get_loss under the destination-row partition.  ``embed`` is z of ALL positions; the decoder, the regulariser
    and MMD run on it (on this rank's triplets / draws), KL on the rank's own rows.  Returns the rank's SHARE of

        L = mean_p(pred_p + mmd_w mmd_p) + reg_w reg + kl_w KL     (the sum of the shares over the ranks is L)

    so that plain SUMS of the ranks' gradients (reduce-scatter of dL/dz, all-reduce of the parameter arena) are exact."""
    enc = self.encoder
    head, predict_loss, _, mmd = ops.loss_head(embed, None, None, self.w_relation, None, flp, z_pri, pick, labels, tidx,
                                               self.reg_param, 0.0, mmd_w, score_bias=self.n_flows > 0,
                                               embed_rows=part.real_rows)
    kl = None
    if kl_w > 0 and part.own_rows > 0:
        kl = ops.kl_to_mixture(enc.z_own, enc.z_mean, enc.z_sigma, enc.z_pre.squeeze(0), flp)   # mean over own rows
    loss = ops.lincomb2(head, 1.0 / part.world, kl, kl_w * part.own_rows / max(part.real_rows, 1))
    kl_share = (kl.detach() * (part.own_rows / max(part.real_rows, 1))).reshape(()) if kl is not None \
        else torch.zeros(1, device=embed.device)
    return loss, predict_loss, kl_share, (mmd.reshape(()) if mmd_w > 0 else mmd.reshape(1))


LinkPredict._get_loss_rows = _get_loss_rows


def host_state_dict(model):
    """The model's state_dict with every tensor on the HOST: the reference moves the model to the CPU before it saves
    (kgvae/link_predict.py:242-246), so its checkpoints load with a plain ``torch.load`` on a box without a GPU.  HIP modules stay
    on the device under ``.cpu()`` (ops.StayOnDevice), hence the explicit copy here."""
    return {k: v.detach().cpu() for k, v in model.state_dict().items()}


def _sync():
    torch.cuda.synchronize()


def main(args):
    data = load_data(args.dataset)
    num_nodes, num_rels = data.num_nodes, data.num_rels
    train_data, valid_data, test_data = data.train, data.valid, data.test

    if args.gpu < 0 or not torch.cuda.is_available():
        raise RuntimeError('gcn_vae_amd runs on a ROCm device only (pass --gpu N on an MI355X box); '
                           'there is no CPU path')
    torch.cuda.set_device(args.gpu)
    dev = torch.device('cuda', args.gpu)
    ops.set_gemm_precision('bf16' if getattr(args, 'bf16', False) else 'f32')

    model_class = KGVAE if args.model_class == "KGVAE" else RGCN
    model = LinkPredict(model_class=model_class, in_dim=num_nodes, h_dim=args.n_hidden, num_rels=num_rels,
                        num_bases=args.n_bases, num_hidden_layers=args.n_layers, dropout=args.dropout, use_cuda=True,
                        reg_param=args.regularization, kl_param=args.kl_param, mmd_param=args.mmd_param,
                        k=args.mog_k, n_flows=args.n_flows)
    model.to(dev)

    def graph_inputs(triplets):
        g, rel, norm = sampling.build_test_graph(num_nodes, num_rels, triplets)
        node_id = torch.arange(0, num_nodes, dtype=torch.long, device=dev).view(-1, 1)
        enorm = node_norm_to_edge_norm(g, torch.from_numpy(norm).view(-1, 1)).to(dev)
        return g, node_id, torch.from_numpy(rel).to(dev), enorm

    valid_t = torch.as_tensor(valid_data, dtype=torch.long, device=dev)
    val_graph, val_node_id, val_rel, val_norm = graph_inputs(valid_data)     # eval graph from VALID triplets (:141-147)
    adj_list, degrees = sampling.get_adj_and_degrees(num_nodes, train_data)
    optimizer = FlatAdam(model.parameters(), lr=args.lr, max_grad_norm=args.grad_norm)   # clip + Adam, one arena
    forward_time, backward_time, step_time = [], [], []

    if args.test_mode is True:
        print("\nstart testing:")
        checkpoint = torch.load(args.model_state_file, map_location=dev)
        test_t = torch.as_tensor(test_data, dtype=torch.long, device=dev)
        test_graph, test_node_id, test_rel, test_norm = graph_inputs(test_data)
        model.eval()
        model.load_state_dict(checkpoint['state_dict'])
        print("Using best epoch: {}".format(checkpoint['epoch']))
        with torch.no_grad():
            embed = model(test_graph, test_node_id, test_rel, test_norm)
        return ranking.calc_mrr(embed, model.w_relation, test_t, hits=[1, 3, 10], eval_bz=args.eval_batch_size,
                                all_batches=True, flow_log_prob=model.encoder.get_flow_log_prob())

    print("start training...")
    epoch, best_mrr = 0, 0
    if args.load is True:
        print(f"Loading checkpoint file {args.model_state_file} for training")
        checkpoint = torch.load(args.model_state_file, map_location=dev)
        model.load_state_dict(checkpoint['state_dict'])
        epoch = checkpoint['epoch']

    dev_sampler = None
    if getattr(args, 'device_sampler', False):
        from .device_sampling import DeviceSampler
        dev_sampler = DeviceSampler(train_data, num_nodes, num_rels, dev, sampler=args.edge_sampler)

    graphed = None
    if getattr(args, 'graph_step', False):
        if dev_sampler is None:
            raise ValueError('--graph-step records the device sampler with the step: pass --device-sampler too')
        from .graph_step import GraphedMiniBatchStep
        if args.kl_param <= 0:
            raise ValueError('--graph-step needs --kl-param > 0 (the captured loss head reads the device row count there)')
        model.train()
        graphed = GraphedMiniBatchStep(model, optimizer, dev_sampler, args.graph_batch_size, args.graph_split_size,
                                       args.negative_sample)
        graph_warmup = 3                             # the first steps run eagerly (ordinary epochs: printed, evaluated), then the recording

    while True:
        model.train()
        epoch += 1
        if graphed is not None:      # the whole step -- sampling to Adam -- is one hipGraph replay
            if graphed.graph is None and graph_warmup == 0:
                graphed.capture(warmup=0)
            graph_warmup = max(0, graph_warmup - 1)
            _sync()
            t0 = time.time()
            loss, pred_loss, kl, mmd = graphed()
            _sync()
            step_time.append(time.time() - t0)
            print("Epoch {:04d} | Loss {:.4f} | Best MRR {:.4f} | pred_loss {:.4f} | kl {:.4f} | mmd {:.4f}".format(
                epoch, loss.item(), best_mrr, pred_loss.item(), kl.item(), mmd.item()))
        elif dev_sampler is not None:
            b = dev_sampler.sample(args.graph_batch_size, args.graph_split_size, args.negative_sample)
            g, node_id, edge_type, edge_norm, batch, labels = b.g, b.node_id, b.edge_type, b.edge_norm, b.samples, b.labels
        else:
            g, node_id, edge_type, node_norm, batch, labels = sampling.generate_sampled_graph_and_labels(
                train_data, args.graph_batch_size, args.graph_split_size, num_rels, adj_list, degrees,
                args.negative_sample, args.edge_sampler)
            node_id = torch.from_numpy(node_id).view(-1, 1).long().to(dev)
            edge_type = torch.from_numpy(edge_type).to(dev)
            edge_norm = node_norm_to_edge_norm(g, torch.from_numpy(node_norm).view(-1, 1)).to(dev)
            batch, labels = torch.from_numpy(batch).to(dev), torch.from_numpy(labels).to(dev)

        if graphed is None:
            _sync()
            t0 = time.time()
            embed = model(g, node_id, edge_type, edge_norm)
            loss, pred_loss, kl, mmd = model.get_loss(g, embed, batch, labels)
            _sync()
            t1 = time.time()
            loss.backward()
            optimizer.step()              # clip_grad_norm_(grad_norm) + Adam, fused
            _sync()
            t2 = time.time()
            forward_time.append(t1 - t0)
            backward_time.append(t2 - t1)
            print("Epoch {:04d} | Loss {:.4f} | Best MRR {:.4f} | pred_loss {:.4f} | kl {:.4f} | mmd {:.4f}".format(
                epoch, loss.item(), best_mrr, pred_loss.item(), kl.item(), mmd.item()))
            optimizer.zero_grad()

        if epoch % args.evaluate_every == 0:
            model.eval()
            print("start eval")
            torch.save({'state_dict': host_state_dict(model), 'epoch': epoch}, args.model_state_file)
            with torch.no_grad():
                embed = model(val_graph, val_node_id, val_rel, val_norm)
            mrr = ranking.calc_mrr(embed, model.w_relation, valid_t, hits=[1, 3, 10], eval_bz=args.eval_batch_size,
                                   all_batches=False, flow_log_prob=model.encoder.get_flow_log_prob())
            if mrr < best_mrr:
                torch.save({'state_dict': host_state_dict(model), 'epoch': epoch}, args.model_state_file + "_latest")
            else:
                best_mrr = mrr
                torch.save({'state_dict': host_state_dict(model), 'epoch': epoch}, args.model_state_file)
        if epoch >= args.n_epochs:
            break

    print("training done")
    if step_time:     # the captured step has no forward / backward boundary on the host: one figure
        print("Mean step time (forward + backward + update, one hipGraph): {:4f}s".format(np.mean(step_time)))
    else:
        print("Mean forward time: {:4f}s".format(np.mean(forward_time)))
        print("Mean Backward time: {:4f}s".format(np.mean(backward_time)))
    return best_mrr


def build_parser():
    p = argparse.ArgumentParser(description='Link Prediction')
    p.add_argument("--dropout", type=float, default=0.2, help="dropout probability")
    p.add_argument("--n-hidden", type=int, default=500, help="number of hidden units")
    p.add_argument("--gpu", type=int, default=-1, help="gpu")
    p.add_argument("--lr", type=float, default=1e-3, help="learning rate")
    p.add_argument("--n-bases", type=int, default=100, help="number of weight blocks for each relation")
    p.add_argument("--n-layers", type=int, default=2, help="number of propagation rounds")
    p.add_argument("--n-epochs", type=int, default=1e5, help="number of minimum training epochs")
    p.add_argument("-d", "--dataset", type=str, required=True, help="dataset to use")
    p.add_argument("--eval-batch-size", type=int, default=400, help="batch size when evaluating")
    p.add_argument("--regularization", type=float, default=0.01, help="regularization weight")
    p.add_argument("--kl-param", type=float, default=1e-5, help="kl regularization weight")
    p.add_argument("--mmd-param", type=float, default=0, help="mmd regularization weight")
    p.add_argument("--mog-k", type=int, default=10, help="number of mixture of gaussian")
    p.add_argument("--n-flows", type=int, default=0, help="number of flow transform layers")
    p.add_argument("--grad-norm", type=float, default=1.0, help="norm to clip gradient to")
    p.add_argument("--graph-batch-size", type=int, default=20000, help="number of edges to sample in each iteration")
    p.add_argument("--graph-split-size", type=float, default=0.5, help="portion of edges used as positive sample")
    p.add_argument("--negative-sample", type=int, default=10, help="number of negative samples per positive sample")
    p.add_argument("--evaluate-every", type=int, default=200, help="perform evaluation every n epochs")
    p.add_argument("--edge-sampler", type=str, default="uniform", help="type of edge sampler: 'uniform' or 'neighbor'")
    p.add_argument("--test-mode", type=bool, default=False, help="only evaluate on test dataset")
    p.add_argument("--model-state-file", type=str, default='model_state.pth', help="model state file to load or save")
    p.add_argument("--model-class", type=str, default='KGVAE', help="model class")
    p.add_argument("--load", type=bool, default=False, help="whether to load a model state file for training")
    p.add_argument("--generate", type=bool, default=False, help="(reference demo; not supported here)")
    p.add_argument("--bf16", action="store_true",
                   help="dense products (MaskedLinear, self-loop term, evaluation scorer) with bf16 operands and fp32 "
                        "accumulation (BASELINE configs[2]); not a reference flag, default fp32")
    p.add_argument("--graph-step", action="store_true",
                   help="with --device-sampler: record sampling + forward + loss + backward + clip/Adam once and replay it "
                        "as one hipGraph per step (static shapes: node rows padded, counts kept on the device)")
    p.add_argument("--device-sampler", action="store_true",
                   help="prepare batches on the GPU (uniform sampler, torch's device RNG instead of numpy's: not the "
                        "reference's random stream, ~10x less host time per step)")
    return p


if __name__ == '__main__':
    cli = build_parser().parse_args()
    print(cli)
    main(cli)
