"""Raw-MRR evaluation with the reference's function names (kgvae/utils.py:180-221, :293-314).

The reference scores a batch against every entity by materialising a (h, Eb, V) outer-product
tensor and summing over h, adds ``flow_log_prob``, applies a sigmoid, sorts every row and looks the
target up.  Here ``ops.rank_scores`` (gv_rank_scores) forms (e_a * w_r) @ E^T + flow_log_prob tile by tile on the
f32 MFMA and counts, per query, the entities that beat the target -- the score matrix is never stored and
thousands of queries go through one launch.  Ranking is done on the LOGITS: the sigmoid is monotone, so tie-free
ranks equal the reference's sort-and-find, but it saturates (every candidate at 1.0f once the logits are large --
likely with ``flow_log_prob`` added to all of them), where the reference's rank is wherever ``torch.sort`` happens
to leave the target inside the tie block.  Ties are counted explicitly: rank = #better + #equal / 2 (the expected
position in such a block; ranks are floats), and a NaN target score (diverged run) ranks LAST -- never the
optimistic rank 1 that would make ``main`` keep a broken checkpoint as "best".
``perturb_and_get_rank_unfused`` keeps the materialised form (one GEMM + torch ops) as the in-repo cross-check.
"""
import torch

from . import ops


def sort_and_rank(score, target):
    """0-based mid-rank of ``target`` in every row of a materialised (logit) score matrix; NaN never ranks well."""
    tgt = score.gather(1, target.view(-1, 1))
    other = torch.ones_like(score, dtype=torch.bool).scatter_(1, target.view(-1, 1), False)
    better = (~(score <= tgt)) & other
    equal = (score == tgt) & other
    return better.sum(dim=1).float() + 0.5 * equal.sum(dim=1).float()


MAX_QUERY_ROWS = 16384      # queries per gv_rank_scores launch (bounds the (rows, h) query matrix, nothing else)


def perturb_and_get_rank(embedding, w, a, r, b, test_size, batch_size=100, all_batches=True, flow_log_prob=None,
                         verbose=False):
    """Ranks of ``b`` for the queries (a, r).  ``batch_size`` only matters with ``all_batches=False`` (the reference's
    quick validation scores the first batch only, kgvae/utils.py:183-186): the fused scorer has no (h, Eb, V) tensor to bound."""
    n = min(test_size, batch_size) if all_batches is False else test_size
    emb = embedding.detach().contiguous()
    wd = w.detach()
    ranks = []
    for lo in range(0, n, MAX_QUERY_ROWS):
        hi = min(n, lo + MAX_QUERY_ROWS)
        q = ops.mul(emb[a[lo:hi]].contiguous(), wd[r[lo:hi]].contiguous())
        ranks.append(ops.rank_scores(q, emb, b[lo:hi], flow_log_prob))
        if verbose:
            rr = 1.0 + torch.cat(ranks).float()
            print("rows {} / {}: MR : {:.6f} |  MRR : {:.6f}".format(hi, n, rr.mean().item(), (1.0 / rr).mean().item()))
    return torch.cat(ranks) if ranks else torch.zeros(0, dtype=torch.float32, device=emb.device)


def perturb_and_get_rank_unfused(embedding, w, a, r, b, test_size, batch_size=100, all_batches=True, flow_log_prob=None,
                                 verbose=False):
    n_batch = (test_size + batch_size - 1) // batch_size
    if all_batches is False:
        n_batch = 1
    ranks = []
    emb = embedding.detach().contiguous()
    for idx in range(n_batch):
        lo, hi = idx * batch_size, min(test_size, (idx + 1) * batch_size)
        emb_ar = ops.mul(emb[a[lo:hi]].contiguous(), w.detach()[r[lo:hi]].contiguous())
        score = ops.gemm(emb_ar, emb, trans_b=True)                       # (Eb, V)
        if flow_log_prob is not None:
            score = score + flow_log_prob
        ranks.append(sort_and_rank(score, b[lo:hi]))              # on the logits (see the module docstring)
        if verbose:
            rr = 1.0 + torch.cat(ranks).float()
            print("batch {} / {}: MR : {:.6f} |  MRR : {:.6f}".format(idx, n_batch, rr.mean().item(),
                                                                      (1.0 / rr).mean().item()))
    return torch.cat(ranks)


def calc_mrr(embedding, w, test_triplets, hits=[], eval_bz=100, all_batches=True, flow_log_prob=None,
             verbose=True):
    with torch.no_grad():
        # the reference validates with the model "on the CPU" (kgvae/link_predict.py:239-251): whatever side the caller's
        # tensors are on, the scorer runs where the embedding is
        test_triplets = test_triplets.to(embedding.device)
        w = w.to(embedding.device)
        if isinstance(flow_log_prob, torch.Tensor):
            flow_log_prob = flow_log_prob.to(embedding.device)
        s, r, o = test_triplets[:, 0], test_triplets[:, 1], test_triplets[:, 2]
        n = test_triplets.shape[0]
        ranks_s = perturb_and_get_rank(embedding, w, o, r, s, n, eval_bz, all_batches, flow_log_prob)
        ranks_o = perturb_and_get_rank(embedding, w, s, r, o, n, eval_bz, all_batches, flow_log_prob)
        ranks = torch.cat([ranks_s, ranks_o]) + 1
        mrr = torch.mean(1.0 / ranks.float())
        if verbose:
            print("MRR (raw): {:.6f}".format(mrr.item()))
            for hit in hits:
                print("Hits (raw) @ {}: {:.6f}".format(hit, torch.mean((ranks <= hit).float()).item()))
    return mrr.item()
