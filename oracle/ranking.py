"""Oracle restatement of the raw-MRR evaluator  --  TEST INFRASTRUCTURE (pinned by golden vectors).

Follows /root/reference/kgvae/utils.py:
  sort_and_rank         :180-184   full descending sort, position of the target
  perturb_and_get_rank  :187-221   (D,E,1)x(D,1,V) bmm -> sum over D -> + flow_log_prob -> sigmoid -> rank
  calc_mrr              :293-314   subject pass (o, r -> s) then object pass (s, r -> o); 1-indexed
Console prints of the reference are dropped; the returned numbers are the same.
"""
import torch


def sort_and_rank(score, target):
    _, order = torch.sort(score, dim=1, descending=True)
    hit = torch.nonzero(order == target.view(-1, 1))
    return hit[:, 1].view(-1)


def perturb_and_get_rank(embedding, w, a, r, b, test_size, batch_size=100, all_batches=True,
                         flow_log_prob=None):
    n_batch = (test_size + batch_size - 1) // batch_size
    if all_batches is False:
        n_batch = 1
    ranks = []
    for bi in range(n_batch):
        lo, hi = bi * batch_size, min(test_size, (bi + 1) * batch_size)
        emb_ar = (embedding[a[lo:hi]] * w[r[lo:hi]]).transpose(0, 1).unsqueeze(2)   # D x E x 1
        emb_c = embedding.transpose(0, 1).unsqueeze(1)                              # D x 1 x V
        score = torch.sum(torch.bmm(emb_ar, emb_c), dim=0)                          # E x V
        score = torch.sigmoid(score + flow_log_prob)
        ranks.append(sort_and_rank(score, b[lo:hi]))
    return torch.cat(ranks)


def calc_mrr(embedding, w, test_triplets, hits=(), eval_bz=100, all_batches=True, flow_log_prob=None):
    """Returns (mrr, {hit: fraction}, ranks 1-indexed)."""
    with torch.no_grad():
        s, r, o = test_triplets[:, 0], test_triplets[:, 1], test_triplets[:, 2]
        n = test_triplets.shape[0]
        ranks_s = perturb_and_get_rank(embedding, w, o, r, s, n, eval_bz, all_batches, flow_log_prob)
        ranks_o = perturb_and_get_rank(embedding, w, s, r, o, n, eval_bz, all_batches, flow_log_prob)
        ranks = torch.cat([ranks_s, ranks_o]) + 1
        mrr = torch.mean(1.0 / ranks.float()).item()
        hit_frac = {h: torch.mean((ranks <= h).float()).item() for h in hits}
    return mrr, hit_frac, ranks
