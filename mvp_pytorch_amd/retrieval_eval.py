"""Device-side coarse stage of the two-stage retrieval evaluation.

`compute_ranks_coarse` (oscar/run_retrieval.py:481-522) pulls the [n_img, n_cap] similarity matrix
to the host, argsorts every row and every column in numpy and walks the sorted lists in Python
(1 000 x 5 000 on COCO-1k: 6 000 argsorts + Python loops).  Here the matrix stays in HBM:

  * rank of the first ground-truth item = number of entries that beat the best ground-truth
    similarity (the position at which the reference's descending walk first meets a ground-truth
    index);
  * candidate lists = torch.topk along the row / column (sorted, descending) — the first
    num_captions_per_img_val / num_images_per_cap_val entries of the reference's argsort[::-1].

Integer outputs; equal to the reference whenever the compared similarities are distinct (ties are
ordered by numpy's unstable argsort there, i.e. unspecified).
"""
import torch


@torch.no_grad()
def coarse_ranks(similarities, num_captions_per_img, num_captions_per_img_val, num_images_per_cap_val):
    """similarities: [n_img, n_cap] (caption j belongs to image j // num_captions_per_img).
    Returns dict of int64 device tensors:
      i2t_ranks [n_img], t2i_ranks [n_cap]           — run_retrieval.py:488-495, 508-516
      i2t_topk  [n_img, k_c] caption indices          — :496-503 (the reference stores them as
                                                        (img_key[ind // c], ind % c) pairs)
      t2i_topk  [n_cap, k_i] image indices            — :517-520
    """
    sim = similarities
    if sim.dim() != 2:
        raise ValueError("similarities must be [n_img, n_cap]")
    n_img, n_cap = sim.shape
    c = int(num_captions_per_img)
    if n_cap != n_img * c:
        raise ValueError("n_cap=%d is not n_img=%d x num_captions_per_img=%d" % (n_cap, n_img, c))
    gt_best = sim.view(n_img, n_img, c)[torch.arange(n_img, device=sim.device), torch.arange(n_img, device=sim.device)].max(dim=1)[0]
    i2t_ranks = (sim > gt_best[:, None]).sum(dim=1)
    cap_img = torch.arange(n_cap, device=sim.device) // c
    gt_col = sim[cap_img, torch.arange(n_cap, device=sim.device)]
    t2i_ranks = (sim > gt_col[None, :]).sum(dim=0)
    k_c = min(int(num_captions_per_img_val), n_cap)
    k_i = min(int(num_images_per_cap_val), n_img)
    i2t_topk = sim.topk(k_c, dim=1, largest=True, sorted=True)[1]
    t2i_topk = sim.t().topk(k_i, dim=1, largest=True, sorted=True)[1]
    return dict(i2t_ranks=i2t_ranks, t2i_ranks=t2i_ranks, i2t_topk=i2t_topk, t2i_topk=t2i_topk)


@torch.no_grad()
def recall_at(ranks, ks=(1, 5, 10)):
    """R@k of a rank vector (run_retrieval.py:853-860: `(ranks < k).mean()`), as python floats."""
    r = ranks.to(torch.float32)
    return [float((r < k).float().mean()) for k in ks]


@torch.no_grad()
def rerank_ranks(scores, candidates, gt_mask):
    """Second-stage ranks (compute_ranks / compute_ranks_t2i, run_retrieval.py:424-479): for every
    query, candidates sorted by descending re-ranker score; rank = position of the first ground-truth
    candidate, or the list length when none is among the candidates.
    scores [n_q, k] f32, candidates [n_q, k] (unused here, kept for symmetry), gt_mask [n_q, k] bool."""
    del candidates
    n_q, k = scores.shape
    masked = torch.where(gt_mask, scores, torch.full_like(scores, float("-inf")))
    best = masked.max(dim=1)[0]
    rank = (scores > best[:, None]).sum(dim=1)
    return torch.where(gt_mask.any(dim=1), rank, torch.full_like(rank, k))
