"""Fréchet distance between two feature sets: the statistic behind the reference's FVD
(src/utils/evaluator.py:118-179, itself the TF-GAN formulation).  The I3D feature extractor the reference feeds it with needs
weights that cannot be obtained offline, so only the statistic is provided; it is evaluation glue (plain torch linear algebra on
whatever device the features live on), not part of the hot path."""
import torch


def _sym_sqrt(mat, eps=1e-10):
    """U diag(sqrt(s)) V^T with singular values below eps left as they are (evaluator.py:119-122)."""
    u, s, vh = torch.linalg.svd(mat)
    si = torch.where(s < eps, s, torch.sqrt(s))
    return (u * si) @ vh


def trace_sqrt_product(sigma, sigma_v):
    """tr(sqrt(sigma^(1/2) sigma_v sigma^(1/2))) (evaluator.py:125-128)."""
    root = _sym_sqrt(sigma)
    return torch.trace(_sym_sqrt(root @ (sigma_v @ root)))


def covariance(x):
    """Unbiased covariance of observations in rows (evaluator.py:131-163 with rowvar=False)."""
    xc = x - x.mean(dim=0, keepdim=True)
    return (xc.t() @ xc) / (x.shape[0] - 1)


def frechet_distance(x1, x2):
    """|m1 - m2|^2 + tr(S1 + S2 - 2 sqrt(S1 S2)) over features flattened per sample (evaluator.py:166-179)."""
    x1 = torch.as_tensor(x1).flatten(start_dim=1)
    x2 = torch.as_tensor(x2).flatten(start_dim=1)
    m, m_w = x1.mean(dim=0), x2.mean(dim=0)
    sigma, sigma_w = covariance(x1), covariance(x2)
    trace = torch.trace(sigma + sigma_w) - 2.0 * trace_sqrt_product(sigma, sigma_w)
    return trace + torch.sum((m - m_w) ** 2)
