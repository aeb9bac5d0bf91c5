"""``normalize_tensor`` of the reference's lpips/common.py:12-14 (eps outside the sqrt), for host-side callers."""
import torch


def normalize_tensor(in_feat, eps=1e-10):
    return in_feat / (torch.sqrt(torch.sum(in_feat ** 2, dim=1, keepdim=True)) + eps)
