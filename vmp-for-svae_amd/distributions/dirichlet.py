"""Dirichlet parameter algebra - mirror of reference distributions/dirichlet.py (K-sized, torch)."""
import torch


def expected_log_pi(dir_standard_param):
    """reference dirichlet.py:8-12: psi(alpha_k) - psi(sum alpha)."""
    a = dir_standard_param
    return torch.special.digamma(a) - torch.special.digamma(a.sum(dim=-1, keepdim=True))


def standard_to_natural(alpha):
    """reference dirichlet.py:15-17."""
    return alpha - 1


def natural_to_standard(alpha_nat):
    """reference dirichlet.py:20-22."""
    return alpha_nat + 1
