"""Gaussian natural-parameter algebra - mirror of reference distributions/gaussian.py.

``standard_to_natural`` / ``natural_to_standard`` are K-sized (torch, autograd-capable).  The two
N-sized log-densities are implemented by HIP kernels (see models/svae.py for the fused forms the
training step uses).
"""
import torch

from .. import _klinalg


def standard_to_natural(mu, sigma, name='gauss_to_nat'):
    """reference gaussian.py:11-19: eta2 = -1/2 Sigma^-1, eta1 = Sigma^-1 mu."""
    prec = _klinalg.inv(sigma)
    eta1 = torch.einsum('...ij,...j->...i', prec, mu)
    return eta1, -0.5 * prec


def natural_to_standard(eta1, eta2, name='gauss_to_stndrd'):
    """reference gaussian.py:22-27: Sigma = (-2 eta2)^-1, mu = Sigma eta1."""
    sigma = _klinalg.inv(-2.0 * eta2)
    return torch.einsum('...ij,...j->...i', sigma, eta1), sigma


def log_probability_nat(x, eta1, eta2, weights=None):
    """reference gaussian.py:30-71 (normalised over k).  HIP kernel: vmp_gauss_logprob_nat."""
    from ..models import _svae_ops
    return _svae_ops.gauss_logprob_nat(x, eta1, eta2, weights)


def log_probability_nat_per_samp(x_samps, eta1, eta2):
    """reference gaussian.py:74-105, (N,K,S,D) -> (N,K,S).  HIP kernel: vmp_gauss_logprob_per_samp."""
    from ..models import _svae_ops
    return _svae_ops.gauss_logprob_per_samp(x_samps, eta1, eta2)
