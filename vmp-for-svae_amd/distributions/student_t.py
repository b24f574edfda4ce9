"""Multivariate Student-t log-density - mirror of reference distributions/student_t.py."""


def log_probability_per_samp(y, mu, sigma, v, name='student_t_logprob_per_samp'):
    """reference student_t.py:7-39,59-61: y (N,K,S,D), mu (K,D), sigma (K,D,D), v (K) -> (N,K,S).
    HIP kernel: vmp_student_t_logprob (K distinct scale matrices are factorised once, not N*K*S times)."""
    from ..models import _svae_ops
    return _svae_ops.student_t_logprob(y, mu, sigma, v)
