"""Multivariate Student-t log-density - mirror of reference distributions/student_t.py."""


def log_probability_per_samp(y, mu, sigma, v, name='student_t_logprob_per_samp'):
    """reference student_t.py:7-39,59-61: y (N,K,S,D), mu (K,D), sigma (K,D,D), v (K) -> (N,K,S).
    HIP kernel: vmp_student_t_logprob (K distinct scale matrices are factorised once, not N*K*S times)."""
    from ..models import _svae_ops
    return _svae_ops.student_t_logprob(y, mu, sigma, v)


def _logprob_full_scale(y, mu, sigma, v, name='student_t_logprob'):
    """reference student_t.py:7-39 (the function log_probability_per_samp forwards to)."""
    return log_probability_per_samp(y, mu, sigma, v, name)


def logprob_smm_mixture(y, mu, sigma, v, log_pi, name='student_t_logprob'):
    """reference student_t.py:42-56: y (N,D) -> (N,K) log S(y_n | mu_k, sigma_k, v_k) + log pi_k."""
    N, D = y.shape
    K = mu.shape[0]
    yk = y[:, None, None, :].expand(N, K, 1, D).contiguous()
    return log_probability_per_samp(yk, mu, sigma, v).reshape(N, K) + log_pi[None, :]
