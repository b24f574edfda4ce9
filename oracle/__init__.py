"""CPU oracle for the VMP hot path of emtiyaz/vmp-for-svae  --  TEST INFRASTRUCTURE, NOT PRODUCT.

What this is
    A torch-CPU, dtype-generic (fp32 / fp64) op-for-op restatement of the reference's hot path:
    the same einsum contractions, LU solves where the reference calls ``tf.matrix_solve`` /
    ``tf.matrix_inverse``, Cholesky where it calls ``tf.cholesky``, the same quirks.
    Every function cites the reference file:line it follows (paths relative to the reference root).

Who may use it
    Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` - as the
    checker / the timed CPU baseline, never as something shipped.  The product package
    (``vmp-for-svae_amd``) never imports it and raises if its HIP library is missing.

How it is pinned  (parity PINNED by reference execution, not by reference tests - the reference has none)
    ``tests/golden/*.npz`` hold inputs and outputs obtained by running the reference's OWN functions
    (imported from /root/reference on an eager TF-1.3 API stand-in, ``tests/golden/make_fixtures.py``)
    in fp64 ("truth") and fp32.  ``tests/test_oracle_golden.py`` checks every function here against
    them (fp64: <=1e-10 rel; fp32: within the reference's own fp32-vs-fp64 noise).  Independent
    known-answer checks (scipy multivariate normal / t, digamma, row sums) are in the same test file.
    The dense primitives underneath the reference (Eigen LU / LLT, cephes digamma/lgamma inside
    TensorFlow 1.3.0, pinned by the reference's environment.yml:33) are third-party and absent from
    /root/reference; they are replaced by LAPACK (torch.linalg) and torch.special here, i.e. the
    published algorithms (partial-pivot LU, Cholesky), and the fp64 evaluation is the adopted truth.
"""
from . import dists, metrics, mixtures, nets, philox, svae_ref, train_ref  # noqa: F401
