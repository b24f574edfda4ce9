"""Mirror of the reference's ``models`` package: gmm, smm (pure mixture VMP), svae, vae."""
from . import gmm, smm, vae, svae  # noqa: F401
