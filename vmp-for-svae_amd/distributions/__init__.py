"""Mirror of the reference's ``distributions`` package (gaussian, niw, dirichlet, student_t)."""
from . import dirichlet, gaussian, niw, student_t  # noqa: F401
