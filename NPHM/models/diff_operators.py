"""Import-path shim for the reference's src/NPHM/models/diff_operators.py (jac, gradient: the two
operators the hot-path callers use)."""
from nphm_amd.diff_operators import gradient, jac  # noqa: F401
