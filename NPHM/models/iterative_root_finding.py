"""Import-path shim for the reference's src/NPHM/models/iterative_root_finding.py."""
from nphm_amd.iterative_root_finding import broyden, nabla, search  # noqa: F401
from nphm_amd.diff_operators import gradient, jac  # noqa: F401
