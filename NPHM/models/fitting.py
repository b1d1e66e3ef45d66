"""Import-path shim for the reference's src/NPHM/models/fitting.py."""
from nphm_amd.fitting import inference_identity_space, inference_iterative_root_finding_joint  # noqa: F401
