"""Import-path shim for the reference's src/NPHM/models/reconstruction.py."""
from nphm_amd.reconstruction import deform_mesh, get_logits, get_logits_backward  # noqa: F401
