"""Import-path shim for the reference's src/NPHM/models/loss_functions.py (identity-decoder loss)."""
from nphm_amd.loss_functions import actual_compute_loss, compute_loss  # noqa: F401
