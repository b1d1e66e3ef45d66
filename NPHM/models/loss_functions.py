"""Import-path shim for the reference's src/NPHM/models/loss_functions.py."""
from nphm_amd.loss_functions import (actual_compute_loss, compute_loss, compute_loss_corresp_forward,  # noqa: F401
                                     loss_joint)
