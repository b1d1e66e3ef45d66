"""Import-path shim for the reference's src/NPHM/utils/reconstruction.py."""
from nphm_amd.reconstruction import create_grid_points_from_bounds, mesh_from_logits  # noqa: F401
