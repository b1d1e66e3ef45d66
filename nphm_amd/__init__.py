"""nphm_amd — MI355X-native (gfx950) implementation of NPHM's batched neural-field evaluation
hot path behind the reference's own nn.Module API.  See DESIGN.md."""
from .ensembled_deepsdf import (EnsembledDeepSDF, EnsembledLinear, FastEnsembleDeepSDFMirrored,
                                sample_point_feature)
from .deepsdf import DeepSDF, DeformationNetwork
from .reconstruction import (create_grid_points_from_bounds, deform_mesh, get_logits,
                             get_logits_backward, grid_axes, marching_cubes, mesh_from_logits)
from ._lib import NphmAmdError
from .numerics import calibrate_numerics, validate_numerics, validate_training_numerics

__all__ = ["EnsembledDeepSDF", "EnsembledLinear", "FastEnsembleDeepSDFMirrored", "sample_point_feature",
           "DeepSDF", "DeformationNetwork", "create_grid_points_from_bounds", "deform_mesh", "get_logits",
           "get_logits_backward", "grid_axes", "marching_cubes", "mesh_from_logits", "NphmAmdError", "calibrate_numerics", "validate_numerics",
           "validate_training_numerics"]
