"""Import-path shim for the reference's src/NPHM/models/EnsembledDeepSDF.py."""
from nphm_amd.ensembled_deepsdf import (EnsembledDeepSDF, EnsembledLinear,  # noqa: F401
                                        FastEnsembleDeepSDFMirrored, sample_point_feature)
