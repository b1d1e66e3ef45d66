"""Import-path shim for the reference's src/NPHM/models/deepSDF.py."""
from nphm_amd.deepsdf import DeepSDF, DeformationNetwork, sample_point_feature  # noqa: F401
