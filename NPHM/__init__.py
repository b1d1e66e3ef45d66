"""Drop-in import shim: `from NPHM.models.EnsembledDeepSDF import FastEnsembleDeepSDFMirrored` etc.
resolve to the MI355X-native implementations in `nphm_amd` (same import paths as the reference's
`src/NPHM` package)."""
