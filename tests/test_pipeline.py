"""The reference's fitting script, end to end, through its OWN import paths (the `NPHM/` shim):
scripts/fitting/fitting_pointclouds.py:253-283 - fit the codes to the observations, extract the canonical mesh
from the identity field (get_logits -> mesh_from_logits), pose it with the deformation field (deform_mesh).
Small sizes; the GPU run is compared with the same script on the composite PyTorch tier."""
import numpy as np
import pytest
import torch
from scipy.spatial import cKDTree

import _util as U

# the import block of fitting_pointclouds.py:1-6 (what resolves here instead of in the reference's src/)
from NPHM.models.EnsembledDeepSDF import FastEnsembleDeepSDFMirrored          # noqa: F401
from NPHM.models.deepSDF import DeepSDF, DeformationNetwork                   # noqa: F401
from NPHM.models.fitting import inference_iterative_root_finding_joint
from NPHM.models.reconstruction import deform_mesh, get_logits
from NPHM.utils.reconstruction import create_grid_points_from_bounds, mesh_from_logits

LAMBDAS = {"surface": 2.0, "reg_expr": 0.01, "reg_global": 0.25, "reg_unobserved": 10, "reg_loc": 0.05, "symm_dist": 5.0}
SCHEDULE = {"lr": {200: 2, 400: 2, 600: 2, 800: 2}, "symm_dist": {200: 10, 500: 9999}, "reg_glob": {200: 3, 600: 10},
            "reg_loc": {500: 3, 600: 10}, "reg_expr": {600: 10}}


def _chamfer(a, b):
    return max(cKDTree(b).query(a)[0].mean(), cKDTree(a).query(b)[0].mean())


def _script(device, backend, res, n_steps, codes=None):
    """fitting_pointclouds.py:253-283 with its variable names; returns (codes, canonical mesh, posed mesh)"""
    g = U.golden("fitting_long")
    decoder_shape = U.build_identity(device=device)
    decoder_expr = U.build_deformation(device=device).eval()
    if backend is not None:
        decoder_shape.backend = decoder_expr.backend = backend
    all_obs = [torch.from_numpy(g[f"obs{i}"]).float().to(device) for i in range(3)]
    mini, maxi = [-.55, -.5, -.95], [0.55, 0.75, 0.4]
    grid_points = torch.from_numpy(create_grid_points_from_bounds(mini, maxi, res)).to(device, dtype=torch.float)
    grid_points = torch.reshape(grid_points, (1, len(grid_points), 3)).to(device)
    if codes is None:
        decoder_shape.train()
        torch.manual_seed(0)
        lat_reps_expr, lat_rep_shape, anchors = inference_iterative_root_finding_joint(
            decoder_shape, decoder_expr, all_obs, dict(LAMBDAS), schedule_cfg={k: dict(v) for k, v in SCHEDULE.items()},
            n_steps=n_steps, step_scale=1 / 4, verbose=False)
    else:
        lat_reps_expr, lat_rep_shape, anchors = [c.to(device) for c in codes]
    decoder_shape.eval()
    logits = get_logits(decoder_shape, lat_rep_shape, grid_points, nbatch_points=25000)
    mesh_can = mesh_from_logits(logits, mini, maxi, res)
    mesh = deform_mesh(mesh_can, decoder_expr, lat_reps_expr[0, ...].unsqueeze(0), anchors, lat_rep_shape=lat_rep_shape)
    codes = tuple(c.detach().cpu() for c in (lat_reps_expr, lat_rep_shape, anchors))
    return codes, mesh_can, mesh


def test_fitting_script_runs_through_the_shim_cpu():
    codes, mesh_can, mesh = _script("cpu", "composite", res=24, n_steps=8)
    assert codes[0].shape == (3, 1, 200) and codes[1].shape == (1, 1, 1344) and codes[2].shape == (1, 39, 3)
    v, f = np.asarray(mesh_can.vertices), np.asarray(mesh_can.faces)
    assert len(v) > 50 and f.max() < len(v) and np.asarray(mesh.vertices).shape == v.shape
    assert np.array_equal(np.asarray(mesh.faces), f) and np.isfinite(np.asarray(mesh.vertices)).all()


@pytest.mark.gpu
def test_fitting_script_gpu_matches_the_composite_tier():
    """same script on the HIP kernels (hipGraph-replayed fit, fused lattice extraction, GPU marching cubes, fused
    deform_mesh) and on the composite tier evaluated at the SAME fitted codes: canonical and posed meshes agree to
    the 1e-5 Chamfer bar of the north star"""
    dev = torch.device("cuda:0")
    codes, can_h, posed_h = _script(dev, None, res=64, n_steps=160)
    _, can_c, posed_c = _script(dev, "composite", res=64, n_steps=0, codes=codes)
    vh, vc = np.asarray(can_h.vertices), np.asarray(can_c.vertices)
    assert len(vh) > 1000 and abs(len(vh) - len(vc)) <= 0.002 * len(vc)
    assert _chamfer(vh, vc) < 1e-5
    assert _chamfer(np.asarray(posed_h.vertices), np.asarray(posed_c.vertices)) < 1e-5
