"""CPU tests: the PyTorch op-for-op restatement of the reference forward passes (oracle/torch_reference.py,
the cpu_baseline of bench.py) against the fixtures produced by the reference itself.  Same operations in
the same order on the same BLAS -> agreement at round-off level (asserted: 1e-6)."""
import numpy as np
import torch

import _util as U
from oracle import torch_reference as T

TOL = 1e-6


def _sd(net):
    return {k: v.detach().clone() for k, v in net.state_dict().items()}


def test_identity_forward_matches_reference_fixtures():
    g = U.golden("nphm_identity")
    sd, amean = _sd(U.build_identity()), torch.from_numpy(U.anchors_mean()).float()
    xyz = torch.from_numpy(g["xyz"])[None]
    lat = torch.from_numpy(g["lat"])[None, None]
    with torch.no_grad():
        sdf_e, anc = T.nphm_identity_forward(sd, amean, xyz, lat.repeat(1, xyz.shape[1], 1), training=False)
        sdf_t, _ = T.nphm_identity_forward(sd, amean, xyz, lat, training=True)
        sdf_b, anc_b = T.nphm_identity_forward(sd, amean, torch.from_numpy(g["b2_xyz"]),
                                               torch.from_numpy(g["b2_lat"])[:, None], training=False)
        sdf_p, _ = T.nphm_identity_forward(sd, amean, torch.from_numpy(g["pp_xyz"]), torch.from_numpy(g["pp_lat"]),
                                           training=True)
    assert U.maxdiff(sdf_e, g["sdf_eval"]) < TOL and U.maxdiff(anc, g["anchors"]) < TOL
    assert U.maxdiff(sdf_t, g["sdf_train"]) < TOL
    assert U.maxdiff(sdf_b, g["b2_sdf_eval"]) < TOL and U.maxdiff(anc_b, g["b2_anchors"]) < TOL
    assert U.maxdiff(sdf_p, g["pp_sdf_train"]) < TOL


def test_get_logits_chunk_overwrite_matches_reference_fixture():
    from nphm_amd import reconstruction as R
    g = U.golden("nphm_identity")
    sd, amean = _sd(U.build_identity()), torch.from_numpy(U.anchors_mean()).float()
    res, chunk = int(g["grid_res"]), int(g["grid_chunk"])
    grid = torch.from_numpy(R.create_grid_points_from_bounds(U.MINI, U.MAXI, res)).float()[None]
    enc = torch.from_numpy(g["lat"])[None, None]
    vol = T.get_logits(lambda p, l: T.nphm_identity_forward(sd, amean, p, l, training=False)[0], enc, grid, chunk)
    assert vol.shape == (res ** 3,) and U.maxdiff(vol, g["grid_logits_eval"]) < TOL


def test_npm_forward_matches_reference_fixture():
    g = U.golden("npm")
    sd = _sd(U.build_npm())
    xyz = torch.from_numpy(g["xyz"])
    lat = torch.from_numpy(g["lat"])[None, None].repeat(1, xyz.shape[1], 1)
    with torch.no_grad():
        out = T.deepsdf_forward(sd, "", xyz, lat, nlayers=8)
    assert U.maxdiff(out, g["sdf"]) < 2e-6
