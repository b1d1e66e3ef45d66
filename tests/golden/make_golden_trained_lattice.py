"""Lattice-sized golden fixture on the TRAINED-LIKE checkpoints: the REFERENCE's own modules (PyTorch-CPU, fp32) evaluated on
the 64^3 lattice of the reference box - 262 144 points, the size from which this repo's `numerics = "auto"` switches on its
calibrated tiers (DeepSDF.two_pass_min_points = 262 144, FastEnsembleDeepSDFMirrored.AUTO_MIN_POINTS = 65 536).  The small
fixtures (trained.npz, trained_def.npz, trained_npm.npz: <= 8 000 points) never reach those tiers; this one pins them to the
reference directly (verdict round 5, item 3).  Run in the build container only (needs /root/reference; ~3 min on 8 cores):

    python tests/golden/make_golden_trained_lattice.py

Writes trained_lattice.npz:
  def_offsets        [262144, 3]  DeformationNetwork.forward (deepSDF.py:184-239) of pair 0 (subject 0, expression 3), anchors of
                                  the trained identity decoder, in chunks of 25 000 points
  npm_logits         [262144]     get_logits (models/reconstruction.py:6-25) of the trained NPM DeepSDF, code 3, chunk 25 000
  identity_logits    [262144]     get_logits of the trained identity decoder, subject 0, eval mode (chunk overwrite), chunk 25 000
  two_stage_logits   [262144]     get_logits_backward (models/reconstruction.py:28-56) of the trained pair, chunk 25 000"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, HERE)
for missing in ("trimesh", "mcubes"):
    sys.modules.setdefault(missing, types.ModuleType(missing))

import make_golden as G                                                   # noqa: E402
from NPHM.models.deepSDF import DeepSDF, DeformationNetwork               # noqa: E402
from NPHM.models.EnsembledDeepSDF import FastEnsembleDeepSDFMirrored     # noqa: E402
from NPHM.models.reconstruction import get_logits, get_logits_backward    # noqa: E402
from NPHM.utils.reconstruction import create_grid_points_from_bounds      # noqa: E402

RES, CHUNK = 64, 25000
PAIR = (0, 3)
NPM_CODE = 3


def _sd(npz, keys=None):
    return {k[3:]: torch.from_numpy(npz[k]) for k in (keys or npz.files) if k.startswith("sd.")}


def main():
    mean_anchors = torch.from_numpy(np.load(os.path.join(G.ASSETS, "anchors_39.npy"))).float()[None, None]
    ick = np.load(os.path.join(HERE, "trained_state.npz"))
    inet = FastEnsembleDeepSDFMirrored(lat_dim_glob=64, lat_dim_loc=32, n_loc=39, n_symm_pairs=16, anchors=mean_anchors,
                                       hidden_dim=200, n_layers=4, pos_mlp_dim=256)
    inet.load_state_dict(_sd(ick), strict=True)
    inet.eval()
    dck = np.load(os.path.join(HERE, "trained_def_state.npz"))
    dnet = DeformationNetwork(mode="compress", lat_dim_expr=200, lat_dim_id=32, lat_dim_glob_shape=64, lat_dim_loc_shape=32,
                              n_loc=39, anchors=mean_anchors, hidden_dim=512, nlayers=6, input_dim=3, out_dim=3)
    dnet.load_state_dict(_sd(dck), strict=True)
    dnet.eval()
    nck = np.load(os.path.join(HERE, "trained_npm_state.npz"))
    nnet = DeepSDF(lat_dim=512, hidden_dim=1024, nlayers=8, geometric_init=True)
    nnet.load_state_dict(_sd(nck), strict=True)
    nnet.eval()
    pairs = [tuple(int(v) for v in p) for p in dck["pairs"]]
    z_ex = torch.from_numpy(dck["z_ex"][pairs.index(PAIR)]).float()
    lat_id = torch.from_numpy(ick["codes"]).float()[PAIR[0]]
    npm_code = torch.from_numpy(nck["codes"][[int(i) for i in nck["code_ids"]].index(NPM_CODE)]).float()
    grid = torch.from_numpy(create_grid_points_from_bounds(G.MINI, G.MAXI, RES)).float()[None]
    out = {"res": np.int64(RES), "chunk": np.int64(CHUNK), "pair": np.asarray(PAIR), "npm_code": np.int64(NPM_CODE),
           "identity_sha256": np.array(G.state_hash(inet)), "deformation_sha256": np.array(G.state_hash(dnet)),
           "npm_sha256": np.array(G.state_hash(nnet))}
    with torch.no_grad():
        _, anc = inet(torch.zeros(1, 1, 3), lat_id[None, None], None)                    # predicted anchors [1,39,3]
        out["anchors"] = anc.numpy()
        lat_all = torch.cat([lat_id, z_ex])
        offs = []
        for pts in torch.split(grid, CHUNK, dim=1):
            off, _ = dnet(pts, lat_all[None, None].repeat(1, pts.shape[1], 1), anc)
            offs.append(off[0])
        out["def_offsets"] = torch.cat(offs).numpy()
        print("deformation: |offset| max", float(np.abs(out["def_offsets"]).max()), flush=True)
        out["npm_logits"] = get_logits(nnet, npm_code, grid, nbatch_points=CHUNK)
        print("npm: range", float(out["npm_logits"].min()), float(out["npm_logits"].max()), flush=True)
        out["identity_logits"] = get_logits(inet, lat_id, grid, nbatch_points=CHUNK)
        print("identity: range", float(out["identity_logits"].min()), float(out["identity_logits"].max()), flush=True)

        class _Expr(torch.nn.Module):                                                       # see make_golden.py: anchors supplied
            def forward(self, p, l, a):
                return dnet(p, l, anc)
        out["two_stage_logits"] = get_logits_backward(inet, _Expr(), lat_id, lat_all, grid, nbatch_points=CHUNK)
        print("two-stage: range", float(out["two_stage_logits"].min()), float(out["two_stage_logits"].max()), flush=True)
    np.savez_compressed(os.path.join(HERE, "trained_lattice.npz"), **out)
    print("trained_lattice.npz", {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
