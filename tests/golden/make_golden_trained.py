"""Golden fixture on a TRAINED-LIKE checkpoint: the REFERENCE's FastEnsembleDeepSDFMirrored (PyTorch-CPU, fp32)
evaluated on a state_dict that was trained for 5 000 steps on analytic head-like surfaces
(tools/train_synthetic_heads.py, run on an MI355X with this repo's training tier; loss terms and trainer of
training.py:110-135).  No released checkpoint can be fetched here (README.md:151: Google-Drive link), so this is the
fixture that carries trained sharpness: weights up to 1.25 (seeded init: 0.07), surface fitted to 5e-4 mean |sdf|,
|grad| - 1 to 4e-3, predicted anchors on the subjects' surfaces.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_trained.py [gpurun_out/r3a/trained_heads.npz]

Writes
  trained_state.npz   the checkpoint itself: the decoder's state_dict (strict-loadable into the reference's and
                      into this repo's module) and the 64 trained latent codes (fp32)
  trained.npz         reference outputs: per code in CODES the predicted anchors, 4 608 lattice voxels of the 256^3
                      fitting box stratified by blend regime (near an anchor / mid-field / sum of weights < 1e-6),
                      2 048 near-surface points; a 40^3 lattice through the reference's get_logits in eval mode
                      (chunk overwrite voxels included); SHA-256 of the state_dict"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, os.path.dirname(HERE))
for missing in ("trimesh", "mcubes"):          # imported but unused by get_logits
    sys.modules.setdefault(missing, types.ModuleType(missing))

from NPHM.models.EnsembledDeepSDF import FastEnsembleDeepSDFMirrored  # noqa: E402
from NPHM.models.reconstruction import get_logits                      # noqa: E402
from NPHM.utils.reconstruction import create_grid_points_from_bounds   # noqa: E402
import _util as U                                                       # noqa: E402  (stratified_voxels only)

ASSETS = os.path.join(REF, "assets")
MINI = [-.55, -.5, -.95]
MAXI = [0.55, 0.75, 0.4]
CODES = (0, 1, 2, 17)
RES = 256
LATTICE_RES, LATTICE_CHUNK = 40, 9000


def state_hash(sd) -> str:
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def main():
    sys.path.insert(0, HERE)
    import _sources
    ck, from_scratch = _sources.load("heads")       # the training run's output, or the committed trained_state.npz
    sd = {k[3:]: torch.from_numpy(ck[k]) for k in ck if k.startswith("sd.")}
    codes = torch.from_numpy(ck["codes"]).float()
    anchors = torch.from_numpy(np.load(os.path.join(ASSETS, "anchors_39.npy"))).float().unsqueeze(0).unsqueeze(0)
    net = FastEnsembleDeepSDFMirrored(lat_dim_glob=64, lat_dim_loc=32, n_loc=39, n_symm_pairs=16, anchors=anchors,
                                      hidden_dim=200, n_layers=4, pos_mlp_dim=256)
    net.load_state_dict(sd, strict=True)                      # the checkpoint layout of the reference
    if from_scratch:
        np.savez(os.path.join(HERE, "trained_state.npz"), **{"sd." + k: v.numpy() for k, v in net.state_dict().items()},
                 codes=codes.numpy(), trace=ck["trace"], meta=ck["meta"], subject_anchors=ck["anchors"])

    out = {"state_sha256": np.array(state_hash(net.state_dict())), "codes_used": np.array(CODES)}
    axes = [np.linspace(MINI[i], MAXI[i], RES).astype(np.float32) for i in range(3)]      # utils/reconstruction.py:10-12 -> float32
    gen = torch.Generator().manual_seed(0)
    surf = torch.from_numpy(ck["anchors"]).float()           # [S,39,3] anchors on the subjects' surfaces (near-surface seeds)
    net.train()                                               # train mode: no last-point overwrite (EnsembledDeepSDF.py:260)
    with torch.no_grad():
        for c in CODES:
            lat = codes[c][None, None]
            _, a_pred = net(torch.zeros(1, 1, 3), lat, None)
            keep = U.stratified_voxels(axes, a_pred[0].numpy(), 1536, seed=100 + c)
            pts = np.stack([axes[0][keep // (RES * RES)], axes[1][(keep // RES) % RES], axes[2][keep % RES]], -1)
            sdf, _ = net(torch.from_numpy(pts)[None], lat.repeat(1, len(keep), 1), None)
            # near-surface points: around the subject's anchors (which lie on its analytic surface) and between them
            i = torch.randint(0, 39, (2048,), generator=gen)
            j = torch.randint(0, 39, (2048,), generator=gen)
            t = torch.rand(2048, 1, generator=gen) * 0.3
            near = surf[c][i] * (1 - t) + surf[c][j] * t + 0.01 * torch.randn(2048, 3, generator=gen)
            sdf_near, _ = net(near[None], lat.repeat(1, 2048, 1), None)
            out[f"c{c}_anchors"] = a_pred[0].numpy()
            out[f"c{c}_voxels"] = keep.astype(np.int64)
            out[f"c{c}_sdf_voxels"] = sdf.reshape(-1).numpy()
            out[f"c{c}_near"] = near.numpy()
            out[f"c{c}_sdf_near"] = sdf_near.reshape(-1).numpy()
            print(f"code {c}: |sdf| voxels max {float(sdf.abs().max()):.3f}, near-surface mean |sdf| {float(sdf_near.abs().mean()):.4f}, "
                  f"max {float(sdf_near.abs().max()):.3f}")
    # eval-mode lattice through the reference's own get_logits (chunk-overwrite voxels included)
    net.eval()
    grid = torch.from_numpy(create_grid_points_from_bounds(MINI, MAXI, LATTICE_RES)).float()[None]
    vol = get_logits(net, codes[CODES[0]], grid, nbatch_points=LATTICE_CHUNK)
    out["lattice_res"], out["lattice_chunk"] = np.array(LATTICE_RES), np.array(LATTICE_CHUNK)
    out["lattice_volume"] = np.asarray(vol, np.float32)
    print(f"lattice {LATTICE_RES}^3: sdf range {vol.min():.3f} .. {vol.max():.3f}, inside fraction {(vol < 0).mean():.3f}")
    np.savez_compressed(os.path.join(HERE, "trained.npz"), **out)
    print("wrote trained_state.npz (%.1f MiB), trained.npz (%.2f MiB)" % (
        os.path.getsize(os.path.join(HERE, "trained_state.npz")) / 2 ** 20, os.path.getsize(os.path.join(HERE, "trained.npz")) / 2 ** 20))


if __name__ == "__main__":
    main()
