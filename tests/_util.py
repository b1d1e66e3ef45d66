"""Shared helpers for the test-suite: seeded construction of the three fields with nphm_amd's own
modules (which reproduces the reference's seeded weights — proven against the SHA-256 stored in
the golden fixtures), numpy views of state_dicts for the oracle, golden loading."""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import nphm_amd  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
MINI = [-.55, -.5, -.95]
MAXI = [0.55, 0.75, 0.4]

# mean anchors (assets/anchors_39.npy of the reference, float32) travel inside the golden fixture
# as the predicted anchors minus nothing -> we store them explicitly in anchors_mean.npy


def anchors_mean():
    return np.load(os.path.join(GOLDEN, "anchors_mean_39.npy"))


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def state_hash(module) -> str:
    h = hashlib.sha256()
    for k, v in module.state_dict().items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def np_state(module):
    return {k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def build_identity(pos_mlp_dim=256, device="cpu"):
    anchors = torch.from_numpy(anchors_mean()).float().unsqueeze(0).unsqueeze(0)
    torch.manual_seed(0)
    net = nphm_amd.FastEnsembleDeepSDFMirrored(lat_dim_glob=64, lat_dim_loc=32, n_loc=39, n_symm_pairs=16,
                                               anchors=anchors.to(device), hidden_dim=200, n_layers=4,
                                               pos_mlp_dim=pos_mlp_dim)
    return net.to(device)


def build_deformation(device="cpu"):
    anchors = torch.from_numpy(anchors_mean()).float().unsqueeze(0).unsqueeze(0)
    torch.manual_seed(0)
    net = nphm_amd.DeformationNetwork(mode="compress", lat_dim_expr=200, lat_dim_id=32, lat_dim_glob_shape=64,
                                      lat_dim_loc_shape=32, n_loc=39, anchors=anchors.to(device), hidden_dim=512,
                                      nlayers=6, input_dim=3, out_dim=3)
    return net.to(device)


def build_npm(device="cpu"):
    torch.manual_seed(0)
    return nphm_amd.DeepSDF(lat_dim=512, hidden_dim=1024, nlayers=8, geometric_init=True).to(device)


def sample_latent(seed, scale=0.85):
    """Latent with the statistics of the reference's sampling (nphm_lat_mean/std shipped in the golden dir)."""
    st = np.load(os.path.join(GOLDEN, "nphm_lat_stats.npz"))
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(1344, generator=g) * torch.from_numpy(st["std"]) * scale + torch.from_numpy(st["mean"])).float()


def maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))
