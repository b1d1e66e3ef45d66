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


def stratified_voxels(axes, anchors, n_each, seed=0, exclude=None, pool=400_000):
    """Flat indices of 3 * n_each lattice voxels, stratified by where the NPHM blend puts them: near an anchor
    (< 0.05 from the closest one: several members with sizeable weights), mid-field (sum of blend weights >= 1e-6,
    not near) and far field (sum of blend weights < 1e-6: the epsilon of the normaliser dominates,
    EnsembledDeepSDF.py:147).  ``anchors`` [39,3] are the PREDICTED anchors of the latent at hand."""
    rng = np.random.default_rng(seed)
    ax, ay, az = [np.asarray(a) for a in axes]
    ny, nz = len(ay), len(az)
    total = len(ax) * ny * nz
    cand = rng.choice(total, min(pool, total), replace=False)
    if exclude is not None:
        cand = np.setdiff1d(cand, exclude)
    q = np.stack([ax[cand // (ny * nz)], ay[(cand // nz) % ny], az[cand % nz]], -1).astype(np.float32)
    d = np.linalg.norm(q[:, None] - np.asarray(anchors, np.float32)[None], axis=-1) + np.float32(1e-5)
    s_w = np.exp(-(d * d) / np.float32(0.01)).sum(-1) + np.exp(np.float32(-20.0))
    near = d.min(-1) < 0.05
    far = s_w < 1e-6
    mid = ~near & ~far
    picks = []
    for m in (near, mid, far):
        ids = cand[m]
        assert len(ids) >= n_each, (int(near.sum()), int(mid.sum()), int(far.sum()))
        picks.append(rng.choice(ids, n_each, replace=False))
    return np.concatenate(picks)


def trained_checkpoint():
    """(state_dict as torch tensors, codes [64,1344]) of the trained-like checkpoint (tests/golden/trained_state.npz:
    5 000 steps on analytic head-like surfaces, tools/train_synthetic_heads.py; the reference outputs on it are in
    tests/golden/trained.npz, make_golden_trained.py)."""
    ck = np.load(os.path.join(GOLDEN, "trained_state.npz"))
    sd = {k[3:]: torch.from_numpy(ck[k]) for k in ck.files if k.startswith("sd.")}
    return sd, torch.from_numpy(ck["codes"]).float()


def build_trained_identity(device="cpu"):
    net = build_identity(device=device)
    sd, codes = trained_checkpoint()
    net.load_state_dict(sd, strict=True)
    return net, codes.to(device)


def build_trained_deformation(device="cpu"):
    """(deformation net with the trained-like checkpoint tests/golden/trained_def_state.npz - the reference's module trained on
    analytic expression warps, tools/train_synthetic_expressions.py - , expression codes [P,200], (subject, expression) pairs)"""
    ck = np.load(os.path.join(GOLDEN, "trained_def_state.npz"))
    net = build_deformation(device=device)
    net.load_state_dict({k[3:]: torch.from_numpy(ck[k]) for k in ck.files if k.startswith("sd.")}, strict=True)
    return net.eval(), torch.from_numpy(ck["z_ex"]).float().to(device), [tuple(int(v) for v in p) for p in ck["pairs"]]


def build_trained_npm(device="cpu"):
    """(NPM DeepSDF with the trained-like checkpoint tests/golden/trained_npm_state.npz - the reference's module trained on
    analytic head surfaces, tools/train_synthetic_npm.py - , the codes [4,512] of the fixture)"""
    ck = np.load(os.path.join(GOLDEN, "trained_npm_state.npz"))
    net = build_npm(device=device)
    net.load_state_dict({k[3:]: torch.from_numpy(ck[k]) for k in ck.files if k.startswith("sd.")}, strict=True)
    return net.eval(), torch.from_numpy(ck["codes"]).float().to(device)


class start_codes:
    """The fitting loops (the reference's and nphm_amd.fitting's mirror) create their codes with torch.zeros([n_obs, 1, 200])
    and torch.zeros([1, 1, lat_dim]): inside this context those two calls return the given tensors instead, so that a loop's
    first step runs AT these codes (tests/golden/make_golden_fitting_trained.py holds the generator's twin)."""

    def __init__(self, z_expr, z_shape):
        self.init = {tuple(z_expr.shape): z_expr, tuple(z_shape.shape): z_shape}

    def __enter__(self):
        import torch
        self.torch, self.zeros = torch, torch.zeros
        init, zeros = self.init, self.zeros

        def patched(*size, **kw):
            shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (list, tuple)) else tuple(size)
            if shape in init:
                t = init[shape].detach().clone().float()
                return t.to(kw["device"]) if kw.get("device") is not None else t
            return zeros(*size, **kw)
        torch.zeros = patched
        return self

    def __exit__(self, *exc):
        self.torch.zeros = self.zeros
