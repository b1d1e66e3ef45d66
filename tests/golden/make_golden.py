"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE (PyTorch-CPU, fp32).

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference ships no golden vectors; these fixtures pin the oracle
(oracle/nphm_oracle.py) and the HIP path to outputs of the reference's own modules.  Weights are
NOT stored: they are PyTorch's seeded default init (torch.manual_seed(0), construction order of the
reference), which nphm_amd's modules reproduce bit-for-bit; each fixture stores a SHA-256 of the
reference state_dict so the tests can prove the regenerated weights are the reference's.
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "src"))
for missing in ("trimesh", "mcubes"):          # imported but unused by get_logits*
    sys.modules.setdefault(missing, types.ModuleType(missing))

from NPHM.models.EnsembledDeepSDF import FastEnsembleDeepSDFMirrored  # noqa: E402
from NPHM.models.deepSDF import DeepSDF, DeformationNetwork           # noqa: E402
from NPHM.models.reconstruction import get_logits, get_logits_backward  # noqa: E402
from NPHM.utils.reconstruction import create_grid_points_from_bounds   # noqa: E402

ASSETS = os.path.join(REF, "assets")
MINI = [-.55, -.5, -.95]
MAXI = [0.55, 0.75, 0.4]


def state_hash(module) -> str:
    h = hashlib.sha256()
    for k, v in module.state_dict().items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def build_identity(pos_mlp_dim=256):
    anchors = torch.from_numpy(np.load(os.path.join(ASSETS, "anchors_39.npy"))).float().unsqueeze(0).unsqueeze(0)
    torch.manual_seed(0)
    net = FastEnsembleDeepSDFMirrored(lat_dim_glob=64, lat_dim_loc=32, n_loc=39, n_symm_pairs=16,
                                      anchors=anchors, hidden_dim=200, n_layers=4, pos_mlp_dim=pos_mlp_dim)
    return net, anchors


def build_deformation(anchors):
    torch.manual_seed(0)
    return DeformationNetwork(mode="compress", lat_dim_expr=200, lat_dim_id=32, lat_dim_glob_shape=64,
                              lat_dim_loc_shape=32, n_loc=39, anchors=anchors, hidden_dim=512, nlayers=6,
                              input_dim=3, out_dim=3)


def build_npm():
    torch.manual_seed(0)
    return DeepSDF(lat_dim=512, hidden_dim=1024, nlayers=8, geometric_init=True)


def sample_latent(name, gen):
    mean = torch.from_numpy(np.load(os.path.join(ASSETS, f"{name}_lat_mean.npy")))
    std = torch.from_numpy(np.load(os.path.join(ASSETS, f"{name}_lat_std.npy")))
    return torch.randn(mean.shape, generator=gen) * std * 0.85 + mean


def query_points(gen, n_uniform, anchors):
    lo, hi = torch.tensor(MINI), torch.tensor(MAXI)
    uni = torch.rand(n_uniform, 3, generator=gen) * (hi - lo) + lo
    a = anchors.reshape(39, 3)
    on_anchor = a[:8].clone()                                  # exactly on an anchor
    near = a[torch.randint(0, 39, (120,), generator=gen)] + 0.02 * torch.randn(120, 3, generator=gen)
    far = torch.tensor([[3.0, 3.0, 3.0], [-2.0, 0.5, 1.5], [0.55, 0.75, 0.4], [-.55, -.5, -.95]])
    return torch.cat([uni, on_anchor, near, far], dim=0)


def export_inputs():
    """Benchmark/test INPUT data of the reference's assets/ (not source): mean anchors as float32
    (the cast of fitting_pointclouds.py:83) and the latent statistics used for sampling
    (fitting_pointclouds.py:199-206).  The GPU box has no /root/reference, so they travel here."""
    np.save(os.path.join(HERE, "anchors_mean_39.npy"),
            np.load(os.path.join(ASSETS, "anchors_39.npy")).astype(np.float32))
    np.savez(os.path.join(HERE, "nphm_lat_stats.npz"), mean=np.load(os.path.join(ASSETS, "nphm_lat_mean.npy")),
             std=np.load(os.path.join(ASSETS, "nphm_lat_std.npy")))
    np.savez(os.path.join(HERE, "npm_lat_stats.npz"), mean=np.load(os.path.join(ASSETS, "npm_lat_mean.npy")),
             std=np.load(os.path.join(ASSETS, "npm_lat_std.npy")))


def main():
    export_inputs()
    gen = torch.Generator().manual_seed(1234)

    # ------------------------------------------------------------------ NPHM identity
    net, anchors = build_identity()
    out = {"state_sha256": state_hash(net)}
    lat = sample_latent("nphm", gen).float()
    xyz = query_points(gen, 2048, anchors).float()
    out["lat"] = lat.numpy()
    out["xyz"] = xyz.numpy()
    with torch.no_grad():
        net.eval()
        sdf, anc = net(xyz[None], lat[None, None].repeat(1, xyz.shape[0], 1), None)
        out["sdf_eval"] = sdf.numpy()
        out["anchors"] = anc.numpy()
        net.train()
        sdf, _ = net(xyz[None], lat[None, None], None)          # lat_rep.shape[1] == 1 broadcast
        out["sdf_train"] = sdf.numpy()

        # batch of two different latents, N not a multiple of 32, 2-D xyz handled separately
        lat2 = torch.stack([sample_latent("nphm", gen), sample_latent("nphm", gen)]).float()
        xyz2 = torch.stack([query_points(gen, 300 - 132, anchors), query_points(gen, 300 - 132, anchors)]).float()
        net.eval()
        sdf2, anc2 = net(xyz2, lat2[:, None], None)
        out["b2_lat"] = lat2.numpy(); out["b2_xyz"] = xyz2.numpy()
        out["b2_sdf_eval"] = sdf2.numpy(); out["b2_anchors"] = anc2.numpy()

        # per-point latents (general path): every point its own latent
        latp = torch.stack([sample_latent("nphm", gen) for _ in range(48)]).float()[None]
        xyzp = query_points(gen, 48 - 132 + 132, anchors)[:48].float()[None]
        net.train()
        sdfp, _ = net(xyzp, latp, None)
        out["pp_lat"] = latp.numpy(); out["pp_xyz"] = xyzp.numpy(); out["pp_sdf_train"] = sdfp.numpy()

        # get_logits on a small lattice, chunked -> per-chunk overwrite voxels
        net.eval()
        res = 14
        grid = torch.from_numpy(create_grid_points_from_bounds(MINI, MAXI, res)).float()[None]
        out["grid_res"] = np.int64(res)
        out["grid_chunk"] = np.int64(500)
        out["grid_logits_eval"] = get_logits(net, lat, grid, nbatch_points=500)
        net.train()
        out["grid_logits_train"] = get_logits(net, lat, grid, nbatch_points=500)

        # stress: weights x2.5 (sharper, trained-like activations incl. the softplus linear branch)
        net.eval()
        for i in range(5):
            getattr(net.ensembled_deep_sdf, f"lin{i}").weight.mul_(2.5)
        sdf_s, _ = net(xyz[None, :1024], lat[None, None], None)
        out["stress_scale"] = np.float32(2.5)
        out["stress_sdf_eval"] = sdf_s.numpy()
    np.savez_compressed(os.path.join(HERE, "nphm_identity.npz"), **out)
    print("nphm_identity.npz", {k: getattr(v, "shape", v) for k, v in out.items()})

    # pos_mlp_dim = 128 variant (nphm_def.yaml:14)
    net128, _ = build_identity(pos_mlp_dim=128)
    with torch.no_grad():
        net128.eval()
        s128, a128 = net128(xyz[None, :256], lat[None, None], None)
    np.savez_compressed(os.path.join(HERE, "nphm_identity_pos128.npz"), state_sha256=state_hash(net128),
                        lat=lat.numpy(), xyz=xyz[:256].numpy(), sdf_eval=s128.numpy(), anchors=a128.numpy())

    # ------------------------------------------------------------------ deformation (compress)
    net, anchors = build_identity()
    net.eval()
    dnet = build_deformation(anchors)
    dnet.eval()
    out = {"state_sha256": state_hash(dnet)}
    z_ex = 0.01 * torch.randn(200, generator=gen)
    lat_all = torch.cat([lat, z_ex])[None, None]                                   # [1,1,1544]
    xyz_d = query_points(gen, 1024 - 132, anchors).float()[None]
    with torch.no_grad():
        _, anc = net(torch.zeros(1, 1, 3), lat[None, None], None)
        off, rest = dnet(xyz_d, lat_all.repeat(1, xyz_d.shape[1], 1), anc)
        out.update(lat=lat_all.numpy(), xyz=xyz_d.numpy(), anchors=anc.numpy(), offsets=off.numpy(), rest=rest.numpy())
        # two-stage on a small lattice (get_logits_backward needs anchors=None for decoder_expr in
        # the reference, which only an NPM-style DeepSDF accepts; emulate with a closure that
        # supplies the anchors, the arithmetic is the reference's)
        class _Expr(torch.nn.Module):
            def forward(self, p, l, a):
                return dnet(p, l, anc)
        res = 10
        grid = torch.from_numpy(create_grid_points_from_bounds(MINI, MAXI, res)).float()[None]
        out["grid_res"] = np.int64(res)
        out["grid_chunk"] = np.int64(300)
        out["two_stage_logits"] = get_logits_backward(net, _Expr(), lat, lat_all.reshape(-1), grid, nbatch_points=300)
    np.savez_compressed(os.path.join(HERE, "deformation.npz"), **out)
    print("deformation.npz", {k: getattr(v, "shape", v) for k, v in out.items()})

    # ------------------------------------------------------------------ NPM global DeepSDF
    npm = build_npm()
    out = {"state_sha256": state_hash(npm)}
    lat_n = sample_latent("npm", gen).float()
    xyz_n = query_points(gen, 512 - 132, anchors).float()[None]
    with torch.no_grad():
        sdf_n, _ = npm(xyz_n, lat_n[None, None].repeat(1, xyz_n.shape[1], 1))
        res = 8
        grid = torch.from_numpy(create_grid_points_from_bounds(MINI, MAXI, res)).float()[None]
        out.update(lat=lat_n.numpy(), xyz=xyz_n.numpy(), sdf=sdf_n.numpy(), grid_res=np.int64(res),
                   grid_logits=get_logits(npm, lat_n, grid, nbatch_points=200))
    np.savez_compressed(os.path.join(HERE, "npm.npz"), **out)
    print("npm.npz", {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
