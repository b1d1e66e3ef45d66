"""The reference's DeformationNetwork in its OTHER conditioning modes ('glob_only', 'expr_only', 'interpolate', 'GNN';
deepSDF.py:118-239 - 'compress' is in make_golden.py) and its DeepSDF with positional encoding (num_freq_bands,
deepSDF.py:14-37, 64-73), run on PyTorch-CPU fp32 in the build container (needs /root/reference):

    python tests/golden/make_golden_def_modes.py      ->  tests/golden/def_modes.npz

Weights are not stored: seeded default init in the reference's construction order (reproduced bit for bit by nphm_amd's
modules; the SHA-256 of every reference state_dict is inside).  Per mode: a row-constant latent (one code for all points,
the lattice / fitting use) and, for 'interpolate', whose conditioning depends on the point, the per-point form."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G                                            # noqa: E402  (reference imports, builders, samplers)
from NPHM.models.deepSDF import DeepSDF, DeformationNetwork        # noqa: E402

MODES = ("glob_only", "expr_only", "interpolate", "GNN")


def build(mode, anchors, seed):
    torch.manual_seed(seed)
    # (GNN reads lat_rep[..., 64 : 64 + 39 * 32] as 32-wide local codes: lat_dim_loc_shape must be 32 there)
    return DeformationNetwork(mode=mode, lat_dim_expr=200, lat_dim_id=32, lat_dim_glob_shape=64, lat_dim_loc_shape=32,
                              n_loc=39, anchors=anchors, hidden_dim=512, nlayers=6, input_dim=3, out_dim=3)


def main():
    gen = torch.Generator().manual_seed(1234)
    net, anchors = G.build_identity()
    net.eval()
    out = {}
    lat = G.sample_latent("nphm", gen).float()
    with torch.no_grad():
        _, anc = net(torch.zeros(1, 1, 3), lat[None, None], None)            # [1,39,3]
    anc = anc.reshape(1, 1, 39, 3)
    xyz = G.query_points(gen, 640 - 132, anchors).float()[None]              # [1,N,3]
    n = xyz.shape[1]
    out["xyz"] = xyz.numpy()
    out["anchors"] = anc.numpy()
    for i, mode in enumerate(MODES):
        dnet = build(mode, anchors, seed=10 + i).eval()
        out[f"{mode}_sha256"] = G.state_hash(dnet)
        z_ex = 0.05 * torch.randn(200, generator=gen)
        # latent layout per mode (deepSDF.py:196-232): [glob 64 | 39 local codes of 32 | (interpolate: + one spare local code) | expr 200]
        width = 64 + 39 * 32 + (32 if mode == "interpolate" else 0) + 200
        z_id = G.sample_latent("nphm", gen).float()[: 64 + 39 * 32]
        pad = torch.zeros(width - 200 - z_id.numel())
        lat_all = torch.cat([z_id, pad, z_ex])[None, None]                    # [1,1,width]
        with torch.no_grad():
            off, rest = dnet(xyz, lat_all.repeat(1, n, 1), anc.repeat(1, n, 1, 1))    # (the reference's per-point calling form)
        out[f"{mode}_lat"] = lat_all.numpy()
        out[f"{mode}_offsets"] = off.numpy()
        out[f"{mode}_rest"] = rest.numpy()
    # DeepSDF with positional encoding (no config of the reference uses it; the module supports it)
    torch.manual_seed(31)
    pe = DeepSDF(lat_dim=40, hidden_dim=256, nlayers=6, geometric_init=True, out_dim=1, input_dim=3, num_freq_bands=4).eval()
    out["posenc_sha256"] = G.state_hash(pe)
    cond = 0.3 * torch.randn(1, 1, 40, generator=gen)
    with torch.no_grad():
        y, _ = pe(xyz, cond.repeat(1, n, 1))
    out["posenc_cond"] = cond.numpy()
    out["posenc_out"] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "def_modes.npz"), **out)
    print("def_modes.npz", {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
