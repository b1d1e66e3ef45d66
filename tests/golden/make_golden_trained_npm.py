"""Golden fixture on a TRAINED-LIKE checkpoint of the NPM global DeepSDF: the REFERENCE's DeepSDF (src/NPHM/models/deepSDF.py:
6-89, PyTorch-CPU, fp32; npm.yaml sizes) on a state_dict that tools/train_synthetic_npm.py trained on analytic head-like
implicit surfaces.  Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_trained_npm.py [gpurun_out/r4/trained_npm.npz]

Writes trained_npm_state.npz (state_dict as float16-free fp32 arrays + the codes used) and trained_npm.npz (per code in
CODES the reference values at 1 024 points, half near the subject's surface; for code 0 the reference's get_logits on a
16^3 lattice, chunk 1 000)."""
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
from NPHM.models.deepSDF import DeepSDF                                   # noqa: E402
from NPHM.models.reconstruction import get_logits                         # noqa: E402
from NPHM.utils.reconstruction import create_grid_points_from_bounds      # noqa: E402

CODES = (0, 3, 7, 12)
LATTICE_RES, LATTICE_CHUNK = 16, 1000


def main():
    import _sources
    ck, from_scratch = _sources.load("npm")         # the training run's output, or the committed trained_npm_state.npz
    net = DeepSDF(lat_dim=512, hidden_dim=1024, nlayers=8, geometric_init=True)
    net.load_state_dict({k[3:]: torch.from_numpy(ck[k]) for k in ck if k.startswith("sd.")}, strict=True)
    net.eval()
    codes = torch.from_numpy(ck["codes"]).float()
    if from_scratch:
        np.savez_compressed(os.path.join(HERE, "trained_npm_state.npz"), **{"sd." + k: v.numpy() for k, v in net.state_dict().items()},
                            codes=codes[list(CODES)].numpy(), code_ids=np.asarray(CODES), meta=ck["meta"])
    out = {"state_sha256": np.array(G.state_hash(net)), "code_ids": np.asarray(CODES)}
    gen = torch.Generator().manual_seed(21)
    lo, hi = torch.tensor([-0.5, -0.6, -0.55]), torch.tensor([0.5, 0.6, 0.45])
    with torch.no_grad():
        for i, c in enumerate(CODES):
            x = torch.rand(1, 1024, 3, generator=gen) * (hi - lo) + lo
            # pull half of the points towards the zero level set of THIS network: three Newton steps
            xs = x[:, :512].clone()
            for _ in range(3):
                with torch.enable_grad():
                    xs = xs.detach().requires_grad_(True)
                    v, _ = net(xs, codes[c][None, None].repeat(1, 512, 1))
                    (g,) = torch.autograd.grad(v.sum(), xs)
                xs = (xs - v * g / (g.square().sum(-1, keepdim=True) + 1e-9)).detach()
            x = torch.cat([xs.clamp(-0.9, 0.9), x[:, 512:]], 1)
            sdf, _ = net(x, codes[c][None, None].repeat(1, 1024, 1))
            out[f"c{i}_xyz"], out[f"c{i}_sdf"] = x.numpy(), sdf.numpy()
            print(f"code {c}: |sdf| near-surface half mean {float(sdf[:, :512].abs().mean()):.2e}, all max {float(sdf.abs().max()):.3f}")
        grid = torch.from_numpy(create_grid_points_from_bounds(G.MINI, G.MAXI, LATTICE_RES)).float()[None]
        out["lattice_res"], out["lattice_chunk"] = np.int64(LATTICE_RES), np.int64(LATTICE_CHUNK)
        out["lattice_logits"] = get_logits(net, codes[CODES[0]], grid, nbatch_points=LATTICE_CHUNK)
    out["max_weight"] = np.float64(max(float(v.abs().max()) for k, v in net.state_dict().items() if k.endswith("weight")))
    np.savez_compressed(os.path.join(HERE, "trained_npm.npz"), **out)
    print("trained_npm.npz", {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
