"""Golden fixture on a TRAINED-LIKE checkpoint of the forward-deformation network: the REFERENCE's DeformationNetwork
(src/NPHM/models/deepSDF.py:118-239, PyTorch-CPU, fp32) on a state_dict that tools/train_synthetic_expressions.py trained
(the reference's own module, on the CPU of the build container) against analytic expression warps of the synthetic heads,
with the trained-like identity decoder of tests/golden/trained_state.npz frozen - the pair of checkpoints the two-stage
evaluation and the fitting loop meet in practice.  Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_trained_def.py [gpurun_out/r4/trained_expr.npz]

Writes
  trained_def_state.npz  the deformation network's state_dict (strict-loadable into the reference's and this repo's module),
                         the expression codes of PAIRS and the subjects they belong to
  trained_def.npz        reference outputs per (subject, expression) pair: offsets at 2 048 points (half near the subject's
                         anchors, half uniform in the head box), and for pair 0 the reference's get_logits_backward
                         (deformation -> trained identity decoder, eval mode, chunk 1 500) on a 20^3 lattice."""
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
from NPHM.models.deepSDF import DeformationNetwork                        # noqa: E402
from NPHM.models.EnsembledDeepSDF import FastEnsembleDeepSDFMirrored     # noqa: E402
from NPHM.models.reconstruction import get_logits_backward                # noqa: E402
from NPHM.utils.reconstruction import create_grid_points_from_bounds      # noqa: E402

PAIRS = ((0, 3), (1, 7), (2, 11), (5, 0))          # (subject, expression); expression 0 is the neutral one
N_EXPR = 12
LATTICE_RES, LATTICE_CHUNK = 20, 1500


def main():
    import _sources
    ck, from_scratch = _sources.load("expr")        # the training run's output, or trained_def_state.npz + trained_expr_codes.npz
    sd = {k[3:]: torch.from_numpy(ck[k]) for k in ck if k.startswith("sd.")}
    mean_anchors = torch.from_numpy(np.load(os.path.join(G.ASSETS, "anchors_39.npy"))).float()[None, None]
    dnet = DeformationNetwork(mode="compress", lat_dim_expr=200, lat_dim_id=32, lat_dim_glob_shape=64, lat_dim_loc_shape=32,
                              n_loc=39, anchors=mean_anchors, hidden_dim=512, nlayers=6, input_dim=3, out_dim=3)
    dnet.load_state_dict(sd, strict=True)
    dnet.eval()
    ick = np.load(os.path.join(HERE, "trained_state.npz"))
    inet = FastEnsembleDeepSDFMirrored(lat_dim_glob=64, lat_dim_loc=32, n_loc=39, n_symm_pairs=16, anchors=mean_anchors,
                                       hidden_dim=200, n_layers=4, pos_mlp_dim=256)
    inet.load_state_dict({k[3:]: torch.from_numpy(ick[k]) for k in ick.files if k.startswith("sd.")}, strict=True)
    inet.eval()
    codes = torch.from_numpy(ick["codes"]).float()
    z_all = torch.from_numpy(ck["z_ex"]).float()
    z_pairs = torch.stack([z_all[s * N_EXPR + e] for s, e in PAIRS])
    if from_scratch:
        np.savez(os.path.join(HERE, "trained_def_state.npz"), **{"sd." + k: v.numpy() for k, v in dnet.state_dict().items()},
                 z_ex=z_pairs.numpy(), pairs=np.asarray(PAIRS), trace=ck["trace"], meta=ck["meta"])
        np.savez_compressed(os.path.join(HERE, "trained_expr_codes.npz"), z_ex=ck["z_ex"])     # every trained expression code
    out = {"state_sha256": np.array(G.state_hash(dnet)), "pairs": np.asarray(PAIRS)}
    gen = torch.Generator().manual_seed(11)
    lo, hi = torch.tensor([-0.45, -0.50, -0.45]), torch.tensor([0.45, 0.55, 0.45])
    with torch.no_grad():
        for i, (s, e) in enumerate(PAIRS):
            lat_id = codes[s]
            _, anc = inet(torch.zeros(1, 1, 3), lat_id[None, None], None)                    # predicted anchors [1,39,3]
            pick = torch.randint(0, 39, (1024,), generator=gen)
            x = torch.cat([anc[0][pick] + 0.08 * torch.randn(1024, 3, generator=gen),
                           torch.rand(1024, 3, generator=gen) * (hi - lo) + lo])[None]
            lat_all = torch.cat([lat_id, z_pairs[i]])[None, None]
            off, rest = dnet(x, lat_all.repeat(1, x.shape[1], 1), anc)
            out[f"p{i}_xyz"], out[f"p{i}_offsets"], out[f"p{i}_anchors"] = x.numpy(), off.numpy(), anc.numpy()
            print(f"pair {i} (subject {s}, expression {e}): |offset| mean {float(off.norm(dim=-1).mean()):.3e} max {float(off.norm(dim=-1).max()):.3e}")
            if i == 0:
                class _Expr(torch.nn.Module):                                                   # see make_golden.py: anchors supplied
                    def forward(self, p, l, a):
                        return dnet(p, l, anc)
                grid = torch.from_numpy(create_grid_points_from_bounds(G.MINI, G.MAXI, LATTICE_RES)).float()[None]
                out["lattice_res"], out["lattice_chunk"] = np.int64(LATTICE_RES), np.int64(LATTICE_CHUNK)
                out["two_stage_logits"] = get_logits_backward(inet, _Expr(), lat_id, lat_all.reshape(-1), grid,
                                                              nbatch_points=LATTICE_CHUNK)
    wts = {k: float(v.abs().max()) for k, v in dnet.state_dict().items() if k.endswith("weight")}
    out["max_weight"] = np.float64(max(wts.values()))
    np.savez_compressed(os.path.join(HERE, "trained_def.npz"), **out)
    print("trained_def.npz", {k: getattr(v, "shape", v) for k, v in out.items()}, "largest weights", wts)


if __name__ == "__main__":
    main()
