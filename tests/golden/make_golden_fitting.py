"""Golden fixture of the latent-fitting loop, produced by RUNNING THE REFERENCE'S OWN LOOP
(src/NPHM/models/fitting.py, iterative_root_finding.py, diff_operators.py; PyTorch-CPU, fp32) in the
build container:

    python tests/golden/make_golden_fitting.py        ->  tests/golden/fitting.npz

The reference's fitting.py imports pytorch3d / trimesh / pyvista without using them and calls
``.cuda()`` on an index tensor; the former are stubbed, the latter is made a no-op (CPU run).
Observations are synthetic point clouds (no dataset here); the decoders are the seeded random-init
networks of make_golden.py.  Stored: the observations, the schedule/lambdas, the per-step loss terms
and the fitted latents after 4 steps."""
import io
import os
import sys
import types
from contextlib import redirect_stdout

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, HERE)
for missing in ("trimesh", "mcubes", "pyvista", "pytorch3d", "pytorch3d.ops"):
    sys.modules.setdefault(missing, types.ModuleType(missing))
sys.modules["pytorch3d.ops"].knn_points = None
sys.modules["pytorch3d.ops"].knn_gather = None
torch.Tensor.cuda = lambda self, *a, **k: self          # fitting.py:72 on a CPU-only box

import make_golden as G                                  # noqa: E402  (reference builders, same seeds)
from NPHM.models.fitting import inference_identity_space, inference_iterative_root_finding_joint  # noqa: E402

LAMBDAS = {"surface": 2.0, "reg_expr": 0.01, "reg_global": 0.25, "reg_unobserved": 10, "reg_loc": 0.05,
           "symm_dist": 5.0}                                            # fitting_pointclouds.py:253-259
SCHEDULE = {"lr": {2: 2}, "symm_dist": {1: 10, 3: 9999}, "reg_glob": {1: 3}, "reg_loc": {2: 3}, "reg_expr": {3: 10}}
N_STEPS = 4


def parse_history(text, keys):
    rows = []
    for line in text.splitlines():
        if line.startswith("Epoch:"):
            tok = line.split()
            vals = {k: float(tok[tok.index(k) + 1]) for k in keys}
            rows.append([vals[k] for k in keys] + [float(tok[-1]) if len(tok) > 2 + 2 * len(keys) else -1.0])
    return np.asarray(rows, np.float64)


def main():
    shape_net, anchors = G.build_identity()
    expr_net = G.build_deformation(anchors).eval()
    shape_net.train()                                    # fitting_pointclouds.py:268
    gen = torch.Generator().manual_seed(77)
    obs = [(torch.rand(n, 3, generator=gen) - 0.5) * torch.tensor([0.5, 0.6, 0.5]) + torch.tensor([0.0, 0.05, 0.05])
           for n in (300, 300, 300)]      # equal sizes: the loop stacks the samples
    torch.manual_seed(0)
    buf = io.StringIO()
    with redirect_stdout(buf):
        lat_e, lat_s, anc = inference_iterative_root_finding_joint(
            shape_net, expr_net, [o.clone() for o in obs], dict(LAMBDAS), N_STEPS,
            {k: dict(v) for k, v in SCHEDULE.items()})
    keys = list(LAMBDAS.keys())
    hist = parse_history(buf.getvalue(), keys)
    assert hist.shape == (N_STEPS, len(keys) + 1), hist.shape

    lam_id = {k: v for k, v in LAMBDAS.items() if k != "reg_expr"}
    torch.manual_seed(1)
    lat_s2, anc2 = inference_identity_space(shape_net, [o.clone() for o in obs], dict(lam_id), N_STEPS,
                                            {k: dict(v) for k, v in SCHEDULE.items()})
    out = dict(obs0=obs[0].numpy(), obs1=obs[1].numpy(), obs2=obs[2].numpy(), n_steps=np.int64(N_STEPS),
               keys=np.array(keys), history=hist, lat_expr=lat_e.detach().numpy(), lat_shape=lat_s.detach().numpy(),
               anchors=anc.detach().numpy(), id_lat_shape=lat_s2.detach().numpy(), id_anchors=anc2.detach().numpy(),
               shape_sha256=G.state_hash(shape_net), expr_sha256=G.state_hash(expr_net))
    np.savez_compressed(os.path.join(HERE, "fitting.npz"), **out)
    print("fitting.npz", {k: getattr(v, "shape", v) for k, v in out.items()})
    print(hist)


if __name__ == "__main__":
    main()
