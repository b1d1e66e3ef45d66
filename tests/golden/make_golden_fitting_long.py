"""Long-horizon golden fixture of the latent-fitting loop, produced by RUNNING THE REFERENCE'S OWN LOOP
(src/NPHM/models/fitting.py:14-177 and :180-288; PyTorch-CPU, fp32) in the build container:

    python tests/golden/make_golden_fitting_long.py        ->  tests/golden/fitting_long.npz

Configuration = scripts/fitting/fitting_pointclouds.py:253-276 (lambdas, schedule, n_steps = 1000) with the
reference's own ``step_scale`` knob at 1/4: 250 Adam steps that cross EVERY transition of the schedule
(keys 200 / 400 / 500 / 600 / 800 at steps 50 / 100 / 125 / 150 / 200) and both shrinkages of the
surface-loss clamp (steps 62 and 125).  Observations: three synthetic scans of 400 points on the zero
level set of a seeded ground-truth identity (random points projected onto the level set of the
reference network by Newton steps along its gradient; no dataset here).  Stored: observations,
per-step loss terms as the reference prints them (8 decimals), the fitted codes and anchors; for the
identity-only loop (which prints nothing) the total loss of every step and the fitted code.

Takes ~40 minutes on 8 cores (the reference's dense 40-member double-graph step on the CPU).  Regenerating it in
the build container (8 torch threads, recorded in the fixture) reproduced every array of the committed file bit
for bit."""
import io
import os
import sys
import time
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
from make_golden_fitting import parse_history            # noqa: E402

LAMBDAS = {"surface": 2.0, "reg_expr": 0.01, "reg_global": 0.25, "reg_unobserved": 10, "reg_loc": 0.05,
           "symm_dist": 5.0}                                            # fitting_pointclouds.py:253-259
SCHEDULE = {"lr": {200: 2, 400: 2, 600: 2, 800: 2}, "symm_dist": {200: 10, 500: 9999},
            "reg_glob": {200: 3, 600: 10}, "reg_loc": {500: 3, 600: 10}, "reg_expr": {600: 10}}   # :261-266
N_STEPS = 1000
STEP_SCALE = 0.25


def level_set_points(net, lat, n, gen):
    """n points with |sdf| < 1e-4 for the identity code ``lat``: uniform samples in the fitting box,
    three Newton steps along the SDF gradient (train mode: no eval-mode overwrite)."""
    lo, hi = torch.tensor([-0.30, -0.35, -0.30]), torch.tensor([0.30, 0.35, 0.30])
    x = torch.rand(6 * n, 3, generator=gen) * (hi - lo) + lo
    cond = lat.reshape(1, 1, -1)
    for _ in range(4):
        x = x.detach().requires_grad_(True)
        sdf, _ = net(x[None], cond.repeat(1, x.shape[0], 1), None)
        (g,) = torch.autograd.grad(sdf.sum(), x)
        x = x - sdf[0] * g / (g.square().sum(-1, keepdim=True) + 1e-12)
    x = x.detach()
    with torch.no_grad():
        sdf, _ = net(x[None], cond.repeat(1, x.shape[0], 1), None)
    ok = (sdf[0, :, 0].abs() < 1e-4) & ((x > lo) & (x < hi)).all(-1)
    assert int(ok.sum()) >= n, int(ok.sum())
    return x[ok][:n].contiguous()


def main():
    shape_net, anchors = G.build_identity()
    expr_net = G.build_deformation(anchors).eval()
    shape_net.train()                                    # fitting_pointclouds.py:268
    gen = torch.Generator().manual_seed(1234)
    lat_gt = G.sample_latent("nphm", gen).float()
    pts = level_set_points(shape_net, lat_gt, 1200, gen)
    obs = [pts[i * 400:(i + 1) * 400].clone() for i in range(3)]
    keys = list(LAMBDAS.keys())

    t0 = time.time()
    torch.manual_seed(0)
    buf = io.StringIO()
    with redirect_stdout(buf):
        lat_e, lat_s, anc = inference_iterative_root_finding_joint(
            shape_net, expr_net, [o.clone() for o in obs], dict(LAMBDAS), N_STEPS,
            {k: dict(v) for k, v in SCHEDULE.items()}, step_scale=STEP_SCALE)
    hist = parse_history(buf.getvalue(), keys)
    n_iter = int(N_STEPS * STEP_SCALE)
    assert hist.shape == (n_iter, len(keys) + 1), hist.shape
    print("joint loop: %.0f s" % (time.time() - t0), flush=True)

    lam_id = {k: v for k, v in LAMBDAS.items() if k != "reg_expr"}
    keys_id = list(lam_id.keys())
    # the identity-only loop builds its report line but never prints it (fitting.py:280-283): the total loss of
    # every step is recorded at its loss.backward() instead
    totals = []
    backward = torch.Tensor.backward

    def recording_backward(self, *a, **k):
        totals.append(float(self.detach()))
        return backward(self, *a, **k)

    torch.manual_seed(1)
    torch.Tensor.backward = recording_backward
    try:
        lat_s2, anc2 = inference_identity_space(shape_net, [o.clone() for o in obs], dict(lam_id), N_STEPS,
                                                {k: dict(v) for k, v in SCHEDULE.items()}, step_scale=STEP_SCALE)
    finally:
        torch.Tensor.backward = backward
    hist_id = np.asarray(totals, np.float64)
    assert hist_id.shape[0] == n_iter, hist_id.shape
    print("identity loop: %.0f s" % (time.time() - t0), flush=True)

    out = dict(obs0=obs[0].numpy(), obs1=obs[1].numpy(), obs2=obs[2].numpy(), lat_gt=lat_gt.numpy(),
               n_steps=np.int64(N_STEPS), step_scale=np.float64(STEP_SCALE),
               keys=np.array(keys), history=hist, lat_expr=lat_e.detach().numpy(), lat_shape=lat_s.detach().numpy(),
               anchors=anc.detach().numpy(), id_keys=np.array(keys_id), id_total_loss=hist_id,
               id_lat_shape=lat_s2.detach().numpy(), id_anchors=anc2.detach().numpy(),
               shape_sha256=G.state_hash(shape_net), expr_sha256=G.state_hash(expr_net))
    out["torch_threads"] = np.int64(torch.get_num_threads())
    np.savez_compressed(os.environ.get("NPHM_GOLDEN_OUT", os.path.join(HERE, "fitting_long.npz")), **out)
    print("fitting_long.npz", {k: getattr(v, "shape", v) for k, v in out.items()})
    print(hist[::25])


if __name__ == "__main__":
    main()
