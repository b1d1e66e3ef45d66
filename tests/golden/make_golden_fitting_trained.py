"""The reference's joint fitting loop (src/NPHM/models/fitting.py:14-177; PyTorch-CPU, fp32) on the TRAINED-LIKE pair of
checkpoints - identity decoder tests/golden/trained_state.npz, deformation network tests/golden/trained_def_state.npz
(tools/train_synthetic_heads.py / tools/train_synthetic_expressions.py) - in the build container:

    python tests/golden/make_golden_fitting_trained.py [gpurun_out/r4/trained_expr.npz]   ->  tests/golden/fitting_trained.npz

Observations: three "scans" of ONE trained subject under three trained expressions - 400 points each on the zero level set
of the subject's identity code (Newton projection through the reference network), posed by the trained deformation network
with that expression's code (x_posed = x + F_ex(x)).  Configuration of fitting_pointclouds.py:253-276 with the reference's
step_scale at 0.06: 60 Adam steps that still cross every transition of the schedule (steps 12 / 24 / 30 / 36 / 48).
Stored: observations, per-step loss terms as the reference prints them, fitted codes and anchors, and the code gradients of
the first three steps as the reference's autograd computed them (grad_shape [3,1,1,1344], grad_expr [3,3,1,200]) plus those
of ONE step started at the fitted codes (grad_fit_shape, grad_fit_expr; `start_codes`).  ~10 minutes on 8 cores;
`--grads-only` recomputes the gradient arrays from the committed trace (1 minute)."""
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
torch.Tensor.cuda = lambda self, *a, **k: self

import make_golden as G                                                   # noqa: E402
from NPHM.models.deepSDF import DeformationNetwork                        # noqa: E402
from NPHM.models.EnsembledDeepSDF import FastEnsembleDeepSDFMirrored     # noqa: E402
from NPHM.models.fitting import inference_iterative_root_finding_joint    # noqa: E402
from make_golden_fitting import parse_history                             # noqa: E402
from make_golden_fitting_long import LAMBDAS, SCHEDULE, N_STEPS, level_set_points   # noqa: E402

STEP_SCALE = 0.06
SUBJECT, EXPRESSIONS, N_EXPR = 3, (2, 5, 9), 12
GRAD_STEPS = 3          # steps whose code gradients (d loss / d z_id, d loss / d z_ex) are stored


class start_codes:
    """Both loops (the reference's and nphm_amd's mirror) create their codes with torch.zeros([n_obs, 1, 200]) and
    torch.zeros([1, 1, lat_dim]): inside this context those two calls return the given tensors instead - the loop starts
    its first step AT these codes."""

    def __init__(self, z_expr, z_shape):
        self.init = {tuple(z_expr.shape): z_expr, tuple(z_shape.shape): z_shape}

    def __enter__(self):
        self.zeros = torch.zeros
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
        torch.zeros = self.zeros


def record_adam_grads(store, n):
    """torch.optim.Adam.step that first copies the gradient of its (single) parameter into ``store`` (first n calls)"""
    adam_step = torch.optim.Adam.step

    def recording_step(self, *a, **k):
        if len(store) < n:
            store.append([p.grad.detach().clone() for grp in self.param_groups for p in grp["params"]][0])
        return adam_step(self, *a, **k)
    return adam_step, recording_step


def main():
    import _sources
    z_all = torch.from_numpy(_sources.load("expr")[0]["z_ex"]).float()     # (gpurun_out/r4/trained_expr.npz, or the committed codes)
    mean_anchors = torch.from_numpy(np.load(os.path.join(G.ASSETS, "anchors_39.npy"))).float()[None, None]
    ick = np.load(os.path.join(HERE, "trained_state.npz"))
    shape_net = FastEnsembleDeepSDFMirrored(lat_dim_glob=64, lat_dim_loc=32, n_loc=39, n_symm_pairs=16, anchors=mean_anchors,
                                            hidden_dim=200, n_layers=4, pos_mlp_dim=256)
    shape_net.load_state_dict({k[3:]: torch.from_numpy(ick[k]) for k in ick.files if k.startswith("sd.")}, strict=True)
    dck = np.load(os.path.join(HERE, "trained_def_state.npz"))
    expr_net = DeformationNetwork(mode="compress", lat_dim_expr=200, lat_dim_id=32, lat_dim_glob_shape=64, lat_dim_loc_shape=32,
                                  n_loc=39, anchors=mean_anchors, hidden_dim=512, nlayers=6, input_dim=3, out_dim=3)
    expr_net.load_state_dict({k[3:]: torch.from_numpy(dck[k]) for k in dck.files if k.startswith("sd.")}, strict=True)
    expr_net.eval()
    shape_net.train()                                    # fitting_pointclouds.py:268
    codes = torch.from_numpy(ick["codes"]).float()
    lat_gt = codes[SUBJECT]
    gen = torch.Generator().manual_seed(4321)
    obs = []
    with torch.no_grad():
        _, anc = shape_net(torch.zeros(1, 1, 3), lat_gt[None, None], None)
    for e in EXPRESSIONS:
        x = level_set_points(shape_net, lat_gt, 400, gen)
        lat_all = torch.cat([lat_gt, z_all[SUBJECT * N_EXPR + e]])[None, None]
        with torch.no_grad():
            off, _ = expr_net(x[None], lat_all.repeat(1, x.shape[0], 1), anc)
        obs.append((x + off[0]).contiguous())
        print(f"expression {e}: |offset| mean {float(off.norm(dim=-1).mean()):.3e} max {float(off.norm(dim=-1).max()):.3e}", flush=True)
    keys = list(LAMBDAS.keys())
    t0 = time.time()
    torch.manual_seed(0)
    buf = io.StringIO()
    # the gradients the reference's autograd hands its two optimizers in the first GRAD_STEPS steps (opt.step(), then
    # opt_expr.step(), fitting.py:166-167): recorded at the optimizers' door, the loop itself is untouched
    grads = []
    grads_only = "--grads-only" in sys.argv       # reuse the committed trace / codes, (re)compute the gradient arrays only
    if grads_only:
        old = np.load(os.path.join(HERE, "fitting_trained.npz"))
        lat_e, lat_s, anc_f = (torch.from_numpy(old[k]) for k in ("lat_expr", "lat_shape", "anchors"))
    adam_step, recording_step = record_adam_grads(grads, 2 * GRAD_STEPS)
    torch.optim.Adam.step = recording_step
    try:
        with redirect_stdout(buf):
            if grads_only:
                inference_iterative_root_finding_joint(shape_net, expr_net, [o.clone() for o in obs], dict(LAMBDAS), int(np.ceil(GRAD_STEPS / STEP_SCALE)),
                                                       {k: dict(v) for k, v in SCHEDULE.items()}, step_scale=STEP_SCALE)
            else:
                lat_e, lat_s, anc_f = inference_iterative_root_finding_joint(
                    shape_net, expr_net, [o.clone() for o in obs], dict(LAMBDAS), N_STEPS,
                    {k: dict(v) for k, v in SCHEDULE.items()}, step_scale=STEP_SCALE)
    finally:
        torch.optim.Adam.step = adam_step
    # ... and ONE step started AT the fitted codes (same seed, same draw as step 0): the gradients of a backward pass at
    # non-trivial codes, free of the loop's own sensitivity (after Adam's first update the two implementations' codes differ by
    # up to 2 lr in components whose gradient is round-off, so steps 1, 2 of the trace compare trajectories, not kernels)
    grads_fit = []
    adam_step, recording_step = record_adam_grads(grads_fit, 2)
    torch.optim.Adam.step = recording_step
    try:
        torch.manual_seed(0)
        with redirect_stdout(io.StringIO()), start_codes(lat_e.detach(), lat_s.detach()):
            inference_iterative_root_finding_joint(shape_net, expr_net, [o.clone() for o in obs], dict(LAMBDAS), int(np.ceil(1 / STEP_SCALE)),
                                                   {k: dict(v) for k, v in SCHEDULE.items()}, step_scale=STEP_SCALE)
    finally:
        torch.optim.Adam.step = adam_step
    assert len(grads_fit) == 2
    if grads_only:
        out = {k: old[k] for k in old.files}
        assert np.array_equal(out["grad_shape"], torch.stack(grads[0::2]).numpy()) and np.array_equal(out["grad_expr"], torch.stack(grads[1::2]).numpy())
        out.update(grad_fit_shape=grads_fit[0].numpy(), grad_fit_expr=grads_fit[1].numpy())
        np.savez_compressed(os.path.join(HERE, "fitting_trained.npz"), **out)
        print("fitting_trained.npz: gradient arrays refreshed", {k: getattr(v, "shape", v) for k, v in out.items()})
        return
    hist = parse_history(buf.getvalue(), keys)
    n_iter = int(N_STEPS * STEP_SCALE)
    assert hist.shape == (n_iter, len(keys) + 1), hist.shape
    print("joint loop: %.0f s" % (time.time() - t0), flush=True)
    out = dict(obs0=obs[0].numpy(), obs1=obs[1].numpy(), obs2=obs[2].numpy(), lat_gt=lat_gt.numpy(),
               n_steps=np.int64(N_STEPS), step_scale=np.float64(STEP_SCALE), keys=np.array(keys), history=hist,
               lat_expr=lat_e.detach().numpy(), lat_shape=lat_s.detach().numpy(), anchors=anc_f.detach().numpy(),
               grad_shape=torch.stack(grads[0::2]).numpy(), grad_expr=torch.stack(grads[1::2]).numpy(),
               grad_fit_shape=grads_fit[0].numpy(), grad_fit_expr=grads_fit[1].numpy(),
               shape_sha256=G.state_hash(shape_net), expr_sha256=G.state_hash(expr_net),
               torch_threads=np.int64(torch.get_num_threads()))
    np.savez_compressed(os.path.join(HERE, "fitting_trained.npz"), **out)
    print("fitting_trained.npz", {k: getattr(v, "shape", v) for k, v in out.items()})
    print(hist[::6])


if __name__ == "__main__":
    main()
