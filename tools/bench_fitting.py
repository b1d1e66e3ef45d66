#!/usr/bin/env python3
"""Latent-code fitting (BASELINE.json configs[4]): steps/s of inference_iterative_root_finding_joint on
one GPU with synthetic single-view observations (points near the zero level set of a seeded
ground-truth latent; the NPHM dataset is not available).  Development tool."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util as U  # noqa: E402
from nphm_amd import fitting as F  # noqa: E402
from nphm_amd import reconstruction as R  # noqa: E402

LAMBDAS = {"surface": 2.0, "reg_expr": 0.01, "reg_global": 0.25, "reg_unobserved": 10, "reg_loc": 0.05,
           "symm_dist": 5.0}
SCHEDULE = {"lr": {200: 2, 400: 2, 600: 2, 800: 2}, "symm_dist": {200: 10, 500: 9999}, "reg_glob": {200: 3, 600: 10},
            "reg_loc": {500: 3, 600: 10}, "reg_expr": {600: 10}}


def synthetic_observations(shape_net, dev, n_obs=3, n_pts=2500, res=96):
    """Marching-cubes vertices of the ground-truth identity's zero level set, a random subset per
    observation (the deformation is what the fit has to explain away: here the neutral pose)."""
    lat = U.sample_latent(5).to(dev)
    shape_net.eval()
    axes = R.grid_axes(U.MINI, U.MAXI, res)
    vol = R.evaluate_grid(shape_net, lat, axes, hack_chunk=0).cpu().numpy()
    mesh = R.mesh_from_logits(vol, U.MINI, U.MAXI, res)
    v = torch.from_numpy(np.asarray(mesh.vertices)).float()
    g = torch.Generator().manual_seed(0)
    return [v[torch.randperm(v.shape[0], generator=g)[:n_pts]].to(dev) for _ in range(n_obs)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--backend", default=None, choices=[None, "composite"])
    ap.add_argument("--profile", action="store_true", help="synchronising per-phase timers (slower)")
    ap.add_argument("--torch-profile", action="store_true", help="torch.profiler table of 5 steps")
    ap.add_argument("--with-unused-nabla", action="store_true", help="also evaluate the reference's dead nabla() call")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    shape_net = U.build_identity(device=dev)
    expr_net = U.build_deformation(device=dev).eval()
    obs = synthetic_observations(shape_net, dev)
    shape_net.train()
    if args.backend:
        shape_net.backend = args.backend
        expr_net.backend = args.backend
    phases = {}
    if args.profile:
        import nphm_amd.fitting as FM
        import nphm_amd.iterative_root_finding as IRF

        def timed(name, fn):
            def wrapper(*a, **k):
                torch.cuda.synchronize()
                t = time.perf_counter()
                out = fn(*a, **k)
                torch.cuda.synchronize()
                phases[name] = phases.get(name, 0.0) + time.perf_counter() - t
                return out
            return wrapper
        FM.search = timed("search(total)", FM.search)
        IRF.jac = timed("search.jac+inverse-input", IRF.jac)
        IRF.broyden = timed("search.broyden(python)", IRF.broyden)
        if hasattr(expr_net, "broyden"):
            expr_net.broyden = timed("search.broyden(fused)", expr_net.broyden)
        FM.jac = timed("jac(implicit diff)", FM.jac)
        FM.nabla = timed("nabla(unused result)", FM.nabla)
        shape_net.forward = timed("identity forward (composite, all calls)", shape_net.forward)
        expr_net.forward = timed("deformation forward (all calls)", expr_net.forward)
        orig_backward = torch.Tensor.backward
        torch.Tensor.backward = timed("loss.backward", orig_backward)
    torch.manual_seed(0)
    cfg = {k: dict(v) for k, v in SCHEDULE.items()}
    kw = {"compute_unused_sdf_grad": args.with_unused_nabla}
    F.inference_iterative_root_finding_joint(shape_net, expr_net, obs, dict(LAMBDAS), args.warmup, cfg, verbose=False, **kw)
    torch.cuda.synchronize()
    if args.torch_profile:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            F.inference_iterative_root_finding_joint(shape_net, expr_net, obs, dict(LAMBDAS), 5, cfg, verbose=False, **kw)
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=28, max_name_column_width=48))
        print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=14, max_name_column_width=60))
    hist = []
    t0 = time.perf_counter()
    F.inference_iterative_root_finding_joint(shape_net, expr_net, obs, dict(LAMBDAS), args.steps, cfg, verbose=False,
                                             history=hist, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if args.profile:
        for k, v in sorted(phases.items(), key=lambda kv: -kv[1]):
            print(f"  {k:45s} {v / (args.steps + args.warmup) * 1e3:8.2f} ms/step")
    print(json.dumps({"workload": "latent fitting, 3 observations x 2500 pts, 5x1000 pts/step", "steps": args.steps,
                      "steps_per_s": args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                      "backend": args.backend or "hip+composite", "unused_nabla": args.with_unused_nabla, "first_loss": hist[0]["loss"],
                      "last_loss": hist[-1]["loss"], "n_valid_last": hist[-1]["n_valid"]}))


if __name__ == "__main__":
    main()
