#!/usr/bin/env python3
"""Error sweep of the default (pruned, adaptive-precision) identity kernel against the dense exact-fp32
kernel over many latents, latent scales and weight scales on a 64^3 lattice.  Development tool."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util as U  # noqa: E402
from nphm_amd import reconstruction as R  # noqa: E402

dev = torch.device("cuda:0")
axes = R.grid_axes(U.MINI, U.MAXI, 64)
worst = {}
for wscale in (1.0, 1.5, 2.5):
    n = U.build_identity(device=dev).eval()
    with torch.no_grad():
        for i in range(5):
            getattr(n.ensembled_deep_sdf, f"lin{i}").weight.mul_(wscale)
    for lscale in (0.85, 1.5, 3.0):
        errs, mags = [], []
        for seed in range(12):
            lat = U.sample_latent(seed, scale=lscale).to(dev)
            n.precision, n.prune_tol = "f32", -1.0
            ref = R.evaluate_grid(n, lat, axes, hack_chunk=0)
            n.precision, n.prune_tol = os.environ.get("NPHM_SWEEP_PRECISION", "bf16x3a2"), 1e-7
            got = R.evaluate_grid(n, lat, axes, hack_chunk=0)
            errs.append(float((got - ref).abs().max()))
            mags.append(float(ref.abs().max()))
        worst[(wscale, lscale)] = (max(errs), max(mags))
        print(f"weights x{wscale} latent-sigma x{lscale}: max |default - dense f32| over 12 latents = {max(errs):.3e} "
              f"(max |sdf| {max(mags):.3f})")
