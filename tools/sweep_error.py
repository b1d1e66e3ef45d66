#!/usr/bin/env python3
"""Error sweep of the fast modes of the identity kernel (pruned, adaptive precision) against the dense exact-fp32
kernel over many latents, latent scales and weight scales on a 64^3 lattice, plus the trained-like checkpoint
(tests/golden/trained_state.npz, 8 of its codes).  Development tool.
    python tools/sweep_error.py [mode[:light_tol:mid_tol[:prune_tol]] ...]      (default: bf16x3a2 f16x3a2)
"auto" as a mode = the per-checkpoint calibration (numerics = "auto"); its pick is printed per setting."""
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
RES = int(os.environ.get("NPHM_SWEEP_RES", "64"))          # 256: the full extraction lattice (dense fp32 reference 0.3 s per latent)
N_LAT = int(os.environ.get("NPHM_SWEEP_LATENTS", "12"))
axes = R.grid_axes(U.MINI, U.MAXI, RES)
modes = sys.argv[1:] or ["bf16x3a2", "f16x3a2"]
prune = float(os.environ.get("NPHM_SWEEP_PRUNE", "1e-7"))


def apply(n, spec):
    if spec == "auto":
        n.numerics = "auto"
        return
    f = spec.split(":")
    n.precision = f[0]
    n.light_tol = float(f[1]) if len(f) > 1 and f[1] else None
    n.mid_tol = float(f[2]) if len(f) > 2 and f[2] else None
    n.prune_tol = float(f[3]) if len(f) > 3 and f[3] else prune


def sweep(n, lats, label):
    errs, mags, members = {m: [] for m in modes}, [], {m: [] for m in modes}
    for lat in lats:
        n.precision, n.prune_tol = "f32", -1.0
        ref = R.evaluate_grid(n, lat, axes, hack_chunk=0)
        mags.append(float(ref.abs().max()))
        for m in modes:
            apply(n, m)
            st = torch.zeros(16, dtype=torch.int64, device=dev)
            got = R.evaluate_grid(n, lat, axes, hack_chunk=0, stats=st)
            errs[m].append(float((got - ref).abs().max()))
            s = st.cpu().numpy().astype(float)
            members[m].append((s[0] - s[15] - s[14], s[14], s[15]))
    line = f"{label}: max |sdf| {max(mags):.3f};"
    for m in modes:
        h, t, l = np.mean(members[m], axis=0) / RES ** 3
        line += f"  {m} {max(errs[m]):.2e} ({h:.2f}/{t:.2f}/{l:.2f})"
        if m == "auto":
            c = n.calibration
            line += f" [picked {c['precision']} light {c['light_tol']} mid {c['mid_tol']} prune {c['prune_tol']:g}, sample err {c['error']:.2e}]"
    print(line, flush=True)


for wscale in (1.0, 1.5, 2.5):
    n = U.build_identity(device=dev).eval()
    with torch.no_grad():
        for i in range(5):
            getattr(n.ensembled_deep_sdf, f"lin{i}").weight.mul_(wscale)
    for lscale in (0.85, 1.5, 3.0):
        sweep(n, [U.sample_latent(seed, scale=lscale).to(dev) for seed in range(N_LAT)], f"weights x{wscale} latent-sigma x{lscale}")
n, codes = U.build_trained_identity(device=dev)
sweep(n.eval(), [codes[c] for c in range(0, 64, 8)], "trained-like checkpoint, 8 codes")
