#!/usr/bin/env python3
"""What would a relaxed far-field target buy?  Calibrates the identity kernel's knobs at several targets, times a 256^3
lattice with each, and counts the tiles that lie within the marching-cubes band of the zero set (which a far-relaxed
extraction would re-evaluate at the default target).  Development tool."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util as U  # noqa: E402
from nphm_amd import reconstruction as R  # noqa: E402
from nphm_amd.numerics import calibrate_numerics  # noqa: E402


def timeit(fn, steps=5):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def main():
    dev = torch.device("cuda:0")
    res = 256
    axes = R.grid_axes(U.MINI, U.MAXI, res)
    out = {}
    for name in ("seeded", "trained"):
        if name == "seeded":
            net = U.build_identity(device=dev).eval()
            lat = U.sample_latent(0).to(dev)
        else:
            net, codes = U.build_trained_identity(device=dev)
            net.eval()
            lat = codes[3]
        rec = {}
        with torch.no_grad():
            ref = None
            for target in (5e-6, 2e-5, 5e-5, 1e-4):
                c = calibrate_numerics(net, lat[None], device=dev, target=target, target_surface=max(target, 5e-6))
                object.__setattr__(net, "_calibration", (net._weights_key(dev), c))
                net._verified_latents.clear()
                net._verified_latents[net._latent_digest(lat[None].detach().reshape(-1, net.lat_dim)[:8])] = c["error"]
                net.numerics = "auto"
                vol = R.evaluate_grid(net, lat, axes, hack_chunk=0)
                ms = timeit(lambda: R.evaluate_grid(net, lat, axes, hack_chunk=0))
                if ref is None:
                    net.precision, net.prune_tol = "f16x3", -1.0
                    ref = R.evaluate_grid(net, lat, axes, hack_chunk=0).clone()
                    net.numerics = "auto"
                err = float((vol - ref).abs().max())
                rec[str(target)] = {"ms": round(ms, 2), "Mpts/s": round(res ** 3 / ms / 1e3, 1), "full_err": err, "sample_err": c["error"],
                                    "terms_per_point": round(c["terms_per_point"], 2),
                                    "knobs": [c["precision"], c["light_tol"], c["mid_tol"], c["prune_tol"]]}
            # tiles (4x4x2 voxels) with any |value| below L x voxel diagonal
            vx = [(U.MAXI[i] - U.MINI[i]) / (res - 1) for i in range(3)]
            diag = float(np.sqrt(sum(v * v for v in vx)))
            v = ref.view(res // 4, 4, res // 4, 4, res // 2, 2).abs().amin(dim=(1, 3, 5))
            rec["near_tile_fraction"] = {f"L={L}": float((v < L * diag).float().mean()) for L in (1.0, 2.0, 3.0)}
            g = torch.gradient(ref.view(res, res, res), spacing=[float(x) for x in vx])
            gn = torch.sqrt(sum(x * x for x in g))
            nearm = ref.view(res, res, res).abs() < 2 * diag
            rec["grad_norm_near_surface"] = {"max": float(gn[nearm].max()), "q999": float(torch.quantile(gn[nearm][:4000000].float(), 0.999))}
        out[name] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
