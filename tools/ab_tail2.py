#!/usr/bin/env python3
"""The TAIL2 variant of mlp_kernel.hip (single-term layers, the last hidden layer two-term in two point halves) against the
generic kernel with the same masks: run once per library (NPHM_AMD_LIB), `--save out.npz`; then `--compare a.npz b.npz` (the
arithmetic per point is the same: bitwise).  Also times the NPM 64^3 lattice in the tiers.  Development tool."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    if sys.argv[1] == "--compare":
        a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
        for k in a.files:
            print(k, "bitwise" if np.array_equal(a[k], b[k]) else f"max diff {np.abs(a[k] - b[k]).max():.3e}")
        return
    import _util as U
    import nphm_amd
    from nphm_amd import reconstruction as R
    dev = torch.device("cuda:0")
    out = {}
    rec = {}
    g = torch.Generator().manual_seed(3)
    for name, net, lat in (("npm", U.build_npm(device=dev).eval(), None), ("deformation", U.build_deformation(device=dev).eval().defDeepSDF, None)):
        cond = (torch.randn(1, net.lat_dim, generator=g) * 0.05).to(dev)
        x = ((torch.rand(1, 5000 + 77, 3, generator=g) - 0.5) * 0.9).to(dev)
        hid = net._hidden_mask()
        tail = 1 << (net.nlayers - 1)
        net.numerics = "fixed"
        axes = R.grid_axes(U.MINI, U.MAXI, 64 if name == "npm" else 128)
        with torch.no_grad():
            for tier, two, one in (("three", 0, 0), ("two", hid, 0), ("single", 0, hid), ("tail", tail, hid & ~tail)):
                net.two_pass_mask, net.single_mask = two, one
                out[f"{name}_{tier}"] = net.forward_hip(x, cond).cpu().numpy()
                f = lambda: R.evaluate_grid_mlp(net, cond, axes)
                f(); f(); torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); f(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
                rec[f"{name}_{tier}"] = {"ms": round(float(np.median(ts)), 3), "err_vs_three": float(np.abs(out[f"{name}_{tier}"] - out[f"{name}_three"]).max())}
            net.numerics, net.two_pass_mask, net.single_mask = "auto", 0, 0
            R.evaluate_grid_mlp(net, cond, axes)
            rec[f"{name}_auto"] = {k: v for k, v in (net.last_numerics or {}).items() if k in ("single_mask", "mask", "err", "tail_two_term", "single_term", "all_single_err")}
    print(json.dumps(rec, indent=1))
    np.savez(sys.argv[2], **out)


if __name__ == "__main__":
    main()
