#!/usr/bin/env python3
"""Debug aid: error of the dense MLP kernel's tiers against the reference fixture (points launch) and of the lattice launch
against the points launch, with the positions of the worst points.  NPHM_AMD_LIB / NPHM_AMD_MLP_ASYM select the build / schedule."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util as U  # noqa: E402
from nphm_amd import reconstruction as R  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g = U.golden("deformation")
    dnet = U.build_deformation(device=dev).eval()
    mlp = dnet.defDeepSDF
    xyz, lat, anc = [torch.from_numpy(np.ascontiguousarray(g[k])).to(dev) for k in ("xyz", "lat", "anchors")]
    tag = f"lib={os.path.basename(os.environ.get('NPHM_AMD_LIB', 'default'))} asym={os.environ.get('NPHM_AMD_MLP_ASYM', '1')}"
    mlp.numerics = "fixed"
    for tier in ("three", "two", "single"):
        mlp.single_mask = mlp._hidden_mask() if tier == "single" else 0
        mlp.two_pass_mask = mlp._hidden_mask() if tier == "two" else 0
        try:
            with torch.no_grad():
                off, _ = dnet(xyz, lat, anc)
        except Exception as e:
            print(tag, tier, "FAILED", str(e)[:100])
            continue
        err = np.abs(off.cpu().numpy() - g["offsets"]).max(-1)[0]
        bad = np.nonzero(err > 1e-4)[0]
        print(f"{tag} {tier:6s} golden max err {err.max():.3e}  n_bad {len(bad)} / {len(err)}  first bad {bad[:12].tolist()}  bad%128 hist "
              f"{np.bincount(bad % 128 // 16, minlength=8).tolist() if len(bad) else []}")
    # lattice launch vs points launch of the same tier (bitwise equal when both are right)
    lat_ex = torch.from_numpy(g["lat"].reshape(-1)).to(dev)
    m2, cond = R._expr_condition(dnet, lat_ex, anc, dev)
    axes = R.grid_axes(U.MINI, U.MAXI, 40)
    pts = torch.from_numpy(np.stack(np.meshgrid(*axes, indexing="ij"), -1).reshape(1, -1, 3).astype(np.float32)).to(dev)
    for tier in ("three", "two", "single"):
        mlp.single_mask = mlp._hidden_mask() if tier == "single" else 0
        mlp.two_pass_mask = mlp._hidden_mask() if tier == "two" else 0
        try:
            vol = R.evaluate_grid_mlp(m2, cond, axes, add_input=False)
            p = m2.forward_hip(pts, cond, add_input=False)
        except Exception as e:
            print(tag, tier, "FAILED", str(e)[:100])
            continue
        d = (vol.reshape(-1, 3) - p.reshape(-1, 3)).abs().max(-1)[0]
        print(f"{tag} {tier:6s} lattice vs points max diff {float(d.max()):.3e}")


if __name__ == "__main__":
    main()
