#!/usr/bin/env python3
"""Timing of the fused dense skip-MLP kernels (deformation backbone, NPM SDF) and of the two-stage
lattice evaluation on one GPU.  Development tool: bench.py is the contract benchmark."""
import argparse
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


def timeit(fn, warmup=2, steps=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--npm-res", type=int, default=128)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = U.golden("deformation")
    dnet = U.build_deformation(device=dev).eval()
    inet = U.build_identity(device=dev).eval()
    npm = U.build_npm(device=dev).eval()
    axes = R.grid_axes(U.MINI, U.MAXI, args.res)
    lat_id = torch.from_numpy(g["lat"].reshape(-1)[:1344]).to(dev)
    lat_ex = torch.from_numpy(g["lat"].reshape(-1)).to(dev)
    anc = torch.from_numpy(g["anchors"]).to(dev)
    mlp, cond = R._expr_condition(dnet, lat_ex, anc, dev)
    n = args.res ** 3
    out = {}
    ms = timeit(lambda: R.evaluate_grid_mlp(mlp, cond, axes, add_input=True))
    flops = 2 * 1_074_688 * 3
    out["deformation"] = {"ms": ms, "Mpts/s": n / ms / 1e3, "mfma_tflops_3pass": flops * n / ms / 1e9,
                          "frac_of_2.5PF": flops * n / ms / 1e9 / 2500}
    ms = timeit(lambda: R.evaluate_grid_two_stage(inet, dnet, lat_id, lat_ex, axes, hack_chunk=25000))
    out["two_stage"] = {"ms": ms, "Mpts/s": n / ms / 1e3}
    ms = timeit(lambda: R.evaluate_grid(inet, lat_id, axes, hack_chunk=25000))
    out["identity"] = {"ms": ms, "Mpts/s": n / ms / 1e3}
    gn = U.golden("npm")
    axes_n = R.grid_axes(U.MINI, U.MAXI, args.npm_res)
    nn = args.npm_res ** 3
    latn = torch.from_numpy(gn["lat"][None]).to(dev)
    ms = timeit(lambda: R.evaluate_grid_mlp(npm, latn, axes_n))
    flops = 2 * 6_292_480 * 3
    out["npm"] = {"res": args.npm_res, "ms": ms, "Mpts/s": nn / ms / 1e3, "mfma_tflops_3pass": flops * nn / ms / 1e9,
                  "frac_of_2.5PF": flops * nn / ms / 1e9 / 2500}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
