#!/usr/bin/env python3
"""Same-box timing of the dense skip-MLP lattice kernel per numerics tier (three- / two- / single-term, pinned), with the
error of each tier against the three-term product on a 96^3 lattice.  Run once per schedule:
NPHM_AMD_MLP_ASYM=0|1 python tools/ab_mlp.py   (development tool; bench.py is the contract benchmark)"""
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
    ts = []
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--npm-res", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--net", default="both")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    out = {"asym": os.environ.get("NPHM_AMD_MLP_ASYM", "1")}
    nets = []
    if args.net in ("both", "def"):
        g = U.golden("deformation")
        dnet = U.build_deformation(device=dev).eval()
        lat_ex = torch.from_numpy(g["lat"].reshape(-1)).to(dev)
        anc = torch.from_numpy(g["anchors"]).to(dev)
        mlp, cond = R._expr_condition(dnet, lat_ex, anc, dev)
        nets.append(("deformation", mlp, cond, args.res, True, 2 * 1_074_688))
    if args.net in ("both", "npm"):
        gn = U.golden("npm")
        npm = U.build_npm(device=dev).eval()
        nets.append(("npm", npm, torch.from_numpy(gn["lat"][None]).to(dev), args.npm_res, False, 2 * 6_292_480))
    for name, mlp, cond, res, add, flops in nets:
        axes = R.grid_axes(U.MINI, U.MAXI, res)
        axes_s = R.grid_axes(U.MINI, U.MAXI, 96 if name == "deformation" else 48)
        mlp.numerics = "fixed"
        mlp.single_mask, mlp.two_pass_mask = 0, 0
        ref = R.evaluate_grid_mlp(mlp, cond, axes_s, add_input=add).clone()
        rec = {}
        for tier, terms in (("three", 3), ("two", 2), ("single", 1)):
            mlp.single_mask = mlp._hidden_mask() if tier == "single" else 0
            mlp.two_pass_mask = mlp._hidden_mask() if tier == "two" else 0
            err = float((R.evaluate_grid_mlp(mlp, cond, axes_s, add_input=add) - ref).abs().max())
            ms = timeit(lambda: R.evaluate_grid_mlp(mlp, cond, axes, add_input=add), steps=args.steps)
            n = res ** 3
            rec[tier] = {"ms": round(ms, 3), "Mpts/s": round(n / ms / 1e3, 1), "err_vs_three": err,
                         "tflops_exec": round(flops * terms * n / ms / 1e9, 1)}
        # the calibrated default
        mlp.numerics, mlp.single_mask, mlp.two_pass_mask = "auto", 0, 0
        mlp._two_pass_cache = None
        R.evaluate_grid_mlp(mlp, cond, axes, add_input=add)
        rep = dict(mlp.last_numerics or {})
        err = float((R.evaluate_grid_mlp(mlp, cond, axes_s, add_input=add) - ref).abs().max()) if res > 64 else None
        ms = timeit(lambda: R.evaluate_grid_mlp(mlp, cond, axes, add_input=add), steps=args.steps)
        rec["auto"] = {"ms": round(ms, 3), "Mpts/s": round(res ** 3 / ms / 1e3, 1), "two_mask": rep.get("mask"), "single_mask": rep.get("single_mask"),
                       "sample_err": rep.get("err"), "all_single_err": rep.get("all_single_err"), "per_layer_single_err": rep.get("per_layer_single_err"),
                       "err_vs_three_small_lattice": err}
        out[name] = rec
    print(json.dumps(out))


if __name__ == "__main__":
    main()
