#!/usr/bin/env python3
"""Strong-scaling load balance of the multi-GPU partition, measured on ONE GPU: time every rank's
share of the 256^3 extraction separately (contiguous equal x-slabs vs the cyclic 8-plane partition)
and report max/mean — the factor by which the slowest rank limits an N-GPU run.  Development tool."""
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

dev = torch.device("cuda:0")
net = U.build_identity(device=dev).eval()
lat = U.sample_latent(0).to(dev)
res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
axes = R.grid_axes(U.MINI, U.MAXI, res)


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


out = {"single_ms": timeit(lambda: R.evaluate_grid(net, lat, axes, hack_chunk=25000))}
for world in (2, 4, 8):
    cont = [timeit(lambda r=r: R.evaluate_grid(net, lat, axes, hack_chunk=25000, x_range=R.slab_bounds(res, world, r)))
            for r in range(world)]
    cyc = [timeit(lambda r=r: R.evaluate_grid(net, lat, axes, hack_chunk=25000,
                                               x_planes=torch.from_numpy(R.cyclic_planes(res, world, r)).to(dev)))
           for r in range(world)]
    out[world] = {"contiguous_ms": [round(t, 2) for t in cont], "contiguous_max_over_mean": max(cont) / np.mean(cont),
                  "cyclic_ms": [round(t, 2) for t in cyc], "cyclic_max_over_mean": max(cyc) / np.mean(cyc),
                  "cyclic_max_ms": max(cyc), "ideal_speedup_cyclic": out["single_ms"] / max(cyc),
                  "ideal_speedup_contiguous": out["single_ms"] / max(cont)}
print(json.dumps(out, indent=1))
