"""A/B of the two forms of NPM's last hidden layer (two K halves with a workspace | two point halves), interleaved in one
process so that both see the same clocks: median / min HIP-event time of the 64^3 lattice launch per form.
usage (GPU box): python tools/npm_ab.py [rounds]"""
import sys, os, json
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _util as U
from nphm_amd import reconstruction as R

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
npm = U.build_npm(device=dev).eval()
gn = U.golden("npm")
lat = torch.from_numpy(gn["lat"][None]).to(dev)
axes = [torch.from_numpy(a).to(dev) for a in R.grid_axes(U.MINI, U.MAXI, 64)]
out = {}
with torch.no_grad():
    for k in (True, False):
        npm.tail_k_split = k
        for _ in range(3):
            R.evaluate_grid_mlp(npm, lat, axes)
    torch.cuda.synchronize()
    times = {True: [], False: []}
    for r in range(rounds):
        for k in (True, False) if r % 2 == 0 else (False, True):
            npm.tail_k_split = k
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); R.evaluate_grid_mlp(npm, lat, axes); b.record()
                torch.cuda.synchronize()
                times[k].append(a.elapsed_time(b))
print(json.dumps({"numerics": {k: v for k, v in (npm.last_numerics or {}).items() if k in ("mask", "single_mask", "tail_two_term")},
                  "k_halves_ms": {"median": float(np.median(times[True])), "min": min(times[True])},
                  "point_halves_ms": {"median": float(np.median(times[False])), "min": min(times[False])}}))
