#!/usr/bin/env python3
"""GPU marching cubes leg of the mesh extraction at 256^3: cold / warm wall time and its phases.  Development tool."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util as U
from nphm_amd import reconstruction as R
dev = torch.device("cuda:0")
net = U.build_identity(device=dev).eval()
lat = U.sample_latent(0).to(dev)
res = 256
vol = R.evaluate_grid(net, lat, R.grid_axes(U.MINI, U.MAXI, res), hack_chunk=25000).view(res, res, res)
torch.cuda.synchronize()
for it in range(6):
    t0 = time.perf_counter()
    v, f = R.marching_cubes_device(vol, 0.0, negate=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    vh = R.to_host(v); t2 = time.perf_counter()
    fh = R.to_host(f); t3 = time.perf_counter()
    print(f"call {it}: extract {1e3*(t1-t0):.2f} ms, verts d2h {1e3*(t2-t1):.2f} ms ({vh.nbytes/1e6:.1f} MB), faces d2h {1e3*(t3-t2):.2f} ms ({fh.nbytes/1e6:.1f} MB)")
