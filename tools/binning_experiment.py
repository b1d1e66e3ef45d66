#!/usr/bin/env python3
"""Experiment (development tool): what would sorting the 4x4x2 voxel tiles by their member mask buy?
Feeds the 256^3 lattice to the POINT-LIST entry (workgroup = 8 consecutive 32-point tiles) in brick order
and in mask-sorted order.  Needs /tmp/tile_w.npy (tile masks, see profiles/NOTES.md section 4.1) - computed here
on the GPU with torch if absent."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util as U
from nphm_amd import reconstruction as R, _lib

dev = torch.device("cuda:0")
net = U.build_identity(device=dev).eval()
lat = U.sample_latent(0).to(dev)
res = 256
axes = [torch.from_numpy(a).to(dev) for a in R.grid_axes(U.MINI, U.MAXI, res)]
packed, state, anchors = net.prepare_latent(lat[None])
anc = anchors.reshape(-1, 3)

# tile masks with the kernel's budgeted rule (torch, fp32)
tol, T = 1e-7, res // 4
mult = torch.tensor([1, 2, 4, 8, 16, 40], dtype=torch.float32, device=dev)
tile_w = torch.zeros(T, T, res // 2, dtype=torch.int64, device=dev)
tile_h = torch.zeros_like(tile_w)
bits = (1 << torch.arange(40, device=dev, dtype=torch.int64))
for x0 in range(0, res, 4):
    P = torch.stack(torch.meshgrid(axes[0][x0:x0 + 4], axes[1], axes[2], indexing="ij"), -1).reshape(-1, 3)
    d = (P[:, None, :] - anc[None]).norm(dim=-1) + 1e-5
    w = torch.exp(-(d * d) / 0.01)
    wa = torch.cat([w, torch.full((len(w), 1), float(np.exp(-20.0)), device=dev)], 1)
    den = wa.sum(1) + 1e-6
    thr = tol * den
    cut = thr.clone()
    for t in range(1, 6):
        below = (wa * (wa <= mult[t] * thr[:, None])).sum(1)
        cut = torch.where(below <= 40 * thr, mult[t] * thr, cut)
    act = (wa > cut[:, None]).reshape(4, T, 4, res // 2, 2, 40).any(dim=4).any(dim=2).any(dim=0)
    tile_w[x0 // 4] = (act.long() * bits).sum(-1)
    hv = ((wa >= (1e-3 * den)[:, None]) & (wa > cut[:, None])).reshape(4, T, 4, res // 2, 2, 40).any(dim=4).any(dim=2).any(dim=0)
    tile_h[x0 // 4] = (hv.long() * bits).sum(-1)

# point list per tile: j -> (j >> 3, (j >> 1) & 3, j & 1)
tx, ty, tz = torch.meshgrid(torch.arange(T, device=dev), torch.arange(T, device=dev), torch.arange(res // 2, device=dev), indexing="ij")
def tile_points(order):
    ox, oy, oz = tx.reshape(-1)[order], ty.reshape(-1)[order], tz.reshape(-1)[order]
    j = torch.arange(32, device=dev)
    ix = ox[:, None] * 4 + (j >> 3)[None]; iy = oy[:, None] * 4 + ((j >> 1) & 3)[None]; iz = oz[:, None] * 2 + (j & 1)[None]
    return torch.stack([axes[0][ix], axes[1][iy], axes[2][iz]], -1).reshape(1, -1, 3).contiguous(), (ix * res + iy) * res + iz

# brick order: bricks of 2x2x2 tiles
b = ((tx // 2) * (T // 2) + (ty // 2)) * (res // 4) + (tz // 2)
inner = (tz % 2) * 4 + (ty % 2) * 2 + (tx % 2)
natural = torch.argsort((b * 8 + inner).reshape(-1))
sorted_ = torch.argsort(tile_w.reshape(-1), stable=True)
lib = _lib.load()
out = torch.empty(res ** 3, device=dev)
stats = torch.zeros(16, dtype=torch.int64, device=dev)
ref = None
o_h = torch.argsort(tile_h.reshape(-1), stable=True)
sorted_wh = o_h[torch.argsort(tile_w.reshape(-1)[o_h], stable=True)]
pc = lambda m: sum(((m >> i) & 1) for i in range(40))
o_w = torch.argsort(tile_w.reshape(-1), stable=True)
sorted_pc = o_w[torch.argsort(-pc(tile_w.reshape(-1)[o_w]), stable=True)]       # popcount descending, then mask
for name, order in (("brick order", natural), ("mask-sorted", sorted_), ("sorted (w,h)", sorted_wh), ("popcount desc", sorted_pc)):
    pts, lin = tile_points(order)
    n = pts.shape[1]
    def run():
        _lib.check(lib.nphm_identity_eval_points(packed.data_ptr(), state.data_ptr(), pts.data_ptr(), 1, n, 0, float(net.prune_tol),
                                                 net._precision_code(), out.data_ptr(), stats.data_ptr(), None), "eval_points")
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    vol = torch.empty(res ** 3, device=dev); vol[lin.reshape(-1)] = out
    if ref is None: ref = vol
    print(f"{name:12s}: {ms:7.2f} ms  ({n / ms / 1e3:.1f} Mpoints/s)   max |diff to brick order| {float((vol - ref).abs().max()):.2e}")
