"""Time one training step of the identity decoder (training.py:110-135 with the geometry terms of
loss_functions.py:20-110) on the composite PyTorch tier and on the HIP training tier.
Usage: python tools/bench_train.py [B] [n_face] [steps]     (nphm.yaml: 32, 750)"""
import json
import sys
import time

import torch

import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util as U                                     # noqa: E402
from nphm_amd.loss_functions import actual_compute_loss, weighted_total   # noqa: E402


LAMBDAS = {"lat_reg": 0.01, "surf_sdf": 2.0, "normals": 0.3, "space_sdf": 0.01, "grad": 0.1, "anchors": 7.5,
           "symm_dist": 0.01, "middle_dist": 0.0}                     # scripts/configs/nphm.yaml


def synthetic_batch(B, n_face, dev, seed=0):
    """Point sets with the layout of face_dataset.py:93-123 (n_face on-surface face points, n_face/15 non-face,
    as many near-surface points, n_face/8 far points)."""
    g = torch.Generator().manual_seed(seed)
    n_non, n_far = max(n_face // 15, 1), max(n_face // 8, 1)
    box = torch.tensor([0.5, 0.6, 0.5])
    pts = lambda n: ((torch.rand(B, n, 3, generator=g) - 0.5) * box + torch.tensor([0.0, 0.05, 0.05]))
    nrm = lambda n: torch.nn.functional.normalize(torch.randn(B, n, 3, generator=g), dim=-1)
    face, non = pts(n_face), pts(n_non)
    far = nrm(n_far) * torch.rand(B, n_far, 1, generator=g) * 0.5
    near = torch.cat([face, non], 1) + torch.randn(B, n_face + n_non, 3, generator=g) * 0.01
    anchors = torch.from_numpy(U.anchors_mean()).float().reshape(1, 39, 3).repeat(B, 1, 1)
    batch = {"points_face": face, "normals_face": nrm(n_face), "points_non_face": non, "normals_non_face": nrm(n_non),
             "sup_grad_far": far, "sup_grad_near": near, "gt_anchors": anchors}
    return {k: v.float().to(dev) for k, v in batch.items()}


def step(net, lat, batch, opt):
    """training.py:110-135: zero_grad, loss, backward, clip, optimizer steps."""
    opt.zero_grad(set_to_none=True)
    losses = actual_compute_loss(batch, net, lat)
    loss = weighted_total(losses, LAMBDAS)             # (= sum(LAMBDAS[k] * losses[k] for k in losses), training.py:118-122, in 3 launches)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=0.1)
    opt.step()
    return loss.detach()


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    dev = torch.device("cuda:0")
    lat0 = torch.stack([U.sample_latent(10 + b) for b in range(B)])[:, None, :].to(dev)
    batch = synthetic_batch(B, N, dev)
    out = {"B": B, "N": N, "steps": steps}
    cases = (("hip", 1e-7), ("hip", -1.0), ("composite", 1e-7))
    if os.environ.get("NPHM_BENCH_TRAIN_ONLY"):
        cases = cases[:1]
    for backend, tol in cases:
        net = U.build_identity(device=dev).train()
        net.train_backend, net.prune_tol = backend, tol
        lat = lat0.clone().requires_grad_()
        opt = torch.optim.AdamW(list(net.parameters()) + [lat], lr=5e-4)
        torch.cuda.reset_peak_memory_stats()
        for _ in range(2):
            loss = step(net, lat, batch, opt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step(net, lat, batch, opt)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        out[f"{backend}_prune{tol:g}"] = {"ms_per_step": round(ms, 2), "loss": float(loss),
                                          "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2)}
        del net, opt
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
