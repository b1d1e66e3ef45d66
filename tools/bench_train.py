"""Time one training step of the identity decoder (training.py:110-135 with the geometry terms of
loss_functions.py:20-110) on the composite PyTorch tier and on the HIP training tier.
Usage: python tools/bench_train.py [B] [N] [steps]"""
import json
import sys
import time

import torch

sys.path.insert(0, "tests")
import _util as U                                     # noqa: E402
from nphm_amd.diff_operators import gradient          # noqa: E402


def step(net, lat, xyz, nrm, opt):
    opt.zero_grad(set_to_none=True)
    x = xyz.clone().requires_grad_()
    pred, anchors = net(x, lat.repeat(1, x.shape[1], 1), None)
    grad = gradient(pred, x)
    loss = (2.0 * pred.abs().mean() + 0.3 * (grad - nrm).norm(2, dim=-1).mean() + 0.1 * (grad.norm(dim=-1) - 1).abs().mean()
            + 0.01 * torch.exp(-1e1 * pred.abs()).mean() + 7.5 * anchors.square().mean() + 0.01 * (lat.norm(dim=-1) ** 2).mean())
    loss.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=0.1)
    opt.step()
    return loss


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    lat0 = torch.stack([U.sample_latent(10 + b) for b in range(B)])[:, None, :].to(dev)
    xyz = ((torch.rand(B, N, 3, generator=g) - 0.5) * torch.tensor([0.7, 0.9, 0.7])).to(dev)
    nrm = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=-1).to(dev)
    out = {"B": B, "N": N, "steps": steps}
    for backend, tol in (("hip", 1e-7), ("hip", -1.0), ("composite", 1e-7)):
        net = U.build_identity(device=dev).train()
        net.train_backend, net.prune_tol = backend, tol
        lat = lat0.clone().requires_grad_()
        opt = torch.optim.AdamW(list(net.parameters()) + [lat], lr=5e-4)
        torch.cuda.reset_peak_memory_stats()
        for _ in range(2):
            loss = step(net, lat, xyz, nrm, opt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step(net, lat, xyz, nrm, opt)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        out[f"{backend}_prune{tol:g}"] = {"ms_per_step": round(ms, 2), "loss": float(loss),
                                          "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2)}
        del net, opt
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
