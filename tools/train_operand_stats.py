#!/usr/bin/env python3
"""Magnitudes of the stored operands of the training tier's weight gradients (ident_train_kernel.hip: per tile [1005 rows][64
columns] fp32, column = 32 * stream + point; streams: value | tangent), per operand kind and stream: what a 16-bit storage
format has to represent.  Seeded and trained-like weights, nphm.yaml batch.  Development tool."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _util as U  # noqa: E402
import bench_train as BT  # noqa: E402

KINDS = [("IN1 h0|u0", 0, 200), ("IN2 h1|u1", 200, 104), ("IN3 h2|u2", 304, 200), ("D1|T1", 504, 101), ("D2|T2", 605, 200), ("D3|T3", 805, 200)]


def main():
    dev = torch.device("cuda:0")
    out = {}
    for name in ("seeded", "trained"):
        if name == "seeded":
            net = U.build_identity(device=dev).train()
            B = 8
            lat = torch.stack([U.sample_latent(10 + b) for b in range(B)])[:, None, :].to(dev).requires_grad_()
        else:
            net, codes = U.build_trained_identity(device=dev)
            net.train()
            B = 8
            lat = codes[:B, None, :].clone().to(dev).requires_grad_()
        for p in net.parameters():
            p.requires_grad_(True)
        net._keep_train_operands = True
        batch = BT.synthetic_batch(B, 750, dev)
        losses = BT.actual_compute_loss(batch, net, lat)
        loss = sum(BT.LAMBDAS[k] * losses[k] for k in losses)
        loss.backward()
        sv = net._last_train_operands.view(torch.float32).view(-1, 1005, 64)
        rec = {"tiles": int(sv.shape[0]), "seed_max_value_stream": net._last_train_seeds[0], "seed_max_tangent_stream": net._last_train_seeds[1]}
        for kind, r0, n in KINDS:
            blk = sv[:, r0:r0 + n]
            for s, nm in ((0, "value"), (1, "tangent")):
                a = blk[:, :, 32 * s:32 * s + 32].abs().reshape(-1)
                a = a[a > 0]
                q = torch.quantile(a[torch.randperm(a.numel(), device=dev)[:2000000]], torch.tensor([0.001, 0.5, 0.999], device=dev))
                rec[f"{kind} {nm}"] = {"max": float(a.max()), "q999": float(q[2]), "median": float(q[1])}
        out[name] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
