#!/usr/bin/env python3
"""Per-tensor gradient deviation of the HIP training tier from the composite (PyTorch autograd) tier on the 4 x 1000-point
batch of tests/test_hip_train.py, for the operand storage modes and pruning budgets.  Development tool."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util as U  # noqa: E402
import test_hip_train as T  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    out = {}
    for prune in (-1.0, 1e-7):
        net = U.build_identity(device=dev).train()
        net.prune_tol = prune
        lat, xyz, nrm = T._batch(dev, B=4, N=1000)
        ref = T._run(net, "composite", lat, xyz, nrm)
        for ops in ("f32", "bf16") + (("f16",) if "f16" in os.environ.get("NPHM_TRAIN_MODES", "") else ()):
            net.train_operands = ops
            res = T._run(net, "hip", lat, xyz, nrm)
            worst = {k: T._rel(res[k], ref[k]) for k in ref}
            top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
            out[f"prune {prune} operands {ops}"] = {"max": max(worst.values()), "worst": {k: float(f"{v:.3e}") for k, v in top}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
