"""Which Python line issues which aten op in one training step of the identity decoder on the HIP training tier
(development tool, GPU): every non-view aten op with its innermost repo frame, per step.

    python tools/train_step_ops.py [n_steps]"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import _util as U                     # noqa: E402
import bench_train as BT              # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = torch.device("cuda:0")
    B = 32
    batch = BT.synthetic_batch(B, 750, dev)
    lat = torch.stack([U.sample_latent(10 + b) for b in range(B)])[:, None, :].to(dev).requires_grad_()
    net = U.build_identity(device=dev).train()
    opt = torch.optim.AdamW(list(net.parameters()) + [lat], lr=5e-4, weight_decay=0.01)
    for _ in range(2):
        BT.step(net, lat, batch, opt)
    torch.cuda.synchronize()
    from torch.utils._python_dispatch import TorchDispatchMode
    sites = collections.Counter()
    skip = ("aten.view", "aten.detach", "aten.expand", "aten.slice", "aten.select", "aten.unsqueeze", "aten.squeeze",
            "aten.transpose", "aten.t.", "aten.alias", "aten._unsafe_view", "aten.reshape", "aten.permute", "aten.unbind",
            "aten.empty", "aten.as_strided", "aten.lift_fresh", "aten.is_pinned", "aten.split", "aten.new_empty")

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if not name.startswith(skip):
                site = "autograd / no repo frame"
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if fr.filename.startswith(ROOT) and "train_step_ops" not in fr.filename:
                        site = f"{fr.filename.replace(ROOT + '/', '')}:{fr.lineno} {fr.name}"
                        break
                shapes = tuple(tuple(a.shape) for a in args if isinstance(a, torch.Tensor))[:2]
                sites[(name, site, str(shapes))] += 1
            return func(*args, **(kwargs or {}))

    with Spy():
        for _ in range(n):
            BT.step(net, lat, batch, opt)
        torch.cuda.synchronize()
    print(f"{sum(sites.values()) / n:.1f} non-view aten ops per step")
    for (name, site, shapes), c in sorted(sites.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print(f"{c / n:5.1f}x {name[:34]:34s} {site[:72]:72s} {shapes[:70]}")


if __name__ == "__main__":
    main()
