"""A TRAINED-LIKE checkpoint of the NPM global DeepSDF (no released checkpoint can be fetched here, README.md:151): the
REFERENCE's own DeepSDF (src/NPHM/models/deepSDF.py:6-89; npm.yaml: lat 512, hidden 1024, 8 layers, geometric init) trained
on the CPU of the build container as an auto-decoder on the analytic head-like implicit surfaces of
tools/train_synthetic_heads.py (`Subjects.g`, scaled to distance-like units), L1 against the clamped field values + a code
regulariser (the DeepSDF recipe the NPM baseline of the paper follows; scripts/training/train.py builds the same module).

    python tools/train_synthetic_npm.py [--steps 1200] [--out gpurun_out/r4/trained_npm.npz]

~1.2 s per step on 8 cores.  tests/golden/make_golden_trained_npm.py turns the result into the fixture pair
trained_npm_state.npz / trained_npm.npz."""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
for missing in ("trimesh", "mcubes"):
    sys.modules.setdefault(missing, types.ModuleType(missing))

from NPHM.models.deepSDF import DeepSDF                     # noqa: E402
import train_synthetic_heads as H                            # noqa: E402  (the analytic subjects)

SDF_SCALE = 0.35          # g_s is |x / r| - 1 + bumps: times the mean radius it is distance-like near the surface
CLAMP = 0.1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1200)
    ap.add_argument("--subjects", type=int, default=16)
    ap.add_argument("--rows", type=int, default=8)
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--threads", type=int, default=7)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r4", "trained_npm.npz"))
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(3)
    dev = torch.device("cpu")
    subj = H.Subjects(args.subjects, dev, seed=0)
    net = DeepSDF(lat_dim=512, hidden_dim=1024, nlayers=8, geometric_init=True)
    net.train()
    codes = torch.nn.Embedding(args.subjects, 512)
    torch.nn.init.normal_(codes.weight, mean=0.0, std=0.1 / np.sqrt(512))
    opt = torch.optim.AdamW(net.parameters(), lr=5e-4, weight_decay=0.0)
    opt_z = torch.optim.Adam(codes.parameters(), lr=1e-3)
    lo, hi = torch.tensor([-0.5, -0.6, -0.55]), torch.tensor([0.5, 0.6, 0.45])
    trace, t0 = [], time.time()
    for step in range(args.steps):
        if step in (int(args.steps * 0.6), int(args.steps * 0.85)):
            for g in opt.param_groups + opt_z.param_groups:
                g["lr"] *= 0.4
        s_idx = torch.randint(0, args.subjects, (args.rows,), generator=gen)
        n = args.points
        x_u = torch.rand(args.rows, n // 4, 3, generator=gen) * (hi - lo) + lo
        surf, _ = subj.surface(s_idx, n - n // 4, gen)                                         # on-surface points [rows, ., 3]
        x_n = surf.detach() + 0.02 * torch.randn(surf.shape, generator=gen)                     # ... perturbed
        x = torch.cat([x_u, x_n], 1)
        with torch.no_grad():
            target = (SDF_SCALE * subj.g(s_idx, x))[..., None]
        z = codes(s_idx)
        pred, _ = net(x, z[:, None, :].repeat(1, n, 1))
        # (the target is clamped, the prediction is not: the geometric initialisation - a sphere of radius 1 around all samples -
        # starts outside the clamp everywhere, where a clamped prediction has no gradient)
        loss_sdf = (pred - target.clamp(-CLAMP, CLAMP)).abs().mean()
        loss_reg = (z ** 2).sum(-1).mean()
        loss = loss_sdf + 1e-4 * loss_reg
        opt.zero_grad(set_to_none=True)
        opt_z.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        opt_z.step()
        trace.append([float(loss_sdf), float(loss_reg)])
        if step % 25 == 0 or step == args.steps - 1:
            print(f"step {step:5d}  clamped |sdf error| {float(loss_sdf):.3e}  |z|^2 {float(loss_reg):.3e}  {time.time() - t0:.0f} s", flush=True)
    net.eval()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    wmax = max(float(v.abs().max()) for k, v in net.state_dict().items() if k.endswith("weight"))
    meta = dict(steps=args.steps, subjects=args.subjects, final_sdf_error=float(np.mean([t[0] for t in trace[-50:]])), max_weight=wmax)
    np.savez(args.out, **{"sd." + k: v.detach().numpy() for k, v in net.state_dict().items()}, codes=codes.weight.detach().numpy(),
             trace=np.asarray(trace, np.float32), meta=np.array(repr(meta)))
    print("saved", args.out, meta)


if __name__ == "__main__":
    main()
