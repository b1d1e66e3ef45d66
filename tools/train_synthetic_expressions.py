"""A TRAINED-LIKE checkpoint of the forward-deformation network (no released checkpoint can be fetched here, README.md:151):
the REFERENCE's own DeformationNetwork (src/NPHM/models/deepSDF.py:118-239, mode 'compress', nphm_def.yaml sizes) trained
on the CPU, in the build container, against analytic expression warps of the synthetic heads of
tools/train_synthetic_heads.py - the recipe of scripts/training/train_corresp.py in small: the identity decoder and its codes
are frozen (tests/golden/trained_state.npz), the deformation network and one expression code per (subject, expression)
are trained (auto-decoder, Adam) to reproduce the canonical -> posed offsets (loss_functions.py:282-322: an L1 on the
offsets plus a code regulariser).

Warps: expression e of subject s displaces the canonical point x by
    D(x) = sum_j  a_j  exp(-|x - c_j|^2 / (2 sigma_j^2))  u_j ,    c_j = that subject's anchor m_j + a small offset,
four Gaussian bumps per expression around mouth / jaw / cheek / brow anchors, amplitudes up to 4 cm of the unit head
(0.04), widths 0.06 .. 0.16, directions mostly along -y / z (jaw drop, pout) - smooth, localised, up to 0.06 in magnitude,
i.e. what a fitted expression looks like to the kernels: offsets of a few 1e-2 with gradients of order 0.5.

    python tools/train_synthetic_expressions.py [--steps 2500] [--out gpurun_out/r4/trained_expr.npz]

~0.4 s per step on 8 cores.  The result is turned into the fixture pair tests/golden/trained_def_state.npz / trained_def.npz
by tests/golden/make_golden_trained_def.py."""
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
for missing in ("trimesh", "mcubes"):
    sys.modules.setdefault(missing, types.ModuleType(missing))

from NPHM.models.deepSDF import DeformationNetwork                     # noqa: E402
from NPHM.models.EnsembledDeepSDF import FastEnsembleDeepSDFMirrored  # noqa: E402

N_EXPR = 12            # expressions per subject (0 = neutral: zero warp)
BUMP_ANCHORS = (30, 31, 32, 33, 34, 35, 36, 37, 38, 4, 5, 10, 11, 20, 21)   # mid-line + a few paired anchors


def expression_bank(gen):
    """[N_EXPR, 4] bumps: (anchor index, offset [3], sigma, amplitude vector [3]); expression 0 is neutral"""
    bank = []
    for e in range(N_EXPR):
        bumps = []
        for j in range(4):
            a = BUMP_ANCHORS[int(torch.randint(0, len(BUMP_ANCHORS), (1,), generator=gen))]
            off = (torch.rand(3, generator=gen) - 0.5) * 0.06
            sigma = 0.06 + 0.10 * float(torch.rand(1, generator=gen))
            u = torch.randn(3, generator=gen)
            u = u / u.norm()
            u = 0.5 * u + torch.tensor([0.0, -0.6, 0.4]) * (1.0 if j < 2 else 0.2)
            amp = (0.0 if e == 0 else 0.01 + 0.03 * float(torch.rand(1, generator=gen))) * u / u.norm()
            bumps.append((a, off, sigma, amp))
        bank.append(bumps)
    return bank


def warp(bank, e, anchors_s, x):
    """D_e(x) for the subject with anchors anchors_s [39,3]; x [n,3]"""
    d = torch.zeros_like(x)
    for a, off, sigma, amp in bank[e]:
        c = anchors_s[a] + off
        w = torch.exp(-((x - c) ** 2).sum(-1, keepdim=True) / (2 * sigma * sigma))
        d = d + w * amp
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2500)
    ap.add_argument("--subjects", type=int, default=16)
    ap.add_argument("--pairs", type=int, default=8, help="(subject, expression) pairs per step")
    ap.add_argument("--points", type=int, default=1500)
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r4", "trained_expr.npz"))
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(7)

    ck = np.load(os.path.join(ROOT, "tests", "golden", "trained_state.npz"))
    sd = {k[3:]: torch.from_numpy(ck[k]) for k in ck.files if k.startswith("sd.")}
    codes = torch.from_numpy(ck["codes"]).float()[:args.subjects]                  # frozen identity codes
    mean_anchors = torch.from_numpy(np.load(os.path.join(REF, "assets", "anchors_39.npy"))).float()[None, None]
    ident = FastEnsembleDeepSDFMirrored(lat_dim_glob=64, lat_dim_loc=32, n_loc=39, n_symm_pairs=16, anchors=mean_anchors,
                                        hidden_dim=200, n_layers=4, pos_mlp_dim=256)
    ident.load_state_dict(sd, strict=True)
    ident.eval()
    with torch.no_grad():
        anchors = (ident.mlp_pos(codes[:, :64]).view(-1, 39, 3) + mean_anchors[0])   # [S,39,3] (EnsembledDeepSDF.py:228-229)

    net = DeformationNetwork(mode="compress", lat_dim_expr=200, lat_dim_id=32, lat_dim_glob_shape=64, lat_dim_loc_shape=32,
                             n_loc=39, anchors=mean_anchors, hidden_dim=512, nlayers=6, input_dim=3, out_dim=3)
    net.train()
    S = codes.shape[0]
    z_ex = torch.nn.Embedding(S * N_EXPR, 200)
    torch.nn.init.normal_(z_ex.weight, mean=0.0, std=0.01)                         # training_corresp.py:69-73
    opt = torch.optim.AdamW(net.parameters(), lr=5e-4, weight_decay=0.0)
    opt_z = torch.optim.Adam(z_ex.parameters(), lr=2e-3)
    bank = expression_bank(gen)
    lo, hi = torch.tensor([-0.45, -0.50, -0.45]), torch.tensor([0.45, 0.55, 0.45])
    trace = []
    t0 = time.time()
    for step in range(args.steps):
        if step in (int(args.steps * 0.6), int(args.steps * 0.85)):
            for g in opt.param_groups + opt_z.param_groups:
                g["lr"] *= 0.4
        s_idx = torch.randint(0, S, (args.pairs,), generator=gen)
        e_idx = torch.randint(0, N_EXPR, (args.pairs,), generator=gen)
        n = args.points
        # half uniform in the head box, half around the subject's anchors (where the warps live)
        x_u = torch.rand(args.pairs, n // 2, 3, generator=gen) * (hi - lo) + lo
        pick = torch.randint(0, 39, (args.pairs, n - n // 2), generator=gen)
        x_a = anchors[s_idx][torch.arange(args.pairs)[:, None], pick] + 0.08 * torch.randn(args.pairs, n - n // 2, 3, generator=gen)
        x = torch.cat([x_u, x_a], 1)
        target = torch.stack([warp(bank, int(e_idx[b]), anchors[int(s_idx[b])], x[b]) for b in range(args.pairs)])
        z = z_ex(s_idx * N_EXPR + e_idx)                                            # [P,200]
        lat = torch.cat([codes[s_idx], z], -1)[:, None, :].repeat(1, n, 1)          # [P,n,1544] as train_corresp feeds it
        pred, _ = net(x, lat, anchors[s_idx])
        loss_off = (pred - target).abs().mean()
        loss_reg = (z ** 2).sum(-1).mean()
        loss = loss_off + 1e-4 * loss_reg
        opt.zero_grad(set_to_none=True)
        opt_z.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        opt_z.step()
        trace.append([float(loss_off), float(loss_reg)])
        if step % 50 == 0 or step == args.steps - 1:
            print(f"step {step:5d}  |offset error| {float(loss_off):.3e}  (target mean |D| {float(target.abs().mean()):.3e}, "
                  f"max {float(target.norm(dim=-1).max()):.3e})  |z|^2 {float(loss_reg):.3e}  {time.time() - t0:.0f} s", flush=True)
    net.eval()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    out = {"sd." + k: v.detach().numpy() for k, v in net.state_dict().items()}
    wmax = max(float(v.abs().max()) for k, v in net.state_dict().items() if k.endswith("weight"))
    meta = dict(steps=args.steps, subjects=S, n_expr=N_EXPR, final_offset_error=float(np.mean([t[0] for t in trace[-50:]])),
                max_weight=wmax)
    np.savez(args.out, **out, z_ex=z_ex.weight.detach().numpy(), subject_codes=codes.numpy(), subject_anchors=anchors.numpy(),
             trace=np.asarray(trace, np.float32), meta=np.array(repr(meta)))
    print("saved", args.out, meta)


if __name__ == "__main__":
    main()
