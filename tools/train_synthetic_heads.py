"""Train the NPHM identity decoder on ANALYTIC head-like surfaces (GPU, HIP training tier) to obtain a
trained-like checkpoint: no released checkpoint can be fetched here (README.md:151 is a Google-Drive link), and every
parity number would otherwise rest on seeded random-init weights only.

Data: S synthetic subjects, each an ellipsoid through the mean anchors plus Gaussian bumps (nose, brow, eye sockets,
chin, ears, neck), with the point-set layout of the reference's dataset (face_dataset.py:93-123: on-surface face /
non-face points with normals, near-surface points = surface + N(0, 0.01), far points uniform in a ball of radius
0.5, ground-truth anchors = the mean anchors projected onto the subject's surface).  Trainer: train_step of
training.py:110-135 (loss terms of loss_functions.py:20-110 weighted by nphm.yaml's lambdas, clip 0.1 / 0.1, AdamW
5e-4 / weight decay 0.01 on the decoder, Adam 1e-3 on the codes, codes initialised N(0, 0.1 / sqrt(1344)) and
re-normalised to norm <= 1 like nn.Embedding(max_norm=1)).

    python tools/train_synthetic_heads.py --steps 4000 --out gpurun_out/trained_heads.npz

Writes the decoder's state_dict (fp32), the S latent codes, the loss trace and the subjects' parameters."""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util as U                                            # noqa: E402
from nphm_amd.loss_functions import actual_compute_loss     # noqa: E402

LAMBDAS = {"lat_reg": 0.01, "surf_sdf": 2.0, "normals": 0.3, "space_sdf": 0.01, "grad": 0.1, "anchors": 7.5,
           "symm_dist": 0.01, "middle_dist": 0.0}                     # scripts/configs/nphm.yaml

CENTRE = torch.tensor([0.0, 0.0, -0.12])
RADII = torch.tensor([0.30, 0.47, 0.41])
# (position, amplitude, sigma): negative amplitude = outward bulge of the level set
BUMPS = [((0.0, -0.06, 0.30), -0.30, 0.065),      # nose
         ((0.0, 0.12, 0.27), -0.06, 0.10),        # brow
         ((-0.075, 0.07, 0.23), 0.09, 0.040),     # eye sockets
         ((0.075, 0.07, 0.23), 0.09, 0.040),
         ((0.0, -0.36, 0.17), -0.10, 0.085),      # chin
         ((0.0, -0.22, 0.24), -0.05, 0.05),       # lips
         ((-0.30, 0.0, -0.12), -0.16, 0.055),     # ears
         ((0.30, 0.0, -0.12), -0.16, 0.055),
         ((0.0, -0.50, -0.22), -0.35, 0.16)]      # neck


class Subjects:
    """S implicit surfaces g_s(x) = |(x - c) / r_s| - 1 + sum_b A_sb exp(-|x - p_sb|^2 / sigma_b^2) = 0."""

    def __init__(self, n, dev, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.n, self.dev = n, dev
        self.radii = (RADII[None] * (1 + 0.06 * torch.randn(n, 3, generator=g))).to(dev)
        pos = torch.tensor([b[0] for b in BUMPS])
        self.pos = (pos[None] + 0.012 * torch.randn(n, len(BUMPS), 3, generator=g)).to(dev)
        self.amp = (torch.tensor([b[1] for b in BUMPS])[None] * (1 + 0.3 * torch.randn(n, len(BUMPS), generator=g))).to(dev)
        self.sig = torch.tensor([b[2] for b in BUMPS]).to(dev)
        self.centre = CENTRE.to(dev)

    def g(self, s, x):
        """x [B,N,3], s [B] subject ids -> g [B,N]"""
        e = ((x - self.centre) / self.radii[s][:, None]).norm(dim=-1) - 1.0
        d2 = (x[:, :, None, :] - self.pos[s][:, None]).square().sum(-1)                    # [B,N,nb]
        return e + (self.amp[s][:, None] * torch.exp(-d2 / self.sig.square())).sum(-1)

    def project(self, s, dirs):
        """surface point on the ray from the centre along dirs [B,N,3] (unit): bisection in t"""
        lo = torch.full(dirs.shape[:2], 0.03, device=self.dev)
        hi = torch.full(dirs.shape[:2], 1.4, device=self.dev)
        for _ in range(34):
            mid = 0.5 * (lo + hi)
            inside = self.g(s, self.centre + mid[..., None] * dirs) < 0
            lo, hi = torch.where(inside, mid, lo), torch.where(inside, hi, mid)
        return self.centre + (0.5 * (lo + hi))[..., None] * dirs

    def normals(self, s, x):
        x = x.detach().requires_grad_()
        (gr,) = torch.autograd.grad(self.g(s, x).sum(), x)
        return torch.nn.functional.normalize(gr, dim=-1)

    def surface(self, s, n, gen):
        """n random surface points + normals per subject of s, area-agnostic (uniform ray directions)"""
        d = torch.nn.functional.normalize(torch.randn(len(s), n, 3, generator=gen, device=self.dev), dim=-1)
        p = self.project(s, d)
        return p, self.normals(s, p)


def is_face(p):
    """frontal region: what the reference's dataset calls 'face' (the rest: back of the head, neck)"""
    return (p[..., 2] > 0.02) & (p[..., 1] > -0.42) & (p[..., 1] < 0.32)


def make_pool(subj, n_pool, dev, seed=1):
    gen = torch.Generator(device=dev).manual_seed(seed)
    s = torch.arange(subj.n, device=dev)
    p, nrm = subj.surface(s, n_pool, gen)
    amean = torch.from_numpy(U.anchors_mean()).float().to(dev)
    adir = torch.nn.functional.normalize(amean - subj.centre, dim=-1)[None].repeat(subj.n, 1, 1)
    anchors = subj.project(s, adir)
    return {"points": p, "normals": nrm, "face": is_face(p), "anchors": anchors}


def draw_batch(pool, idx, n_face, gen, dev):
    """face_dataset.py:93-123 for the subjects idx [B]"""
    n_non, n_far = max((n_face // 3) // 5, 1), max(n_face // 8, 1)       # nphm.yaml: 750 face, 250 // 5 non-face
    out = {k: [] for k in ("points_face", "normals_face", "points_non_face", "normals_non_face")}
    for s in idx.tolist():
        f = pool["face"][s]
        pf, nf = pool["points"][s][f], pool["normals"][s][f]
        pn, nn = pool["points"][s][~f], pool["normals"][s][~f]
        i = torch.randint(0, pf.shape[0], (n_face,), generator=gen, device=dev)
        j = torch.randint(0, pn.shape[0], (n_non,), generator=gen, device=dev)
        out["points_face"].append(pf[i]); out["normals_face"].append(nf[i])
        out["points_non_face"].append(pn[j]); out["normals_non_face"].append(nn[j])
    b = {k: torch.stack(v) for k, v in out.items()}
    B = len(idx)
    u = torch.nn.functional.normalize(torch.randn(B, n_far, 3, generator=gen, device=dev), dim=-1)
    b["sup_grad_far"] = u * torch.rand(B, n_far, 1, generator=gen, device=dev) ** (1.0 / 3.0) * 0.5      # uniform_ball(rad 0.5)
    near = torch.cat([b["points_face"], b["points_non_face"]], 1)
    b["sup_grad_near"] = near + torch.randn(near.shape, generator=gen, device=dev) * 0.01
    b["gt_anchors"] = pool["anchors"][idx]
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--subjects", type=int, default=64)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--n-face", type=int, default=750)
    ap.add_argument("--pool", type=int, default=30000)
    ap.add_argument("--backend", default="hip", choices=["hip", "composite"])
    ap.add_argument("--out", default="gpurun_out/trained_heads.npz")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    subj = Subjects(args.subjects, dev)
    pool = make_pool(subj, args.pool, dev)
    print("pool: face fraction %.3f, anchors on surface |g| max %.2e" % (
        float(pool["face"].float().mean()), float(subj.g(torch.arange(subj.n, device=dev), pool["anchors"]).abs().max())), flush=True)

    net = U.build_identity(device=dev).train()
    net.train_backend = args.backend
    codes = torch.nn.Parameter(torch.randn(args.subjects, 1344, device=dev) * (0.1 / math.sqrt(1344)))
    opt = torch.optim.AdamW(net.parameters(), lr=5e-4, weight_decay=0.01)
    opt_lat = torch.optim.Adam([codes], lr=1e-3)
    gen = torch.Generator(device=dev).manual_seed(2)
    trace = []
    t0 = time.perf_counter()
    for it in range(args.steps):
        if it and it % 2500 == 0:                                  # training.py:93-108, compressed horizon
            for o in (opt, opt_lat):
                for pg in o.param_groups:
                    pg["lr"] *= 0.5
        idx = torch.randperm(args.subjects, generator=gen, device=dev)[:args.batch]
        batch = draw_batch(pool, idx, args.n_face, gen, dev)
        opt.zero_grad(set_to_none=True); opt_lat.zero_grad(set_to_none=True)
        losses = actual_compute_loss(batch, net, codes[idx][:, None, :])
        loss = sum(LAMBDAS[k] * losses[k] for k in losses)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=0.1)
        torch.nn.utils.clip_grad_norm_([codes], max_norm=0.1)
        opt.step(); opt_lat.step()
        with torch.no_grad():                                      # nn.Embedding(max_norm=1.0)
            n = codes.norm(dim=-1, keepdim=True)
            codes.mul_(torch.clamp(1.0 / (n + 1e-7), max=1.0))
        if it % 100 == 0 or it == args.steps - 1:
            rec = {k: float(v) for k, v in losses.items()}
            rec.update(step=it, loss=float(loss), t=time.perf_counter() - t0)
            trace.append(rec)
            print(json.dumps(rec), flush=True)
    torch.cuda.synchronize()
    print("trained %d steps in %.1f s" % (args.steps, time.perf_counter() - t0), flush=True)

    # quality of the fit: |sdf| on fresh surface points, gradient norm, weight scale
    net.eval()
    with torch.no_grad():
        s = torch.arange(min(8, args.subjects), device=dev)
        d = torch.nn.functional.normalize(torch.randn(len(s), 4000, 3, generator=gen, device=dev), dim=-1)
        p = subj.project(s, d)
        sdf, _ = net(p, codes[s][:, None, :], None)
        print("fresh surface points: mean |sdf| %.4e, max %.3e" % (float(sdf.abs().mean()), float(sdf.abs().max())))
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    for k, v in sd.items():
        if k.endswith("weight"):
            print("%-40s rms %.4f max %.3f" % (k, float(np.sqrt((v ** 2).mean())), float(np.abs(v).max())))
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez(args.out, **{"sd." + k: v for k, v in sd.items()}, codes=codes.detach().cpu().numpy(),
             trace=json.dumps(trace), radii=subj.radii.cpu().numpy(), bump_pos=subj.pos.cpu().numpy(),
             bump_amp=subj.amp.cpu().numpy(), anchors=pool["anchors"].cpu().numpy(),
             meta=json.dumps(vars(args)))
    print("wrote", args.out, os.path.getsize(args.out) / 2 ** 20, "MiB")


if __name__ == "__main__":
    main()
