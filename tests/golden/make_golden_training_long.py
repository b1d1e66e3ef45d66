"""Golden TRACE of the identity decoder's training loop, produced by RUNNING THE REFERENCE (PyTorch-CPU fp32):
120 steps of TrainerAutoDecoder.train_step (src/NPHM/models/training.py:110-135) - compute_loss
(loss_functions.py:7-110) on the reference's FastEnsembleDeepSDFMirrored, loss weights / clipping / AdamW 5e-4 with
weight decay 0.01 of nphm.yaml, latent codes in an nn.Embedding(max_norm=1) under Adam 1e-3, initialised
N(0, 0.1 / sqrt(1344)) (training.py:27-52; the reference's sparse Embedding + SparseAdam cannot run under this container's
torch 2.10 - clip_grad_norm_ rejects sparse gradients - and is the same update here: every step touches all codes).

    python tests/golden/make_golden_training_long.py        ->  tests/golden/training_long.npz   (~10 min on 8 cores)

Data: 4 analytic head-like subjects (tools/train_synthetic_heads.py: ellipsoid + bumps, anchors on the surface), a
fixed pool of surface points per subject generated on the CPU with a seeded generator, one batch of all 4 subjects per
step drawn with the layout of face_dataset.py:93-123 at reduced size (120 face + 8 non-face + 128 near + 15 far points
per subject).  The pool and every step's draw are stored, so the GPU test replays the identical batches.
Stored per step: the loss dictionary and the total; every 20 steps the norm of every parameter tensor and of the codes."""
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.dirname(HERE))
for missing in ("trimesh", "mcubes"):
    sys.modules.setdefault(missing, types.ModuleType(missing))

import make_golden as G                                          # noqa: E402
from NPHM.models.loss_functions import compute_loss             # noqa: E402

LAMBDAS = {"lat_reg": 0.01, "surf_sdf": 2.0, "normals": 0.3, "space_sdf": 0.01, "grad": 0.1, "anchors": 7.5,
           "symm_dist": 0.01, "middle_dist": 0.0}                # scripts/configs/nphm.yaml:24-32
N_SUBJECTS, N_STEPS, N_FACE, POOL, SNAP = 4, 120, 120, 1500, 20


def make_pool():
    """surface pool of the 4 subjects on the CPU (seeded; stored in the fixture)"""
    import train_synthetic_heads as T
    dev = torch.device("cpu")
    subj = T.Subjects(N_SUBJECTS, dev, seed=5)
    gen = torch.Generator().manual_seed(6)
    s = torch.arange(N_SUBJECTS)
    p, nrm = subj.surface(s, POOL, gen)
    amean = torch.from_numpy(np.load(os.path.join(HERE, "anchors_mean_39.npy"))).float()
    adir = torch.nn.functional.normalize(amean - subj.centre, dim=-1)[None].repeat(N_SUBJECTS, 1, 1)
    return {"points": p.detach(), "normals": nrm.detach(), "face": T.is_face(p), "anchors": subj.project(s, adir).detach()}


def draw(pool, gen):
    """one batch of all subjects (face_dataset.py:93-123 at reduced size); returns the batch and the drawn indices"""
    n_non, n_far = max((N_FACE // 3) // 5, 1), max(N_FACE // 8, 1)
    out = {k: [] for k in ("points_face", "normals_face", "points_non_face", "normals_non_face")}
    for s in range(N_SUBJECTS):
        f = pool["face"][s]
        pf, nf, pn, nn = pool["points"][s][f], pool["normals"][s][f], pool["points"][s][~f], pool["normals"][s][~f]
        i = torch.randint(0, pf.shape[0], (N_FACE,), generator=gen)
        j = torch.randint(0, pn.shape[0], (n_non,), generator=gen)
        out["points_face"].append(pf[i]); out["normals_face"].append(nf[i])
        out["points_non_face"].append(pn[j]); out["normals_non_face"].append(nn[j])
    b = {k: torch.stack(v) for k, v in out.items()}
    u = torch.nn.functional.normalize(torch.randn(N_SUBJECTS, n_far, 3, generator=gen), dim=-1)
    b["sup_grad_far"] = u * torch.rand(N_SUBJECTS, n_far, 1, generator=gen) ** (1.0 / 3.0) * 0.5
    near = torch.cat([b["points_face"], b["points_non_face"]], 1)
    b["sup_grad_near"] = near + torch.randn(near.shape, generator=gen) * 0.01
    b["gt_anchors"] = pool["anchors"]
    b["idx"] = torch.arange(N_SUBJECTS)[:, None]
    return b


def snapshot(net, codes):
    names = sorted(n for n, _ in net.named_parameters())
    sd = dict(net.named_parameters())
    return names, np.array([float(sd[n].detach().norm()) for n in names] + [float(codes.weight.detach().norm())])


def main():
    torch.set_num_threads(8)
    net, _ = G.build_identity()
    net.train()
    dev = torch.device("cpu")
    torch.manual_seed(11)
    codes = torch.nn.Embedding(N_SUBJECTS, 1344, max_norm=1.0)
    torch.nn.init.normal_(codes.weight.data, 0.0, 0.1 / math.sqrt(1344))
    opt = torch.optim.AdamW(params=list(net.parameters()), lr=5e-4, weight_decay=0.01)
    opt_lat = torch.optim.Adam(list(codes.parameters()), lr=1e-3)
    pool = make_pool()
    gen = torch.Generator().manual_seed(7)
    keys = list(LAMBDAS)
    trace = np.zeros((N_STEPS, len(keys) + 1), np.float64)
    snaps, batches = [], []
    codes0 = codes.weight.detach().clone().numpy()
    for it in range(N_STEPS):
        if it % SNAP == 0:
            names, v = snapshot(net, codes)
            snaps.append(v)
        batch = draw(pool, gen)
        batches.append({k: v.numpy().copy() for k, v in batch.items() if k not in ("gt_anchors", "idx")})
        opt.zero_grad(); opt_lat.zero_grad()
        losses = compute_loss(batch, net, codes, dev)                      # training.py:116
        total = 0
        for k in losses:
            total = total + LAMBDAS[k] * losses[k]
        total.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=0.1)
        torch.nn.utils.clip_grad_norm_(codes.parameters(), max_norm=0.1)
        opt.step(); opt_lat.step()
        trace[it] = [float(losses[k]) for k in keys] + [float(total)]
        if it % 10 == 0:
            print(it, {k: round(float(losses[k]), 5) for k in keys}, round(float(total), 5), flush=True)
    names, v = snapshot(net, codes)
    snaps.append(v)
    out = {"state_hash_init": np.array(G.state_hash(G.build_identity()[0])), "keys": np.array(keys), "trace": trace,
           "snap_names": np.array(names + ["latent_codes"]), "snaps": np.stack(snaps), "snap_every": np.array(SNAP),
           "codes_init": codes0, "codes_final": codes.weight.detach().numpy(),
           "pool_anchors": pool["anchors"].numpy()}
    for k in batches[0]:
        out["batches_" + k] = np.stack([b[k] for b in batches]).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "training_long.npz"), **out)
    print("wrote training_long.npz", os.path.getsize(os.path.join(HERE, "training_long.npz")) / 2 ** 20, "MiB")


if __name__ == "__main__":
    main()
