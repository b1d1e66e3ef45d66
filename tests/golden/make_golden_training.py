"""Golden fixture of one training step of the identity decoder, produced by RUNNING THE REFERENCE
(src/NPHM/models/loss_functions.py actual_compute_loss with the reference's FastEnsembleDeepSDFMirrored,
PyTorch-CPU fp32) in the build container:

    python tests/golden/make_golden_training.py        ->  tests/golden/training.npz

Batch of 3 subjects with the point-set layout of face_dataset.py:93-123 at reduced size (synthetic points: no
dataset here), latents with the reference's sampling statistics, the seeded random-init decoder of make_golden.py
in train mode.  Stored: the batch, the loss dictionary, the total loss with nphm.yaml's lambdas and - after
loss.backward() (training.py:119-124) - the gradients of the latents, of ensemble tensors (the large ones for three weight sets) and of mlp_pos,
plus the norm of every parameter gradient; and the loss dictionary / code gradient of the validation step (eval mode)."""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, HERE)
for missing in ("trimesh", "mcubes"):
    sys.modules.setdefault(missing, types.ModuleType(missing))

import make_golden as G                                          # noqa: E402
from NPHM.models.loss_functions import actual_compute_loss      # noqa: E402

LAMBDAS = {"lat_reg": 0.01, "surf_sdf": 2.0, "normals": 0.3, "space_sdf": 0.01, "grad": 0.1, "anchors": 7.5,
           "symm_dist": 0.01, "middle_dist": 0.0}                # scripts/configs/nphm.yaml:24-32
SETS = [0, 17, 23]
SIZES = {"face": 150, "non_face": 10, "far": 18}                 # face_dataset.py:94-111 with 750 -> 150


def main():
    net, anchors = G.build_identity()
    net.train()
    B = 3
    gen = torch.Generator().manual_seed(2024)
    st = np.load(os.path.join(HERE, "nphm_lat_stats.npz"))
    lat = (torch.randn(B, 1, 1344, generator=gen) * torch.from_numpy(st["std"]) * 0.85 + torch.from_numpy(st["mean"])).float()
    lat.requires_grad_()
    box = torch.tensor([0.5, 0.6, 0.5])

    def pts(n):
        return ((torch.rand(B, n, 3, generator=gen) - 0.5) * box + torch.tensor([0.0, 0.05, 0.05])).float()

    def nrm(n):
        return torch.nn.functional.normalize(torch.randn(B, n, 3, generator=gen), dim=-1)

    face, non_face = pts(SIZES["face"]), pts(SIZES["non_face"])
    u = torch.nn.functional.normalize(torch.randn(B, SIZES["far"], 3, generator=gen), dim=-1)
    far = u * torch.rand(B, SIZES["far"], 1, generator=gen) * 0.5
    near = torch.cat([face, non_face], dim=1) + torch.randn(B, SIZES["face"] + SIZES["non_face"], 3, generator=gen) * 0.01
    batch = {"points_face": face, "normals_face": nrm(SIZES["face"]), "points_non_face": non_face,
             "normals_non_face": nrm(SIZES["non_face"]), "sup_grad_far": far, "sup_grad_near": near,
             "gt_anchors": anchors.reshape(1, 39, 3).repeat(B, 1, 1) + torch.randn(B, 39, 3, generator=gen) * 0.01}
    losses = actual_compute_loss({k: v.clone() for k, v in batch.items()}, net, lat)
    total = sum(LAMBDAS[k] * losses[k] for k in losses)
    total.backward()
    grads = {n: p.grad for n, p in net.named_parameters()}
    out = {"state_hash": np.array(G.state_hash(net)), "lat": lat.detach().numpy(), "total": total.detach().numpy(),
           "grad_lat": lat.grad.numpy(),
           "grad_names": np.array(sorted(grads)), "grad_norms": np.array([float(grads[n].norm()) for n in sorted(grads)])}
    for k, v in batch.items():
        out["batch_" + k] = v.numpy()
    for k, v in losses.items():
        out["loss_" + k] = v.detach().numpy()
    for n in ("ensembled_deep_sdf.lin0.weight", "ensembled_deep_sdf.lin2.weight", "ensembled_deep_sdf.lin3.weight"):
        out["grad_" + n] = grads[n][SETS].numpy().copy()          # a symmetric, a mid-line and the background set
    for n in ("ensembled_deep_sdf.lin4.weight", "ensembled_deep_sdf.lin1.bias", "mlp_pos.4.weight"):
        out["grad_" + n] = grads[n].numpy().copy()
    out["sets"] = np.array(SETS)
    # the validation step (training.py:250-268): the same loss in eval mode - the decoder overwrites the last point of
    # each of its four calls (EnsembledDeepSDF.py:260-261) - differentiated w.r.t. the codes
    net.eval()
    lat_val = lat.detach().clone().requires_grad_()
    losses_val = actual_compute_loss({k: v.clone() for k, v in batch.items()}, net, lat_val)
    sum(LAMBDAS[k] * losses_val[k] for k in losses_val).backward()
    for k, v in losses_val.items():
        out["val_loss_" + k] = v.detach().numpy()
    out["val_grad_lat"] = lat_val.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "training.npz"), **out)
    print({k: float(v) for k, v in losses.items()}, float(total))


if __name__ == "__main__":
    main()
