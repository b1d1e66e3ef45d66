"""Golden fixtures of the deformation-stage losses, produced by RUNNING THE REFERENCE
(src/NPHM/models/loss_functions.py: compute_loss_corresp_forward :282-326 and loss_joint :113-279, with the
reference's FastEnsembleDeepSDFMirrored / DeformationNetwork, PyTorch-CPU fp32) in the build container:

    python tests/golden/make_golden_training_def.py        ->  tests/golden/training_def.npz

Both functions draw uniform samples with torch.rand (and the 'compress' network adds training noise with
torch.randn): the global RNG is seeded right before each call and the mirror consumes it in the same order.
compute_loss_corresp_forward: 'compress' network in train mode, anchors from the identity decoder's mlp_pos.
loss_joint: 'glob_only' network (the reference passes anchors=None to the expression decoder, which 'compress'
cannot serve), epoch 10, batch of 3 with one neutral entry.  Stored: batches, embedding tables, loss dictionaries,
gradients of the expression codes and the norm of every parameter gradient after backward() of the plain sum."""
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

import make_golden as G                                                              # noqa: E402
from NPHM.models.deepSDF import DeformationNetwork                                   # noqa: E402
from NPHM.models.loss_functions import compute_loss_corresp_forward, loss_joint     # noqa: E402

SEED_CALL = 4242


def build_glob_only(anchors):
    torch.manual_seed(1)
    return DeformationNetwork(mode="glob_only", lat_dim_expr=200, lat_dim_id=32, lat_dim_glob_shape=64,
                              lat_dim_loc_shape=32, n_loc=39, anchors=anchors, hidden_dim=400, nlayers=4,
                              input_dim=3, out_dim=3)


def tables(gen, n_subj, n_expr):
    st = np.load(os.path.join(HERE, "nphm_lat_stats.npz"))
    shape = torch.nn.Embedding(n_subj, 1344)
    expr = torch.nn.Embedding(n_expr, 200)
    with torch.no_grad():
        shape.weight.copy_((torch.randn(n_subj, 1344, generator=gen) * torch.from_numpy(st["std"]) * 0.85 + torch.from_numpy(st["mean"])).float())
        expr.weight.copy_(torch.randn(n_expr, 200, generator=gen) * 0.1)
    return shape, expr


def grads_of(mods):
    names, norms = [], []
    for tag, m in mods:
        for n, p in m.named_parameters():
            names.append(tag + "." + n)
            norms.append(0.0 if p.grad is None else float(p.grad.norm()))
    return np.array(names), np.array(norms)


def main():
    shape_net, anchors = G.build_identity()
    gen = torch.Generator().manual_seed(99)
    B = 3
    box = torch.tensor([0.5, 0.6, 0.5])
    pts = lambda n: ((torch.rand(B, n, 3, generator=gen) - 0.5) * box + torch.tensor([0.0, 0.05, 0.05])).float()
    nrm = lambda n: torch.nn.functional.normalize(torch.randn(B, n, 3, generator=gen), dim=-1)
    lat_shape, lat_expr = tables(gen, 4, 6)
    subj = torch.tensor([[2], [0], [3]])
    idx = torch.tensor([[5], [1], [2]])
    out = {"shape_table": lat_shape.weight.detach().numpy(), "expr_table": lat_expr.weight.detach().numpy(),
           "subj_ind": subj.numpy(), "idx": idx.numpy(), "seed_call": np.array(SEED_CALL)}

    # ---- compute_loss_corresp_forward: 'compress' network, train mode --------------------------------------
    expr_net = G.build_deformation(anchors).train()
    shape_net.train()
    neutral = pts(120)
    batch = {"points_neutral": neutral, "points_posed": torch.cat([neutral + 0.01 * torch.randn(B, 120, 3, generator=gen), nrm(120)], -1),
             "gt_anchors": anchors.reshape(1, 39, 3).repeat(B, 1, 1), "subj_ind": subj, "idx": idx}
    torch.manual_seed(SEED_CALL)
    losses = compute_loss_corresp_forward(dict(batch), expr_net, shape_net, lat_expr, lat_shape, "cpu", epoch=3)
    sum(losses.values()).backward()
    for k, v in batch.items():
        out["c_batch_" + k] = v.numpy()
    for k, v in losses.items():
        out["c_loss_" + k] = v.detach().numpy()
    out["c_grad_expr_table"] = lat_expr.weight.grad.numpy().copy()
    out["c_grad_names"], out["c_grad_norms"] = grads_of([("expr", expr_net), ("shape", shape_net)])
    out["c_state_hash_expr"] = np.array(G.state_hash(expr_net))

    # ---- loss_joint: 'glob_only' network --------------------------------------------------------------------
    for m in (expr_net, shape_net, lat_expr, lat_shape):
        m.zero_grad(set_to_none=True)
    joint_net = build_glob_only(anchors).train()
    n = 60
    surf = pts(n)
    batch = {"points_surface": surf, "normals_surface": nrm(n), "points_surface_outer": pts(20), "normals_surface_outer": nrm(20),
             "points_off_surface": pts(30), "normals_off_surface": nrm(30), "sdfs_off_surface": 0.01 * torch.randn(B, 30, 1, generator=gen),
             "sup_grad_far": nrm(12) * torch.rand(B, 12, 1, generator=gen) * 0.5, "corresp_posed": pts(40),
             "corresp_neutral": pts(40), "is_neutral": torch.tensor([[0.0], [1.0], [0.0]]),
             "gt_anchors": anchors.reshape(1, 39, 3).repeat(B, 1, 1), "subj_ind": subj, "idx": idx}
    torch.manual_seed(SEED_CALL)
    losses = loss_joint(dict(batch), shape_net, joint_net, lat_shape, lat_expr, "cpu", epoch=10)
    sum(losses.values()).backward()
    for k, v in batch.items():
        out["j_batch_" + k] = v.numpy()
    for k, v in losses.items():
        out["j_loss_" + k] = v.detach().numpy()
    out["j_grad_expr_table"] = lat_expr.weight.grad.numpy().copy()
    out["j_grad_shape_table"] = lat_shape.weight.grad.numpy().copy()
    out["j_grad_names"], out["j_grad_norms"] = grads_of([("expr", joint_net), ("shape", shape_net)])
    out["j_state_hash_expr"] = np.array(G.state_hash(joint_net))
    np.savez_compressed(os.path.join(HERE, "training_def.npz"), **out)
    print({k: float(v) for k, v in losses.items()})


if __name__ == "__main__":
    main()
