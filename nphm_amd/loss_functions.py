"""Training loss of the identity decoder - host-side mirror of ``compute_loss`` / ``actual_compute_loss``
(src/NPHM/models/loss_functions.py:7-110): same arguments, same keys and values of the returned dictionary
(the trainer weights them with cfg['lambdas'], training.py:119-124).

The reference evaluates the decoder four times per step (on-surface face / non-face points, near-surface and
far points) and differentiates each prediction w.r.t. its points.  Per point the field is independent of its
neighbours, so the four sets are evaluated here as ONE batch of B x (Nf + Nn + Nr + Nfar) points and the
predictions / gradients are split afterwards: one forward and one backward launch of the training kernels
(ident_train_kernel.hip) instead of four, with identical values."""
from __future__ import annotations

import torch

from .diff_operators import gradient

_POINT_SETS = ("points_face", "points_non_face", "sup_grad_near", "sup_grad_far")


def compute_loss(batch, decoder, latent_codes, device):
    """loss_functions.py:7-18: move the batch to ``device``, look the latent codes up, evaluate the loss terms."""
    batch = {k: v for k, v in batch.items() if k != "path"}
    idx = batch["idx"].to(device)
    tensors = {k: v.to(device).float() for k, v in batch.items()}
    return actual_compute_loss(tensors, decoder, latent_codes(idx))


def actual_compute_loss(batch_cuda, decoder, glob_cond):
    """loss_functions.py:20-110.  batch_cuda: points_face / normals_face [B,Nf,3], points_non_face /
    normals_non_face [B,Nn,3], sup_grad_near [B,Nr,3], sup_grad_far [B,Nfar,3], gt_anchors [B,K,3];
    glob_cond [B,1,lat_dim].  Returns surf_sdf, normals, space_sdf, grad, lat_reg and - for a decoder with
    anchors - anchors, symm_dist, middle_dist."""
    has_anchors = hasattr(decoder, "anchors")
    sizes = [batch_cuda[k].shape[1] for k in _POINT_SETS]
    x = torch.cat([batch_cuda[k] for k in _POINT_SETS], dim=1).clone().detach().requires_grad_()
    # one code per subject: the decoders broadcast a [B,1,L] code over the points themselves (EnsembledDeepSDF.py:223,
    # deepSDF.py); the reference's glob_cond.repeat(1, N, 1) would only be compared back to one row by the HIP tiers
    pred, anchors = decoder(x, glob_cond, batch_cuda["gt_anchors"] if has_anchors else None)
    grad = gradient(pred, x)
    # the point sets are consecutive slices of the batch: [face | non-face | near | far].  The reference's means over
    # concatenated per-set terms are means over slices of ONE tensor (same values, a fraction of the autograd nodes):
    n_face, n_non, n_near, n_far = sizes
    n_surf = n_face + n_non
    sdf = pred.squeeze(-1)
    normals = torch.cat([batch_cuda["normals_face"], batch_cuda["normals_non_face"]], dim=1)
    normal_err = (grad[:, :n_surf] - normals).norm(2, dim=-1)
    # non-face points: error clamped at 0.75 and halved (loss_functions.py:56-57)
    cap = torch.cat([normal_err.new_full((n_face,), float("inf")), normal_err.new_full((n_non,), 0.75)])
    scale = torch.cat([normal_err.new_ones(n_face), normal_err.new_full((n_non,), 0.5)])
    out = {"surf_sdf": sdf[:, :n_surf].abs().mean(),
           "normals": (torch.minimum(normal_err, cap) * scale).mean(),
           "space_sdf": torch.exp(-1e1 * sdf[:, n_surf + n_near:].abs()).mean(),
           "grad": (grad.norm(dim=-1) - 1).abs().mean(),          # all four sets (loss_functions.py:60-66)
           "lat_reg": (torch.norm(glob_cond, dim=-1) ** 2).mean()}
    if anchors is None:
        return out

    out["anchors"] = (anchors - batch_cuda["gt_anchors"]).square().mean()
    if hasattr(decoder, "lat_dim_glob"):
        # local codes of mirror-symmetric anchors should agree; the mid-line codes pairwise (an odd one out is skipped)
        z = glob_cond.squeeze(1)
        g, loc, n_symm = decoder.lat_dim_glob, decoder.lat_dim_loc, decoder.num_symm_pairs
        pairs = z[:, g:g + 2 * n_symm * loc].view(z.shape[0], 2 * n_symm, loc)
        middle = z[:, g + 2 * n_symm * loc:-loc].view(z.shape[0], decoder.num_kps - 2 * n_symm, loc)
        n_mid = middle.shape[1] - middle.shape[1] % 2
        out["symm_dist"] = torch.norm(pairs[:, ::2] - pairs[:, 1::2], dim=-1).mean()
        out["middle_dist"] = torch.norm(middle[:, :n_mid:2] - middle[:, 1:n_mid:2], dim=-1).mean()
    else:
        out["symm_dist"] = None
        out["middle_dist"] = None
    return out
