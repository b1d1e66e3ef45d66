"""Training loss of the identity decoder - host-side mirror of ``compute_loss`` / ``actual_compute_loss``
(src/NPHM/models/loss_functions.py:7-110): same arguments, same keys and values of the returned dictionary
(the trainer weights them with cfg['lambdas'], training.py:119-124), and of the deformation-stage losses
``compute_loss_corresp_forward`` (:282-326) and ``loss_joint`` (:113-279).

The reference evaluates the decoder four times per step (on-surface face / non-face points, near-surface and
far points) and differentiates each prediction w.r.t. its points.  Per point the field is independent of its
neighbours, so the four sets are evaluated here as ONE batch of B x (Nf + Nn + Nr + Nfar) points and the
predictions / gradients are split afterwards: one forward and one backward launch of the training kernels
(ident_train_kernel.hip) instead of four, with identical values."""
from __future__ import annotations

import torch

from .diff_operators import gradient

_POINT_SETS = ("points_face", "points_non_face", "sup_grad_near", "sup_grad_far")
_TERMS = ("surf_sdf", "normals", "space_sdf", "grad", "lat_reg", "anchors", "symm_dist", "middle_dist")


class _TrainLossFn(torch.autograd.Function):
    """The eight loss terms below from ONE launch and their gradients w.r.t. the SDF values, the spatial gradients, the codes
    and the predicted anchors from a second one (``nphm_train_loss[_backward]``, csrc/train_loss_kernels.hip) - the PyTorch
    formulation of the same terms is ~40 launches forward and ~50 in the backward pass, 4 % of a training step that three
    large kernels bound.  Outputs: eight 0-dim tensors in ``_TERMS`` order (the last three are zeros without anchors / local
    codes and are dropped by the caller)."""

    @staticmethod
    def forward(ctx, sdf, grad, normals, z, anchors, anchors_gt, sizes, layout):
        import ctypes
        from . import _lib
        lib = _lib.load()
        dev = sdf.device
        B = sdf.shape[0]
        sdf_c, grad_c = sdf.detach().reshape(B, -1).contiguous(), grad.detach().contiguous()
        nrm_c, z_c = normals.detach().contiguous(), z.detach().reshape(B, -1).contiguous()
        a_c = None if anchors is None else anchors.detach().contiguous()
        gt_c = None if anchors is None else anchors_gt.detach().contiguous().float()
        stream = torch.cuda.current_stream(dev).cuda_stream
        key = (str(dev), stream)                # the partial sums and the arrival counter belong to one stream's launches
        ws = _TrainLossFn._scratch.get(key)
        if ws is None:
            ws = (torch.zeros(lib.nphm_train_loss_blocks() * 8, dtype=torch.float32, device=dev),
                  torch.zeros(1, dtype=torch.int32, device=dev))
            _TrainLossFn._scratch[key] = ws
        row = torch.empty(8, dtype=torch.float32, device=dev)
        csizes, clayout = (ctypes.c_int * 4)(*sizes), (ctypes.c_int * 4)(*layout)
        ptr = lambda t: None if t is None else t.data_ptr()
        K = 0 if a_c is None else a_c.shape[1]
        _lib.check(lib.nphm_train_loss(sdf_c.data_ptr(), grad_c.data_ptr(), nrm_c.data_ptr(), z_c.data_ptr(), ptr(a_c), ptr(gt_c), B,
                                       csizes, z_c.shape[1], K, clayout, ws[0].data_ptr(), ws[1].data_ptr(), row.data_ptr(), stream),
                   "nphm_train_loss")
        ctx.save_for_backward(sdf_c, grad_c, nrm_c, z_c, a_c, gt_c)
        ctx.meta = (tuple(sizes), tuple(layout), sdf.shape, z.shape)
        return tuple(row.unbind(0))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *g_terms):
        import ctypes
        from . import _lib
        lib = _lib.load()
        sdf_c, grad_c, nrm_c, z_c, a_c, gt_c = ctx.saved_tensors
        sizes, layout, sdf_shape, z_shape = ctx.meta
        dev = sdf_c.device
        zero = None
        parts = []
        for g in g_terms:
            if g is None:
                zero = torch.zeros((), dtype=torch.float32, device=dev) if zero is None else zero
                g = zero
            parts.append(g.reshape(()).float())
        c = torch.stack(parts)
        g_sdf, g_grad, g_z = torch.empty_like(sdf_c), torch.empty_like(grad_c), torch.empty_like(z_c)
        g_a = None if a_c is None else torch.empty_like(a_c)
        ptr = lambda t: None if t is None else t.data_ptr()
        B = sdf_c.shape[0]
        _lib.check(lib.nphm_train_loss_backward(sdf_c.data_ptr(), grad_c.data_ptr(), nrm_c.data_ptr(), z_c.data_ptr(), ptr(a_c), ptr(gt_c),
                                                B, (ctypes.c_int * 4)(*sizes), z_c.shape[1], 0 if a_c is None else a_c.shape[1],
                                                (ctypes.c_int * 4)(*layout), c.data_ptr(), g_sdf.data_ptr(), g_grad.data_ptr(),
                                                g_z.data_ptr(), ptr(g_a), torch.cuda.current_stream(dev).cuda_stream),
                   "nphm_train_loss_backward")
        return g_sdf.view(sdf_shape), g_grad, None, g_z.view(z_shape), g_a, None, None, None


_TrainLossFn._scratch = {}


def _fused_terms(decoder, pred, grad, normals, glob_cond, anchors, anchors_gt, sizes):
    """the loss dictionary through ``_TrainLossFn`` or None (CPU tensors, other dtypes, NPHM_AMD_TRAIN_LOSS_FUSED=0)"""
    import os
    if not (pred.is_cuda and pred.dtype == torch.float32 and grad.dtype == torch.float32 and glob_cond.dim() == 3
            and glob_cond.shape[1] == 1 and os.environ.get("NPHM_AMD_TRAIN_LOSS_FUSED", "1") not in ("0", "")):
        return None
    layout = (glob_cond.shape[-1], 0, 0, 0)
    if hasattr(decoder, "lat_dim_glob"):
        n_mid = decoder.num_kps - 2 * decoder.num_symm_pairs
        layout = (decoder.lat_dim_glob, decoder.lat_dim_loc, decoder.num_symm_pairs, n_mid // 2)
    terms = _TrainLossFn.apply(pred.squeeze(-1), grad, normals, glob_cond, anchors, anchors_gt, sizes, layout)
    n = 5 if anchors is None else (6 if not hasattr(decoder, "lat_dim_glob") else 8)
    out = dict(zip(_TERMS[:n], terms[:n]))
    if n == 6:
        out["symm_dist"], out["middle_dist"] = None, None          # as _lat_regularisers answers without local codes
    return out


_LAMBDA_VECTORS = {}


def weighted_total(loss_dict, lambdas):
    """``sum(lambdas[k] * loss_dict[k] for k in loss_dict)`` - the line of the trainers that turns the loss terms into the
    scalar they call ``backward()`` on (training.py:118-122: ``loss += loss_dict[k] * lambdas[k]`` over the terms) - as a stack
    and a dot product against a cached device vector of the weights: 3 launches forward and 3 backward where the Python sum over
    eight 0-dim tensors is 16 + 16 (a training step is ~150 launches around three large kernels).  Terms that are None or carry
    no weight entry are skipped, like a trainer that only weights what it lists."""
    keys = [k for k, v in loss_dict.items() if v is not None and k in lambdas]
    terms = [loss_dict[k] for k in keys]
    if not terms:
        raise ValueError("weighted_total: no weighted loss term")
    dev = terms[0].device
    key = (str(dev), tuple(float(lambdas[k]) for k in keys))
    lam = _LAMBDA_VECTORS.get(key)
    if lam is None:
        if len(_LAMBDA_VECTORS) > 64:
            _LAMBDA_VECTORS.clear()
        lam = _LAMBDA_VECTORS[key] = torch.tensor(key[1], dtype=torch.float32, device=dev)
    return torch.dot(torch.stack([t.reshape(()).float() for t in terms]), lam)


def compute_loss(batch, decoder, latent_codes, device):
    """loss_functions.py:7-18: move the batch to ``device``, look the latent codes up, evaluate the loss terms."""
    return actual_compute_loss(_to_device(batch, device), decoder, latent_codes(batch["idx"].to(device)))


def actual_compute_loss(batch_cuda, decoder, glob_cond):
    """loss_functions.py:20-110.  batch_cuda: points_face / normals_face [B,Nf,3], points_non_face /
    normals_non_face [B,Nn,3], sup_grad_near [B,Nr,3], sup_grad_far [B,Nfar,3], gt_anchors [B,K,3];
    glob_cond [B,1,lat_dim].  Returns surf_sdf, normals, space_sdf, grad, lat_reg and - for a decoder with
    anchors - anchors, symm_dist, middle_dist."""
    has_anchors = hasattr(decoder, "anchors")
    sizes = [batch_cuda[k].shape[1] for k in _POINT_SETS]
    anchors_gt = batch_cuda["gt_anchors"] if has_anchors else None
    if decoder.training or not hasattr(decoder, "num_kps"):
        x = torch.cat([batch_cuda[k] for k in _POINT_SETS], dim=1).clone().detach().requires_grad_()
        # one code per subject: the decoders broadcast a [B,1,L] code over the points themselves (EnsembledDeepSDF.py:223,
        # deepSDF.py); the reference's glob_cond.repeat(1, N, 1) would only be compared back to one row by the HIP tiers
        fused = decoder.value_and_gradient(x, glob_cond) if hasattr(decoder, "value_and_gradient") else None
        if fused is not None:              # HIP training tier: value and spatial gradient from one evaluation
            pred, grad, anchors = fused
        else:
            pred, anchors = decoder(x, glob_cond, anchors_gt)
            grad = gradient(pred, x)
    else:
        # eval mode (validation, training.py:250-268): the NPHM decoder overwrites the member values of the LAST point of
        # every call (EnsembledDeepSDF.py:260-261) - four calls, four overwritten points, as in the reference.  The HIP
        # training tier takes the four sets as one batch with those four points named; otherwise four calls
        fused = None
        if hasattr(decoder, "value_and_gradient") and min(sizes) > 0:
            x = torch.cat([batch_cuda[k] for k in _POINT_SETS], dim=1).clone().detach().requires_grad_()
            ends = [sum(sizes[:i + 1]) - 1 for i in range(len(sizes))]
            fused = decoder.value_and_gradient(x, glob_cond, last_points=ends)
        if fused is not None:
            pred, grad, anchors = fused
        else:
            preds, grads = [], []
            for k in _POINT_SETS:
                x = batch_cuda[k].clone().detach().requires_grad_()
                p_k, anchors = decoder(x, glob_cond, anchors_gt)
                preds.append(p_k)
                grads.append(gradient(p_k, x))
            pred, grad = torch.cat(preds, dim=1), torch.cat(grads, dim=1)
    # the point sets are consecutive slices of the batch: [face | non-face | near | far].  The reference's means over
    # concatenated per-set terms are means over slices of ONE tensor (same values, a fraction of the autograd nodes):
    n_face, n_non, n_near, n_far = sizes
    n_surf = n_face + n_non
    sdf = pred.squeeze(-1)
    normals = torch.cat([batch_cuda["normals_face"], batch_cuda["normals_non_face"]], dim=1)
    fused_out = _fused_terms(decoder, pred, grad, normals, glob_cond, anchors,
                             None if anchors is None else batch_cuda["gt_anchors"], sizes)
    if fused_out is not None:
        return fused_out
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
    out["symm_dist"], out["middle_dist"] = _lat_regularisers(decoder, glob_cond)
    return out


# ------------------------------------------------------------------------------------------------------------------
# deformation network (second training stage)
# ------------------------------------------------------------------------------------------------------------------
def _to_device(batch, device):
    return {k: v.to(device).float() for k, v in batch.items() if k != "path"}


def _lat_regularisers(decoder_shape, cond_shape):
    """symm_dist / middle_dist of the identity codes (loss_functions.py:73-88, :219-234): None, None for a decoder
    without local codes."""
    if not hasattr(decoder_shape, "lat_dim_glob"):
        return None, None
    z = cond_shape.squeeze(1)
    g, loc, n_symm = decoder_shape.lat_dim_glob, decoder_shape.lat_dim_loc, decoder_shape.num_symm_pairs
    pairs = z[:, g:g + 2 * n_symm * loc].view(z.shape[0], 2 * n_symm, loc)
    middle = z[:, g + 2 * n_symm * loc:-loc].view(z.shape[0], decoder_shape.num_kps - 2 * n_symm, loc)
    n_mid = middle.shape[1] - middle.shape[1] % 2
    return (torch.norm(pairs[:, ::2] - pairs[:, 1::2], dim=-1).mean(),
            torch.norm(middle[:, :n_mid:2] - middle[:, 1:n_mid:2], dim=-1).mean())


def compute_loss_corresp_forward(batch, decoder, decoder_shape, latent_codes, latent_codes_shape, device, epoch=-1,
                                 exp_path=None):
    """Loss of the forward-deformation network (loss_functions.py:282-326; trainer: training_corresp.py:159):
    neutral points displaced by the field should land on their posed correspondences; the field should vanish on
    uniform samples of the [-1.25, 1.25]^3 box; expression codes are regularised.  First order only."""
    batch_cuda = _to_device(batch, device)
    cond_shape = latent_codes_shape(batch["subj_ind"].to(device))
    cond_pose = latent_codes(batch["idx"].to(device))
    if decoder_shape is not None and decoder_shape.mlp_pos is not None:
        # anchors of the identity code (EnsembledDeepSDF.py:228-229), as the deformation network is conditioned on them
        anchors = decoder_shape.mlp_pos(cond_shape[..., :decoder_shape.lat_dim_glob]).view(cond_pose.shape[0], -1, 3)
        anchors = anchors + decoder.anchors.squeeze(0)
    else:
        anchors = batch_cuda["gt_anchors"]
    cond = torch.cat([cond_shape, cond_pose], dim=-1)

    # (the reference marks the neutral points as requiring a gradient, :301, and never reads it: without the mark the
    # backbone's first-order training tier serves the call - same losses, same parameter / code gradients)
    neutral = batch_cuda["points_neutral"].clone().detach()
    cond_rep = cond.repeat(1, neutral.shape[1], 1)
    delta, _ = decoder(neutral, cond_rep, anchors)
    posed = neutral + delta.squeeze()
    samples = (torch.rand(cond_rep.shape[0], 100, 3, device=cond_rep.device, dtype=cond_rep.dtype) - 0.5) * 2.5
    delta_free, _ = decoder(samples, cond_rep[:, :100, :], anchors)
    return {"corresp": ((posed - batch_cuda["points_posed"][:, :, :3]) ** 2).mean(),
            "lat_reg": (torch.norm(cond_pose, dim=-1) ** 2).mean(),
            "loss_reg_zero": (delta_free ** 2).mean()}


def loss_joint(batch, decoder_shape, decoder_expr, latent_codes_shape, latent_codes_expr, device, epoch):
    """Joint loss of identity and deformation network (loss_functions.py:113-279): SDF / normal / eikonal terms of
    the identity field at posed points pulled back through the deformation, canonical far points, latent
    regularisers, anchor, correspondence and deformation regularisers.  The identity decoder is evaluated on its
    training tier (twice differentiable in the canonical points, so the gradients w.r.t. the POSED points chain
    through the deformation network's autograd graph)."""
    batch_cuda = _to_device(batch, device)
    cond_shape = latent_codes_shape(batch["subj_ind"].to(device))
    cond_expr = latent_codes_expr(batch["idx"].to(device))
    cond_cat = torch.cat([cond_shape, cond_expr], dim=-1)
    neutral = batch_cuda["is_neutral"].squeeze(dim=-1) == 1
    any_neutral = bool(neutral.sum() > 0)

    def pulled_back(points, rows=None):
        """(sdf, d sdf / d posed points, offsets) of posed points; ``rows``: boolean selection of batch entries."""
        x = (points if rows is None else points[rows]).clone().detach().requires_grad_()
        c_cat = cond_cat.repeat(1, x.shape[1], 1) if rows is None else cond_cat.repeat(1, x.shape[1], 1)[rows]
        c_shape = cond_shape.repeat(1, x.shape[1], 1) if rows is None else cond_shape.repeat(1, x.shape[1], 1)[rows]
        offsets, _ = decoder_expr(x, c_cat, None)
        sdf, anchors = decoder_shape(x + offsets, c_shape, None)
        return sdf, gradient(sdf, x), offsets, anchors

    def clamped_normal(g, n):
        return torch.clamp((g - n).norm(2, dim=-1), None, 0.75 * 100) / 2

    sdf_s, g_s, off_s, _ = pulled_back(batch_cuda["points_surface"])
    sdf_terms = [sdf_s.abs().squeeze(dim=-1).reshape(-1)]
    normal_terms = [(g_s - batch_cuda["normals_surface"]).norm(2, dim=-1).reshape(-1)]
    eikonal_terms = [(g_s.norm(dim=-1) - 1).abs().reshape(-1)]
    if any_neutral:
        sdf_o, g_o, off_o, _ = pulled_back(batch_cuda["points_surface_outer"], neutral)
        sdf_f, g_f, off_f, _ = pulled_back(batch_cuda["points_off_surface"], neutral)
        sdf_terms += [sdf_o.abs().squeeze(dim=-1).reshape(-1),
                      (sdf_f - batch_cuda["sdfs_off_surface"][neutral]).abs().squeeze(dim=-1).reshape(-1)]
        normal_terms += [clamped_normal(g_o, batch_cuda["normals_surface_outer"][neutral]).reshape(-1),
                         clamped_normal(g_f, batch_cuda["normals_off_surface"][neutral]).reshape(-1)]
        eikonal_terms += [(g_o.norm(dim=-1) - 1).abs().reshape(-1), (g_f.norm(dim=-1) - 1).abs().reshape(-1)]

    # canonical space only: far points
    far = batch_cuda["sup_grad_far"].clone().detach().requires_grad_()
    sdf_far, anchors_pred = decoder_shape(far, cond_shape.repeat(1, far.shape[1], 1), None)
    g_far = gradient(sdf_far, far)
    eikonal_terms = [(g_far.norm(dim=-1) - 1).abs().reshape(-1)] + eikonal_terms

    symm_dist, middle_dist = _lat_regularisers(decoder_shape, cond_shape)

    corresp_posed = batch_cuda["corresp_posed"].clone().detach().requires_grad_()
    if epoch < 3000:
        delta, _ = decoder_expr(corresp_posed, cond_cat.repeat(1, corresp_posed.shape[1], 1), None)
        loss_corresp = (corresp_posed + delta - batch_cuda["corresp_neutral"]).square().mean()
        if epoch > 750:
            loss_corresp = loss_corresp * 0.25
    else:
        loss_corresp = torch.zeros((), device=cond_cat.device, dtype=cond_cat.dtype)

    n_free = min(100, batch_cuda["corresp_posed"].shape[1])
    samples = (torch.rand(cond_shape.shape[0], n_free, 3, device=cond_shape.device, dtype=cond_shape.dtype) - 0.5) * 2.5
    delta_free, _ = decoder_expr(samples, cond_cat.repeat(1, n_free, 1), None)
    loss_reg_zero = delta_free.square().mean()

    if any_neutral:
        loss_neutral = off_s[neutral].square().mean() + off_o.square().mean() + off_f.square().mean()
    else:
        loss_neutral = torch.zeros_like(loss_reg_zero)

    return {"surf_sdf_loss": torch.cat(sdf_terms).mean(),
            "normal_loss": torch.cat(normal_terms).mean(),
            "space_sdf_loss": torch.exp(-1e1 * sdf_far.abs()).mean(),
            "eik_loss": torch.cat(eikonal_terms).mean(),
            "reg_shape": (torch.norm(cond_shape, dim=-1) ** 2).mean(),
            "reg_expr": (torch.norm(cond_expr, dim=-1) ** 2).mean(),
            "anchors": (anchors_pred - batch_cuda["gt_anchors"]).square().mean().mean(),
            "symm_dist": symm_dist.mean(),
            "middle_dist": middle_dist.mean(),
            "corresp": loss_corresp,
            "loss_reg_zero": loss_reg_zero,
            "loss_neutral_zero": loss_neutral}
