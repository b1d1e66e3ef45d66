"""Correspondence search of the fitting loop — host-side mirror of
src/NPHM/models/iterative_root_finding.py: for every observed (posed) point find the canonical
point x_c with x_c + F_ex(x_c) = x_obs by Broyden's method (``broyden`` :5-71, ``search`` :91-168),
plus ``nabla`` (:75-87).  The deformation field is evaluated under ``no_grad`` inside the
iteration, i.e. by the fused HIP kernel on a ROCm device."""
from __future__ import annotations

import torch

from .diff_operators import gradient, inverse3x3, jac


def broyden(g, x_init, J_inv_init, max_steps=50, cvg_thresh=1e-5, dvg_thresh=1, eps=1e-6):
    """Batched Broyden root finding (iterative_root_finding.py:5-71).
    g: ([P,3,1], mask [P]) -> residuals of the masked points [p,3,1]; x_init [P,3,1];
    J_inv_init [P,3,3].  Returns {'result' [P,3,1], 'diff' [P], 'valid_ids' [P]}.

    Faithful to the reference including its aliasing: ``x_opt`` IS ``x`` there (:33), so 'result' is
    the final iterate of every point (points stop moving once converged / diverged), while 'diff'
    is the smallest residual norm seen."""
    x = x_init.clone().detach()
    J_inv = J_inv_init.clone().detach()
    active = torch.ones(x.shape[0], dtype=torch.bool)
    gx = g(x, mask=active)
    update = -J_inv.bmm(gx)
    best_norm = torch.linalg.norm(gx.squeeze(-1), dim=-1)
    delta_gx = torch.zeros_like(gx)
    delta_x = torch.zeros_like(x)
    active = torch.ones_like(best_norm).bool()

    for _ in range(max_steps):
        delta_x[active] = update
        x[active] += delta_x[active]
        delta_gx[active] = g(x, mask=active) - gx[active]
        gx[active] += delta_gx[active]

        norm = torch.linalg.norm(gx.squeeze(-1), dim=-1)
        improved = norm < best_norm
        best_norm[improved] = norm.clone().detach()[improved]

        active = (best_norm > cvg_thresh) & (norm < dvg_thresh)
        if active.sum() <= 0:
            break

        # rank-one update of the inverse Jacobian ("good Broyden")
        dx, dg, Ji = delta_x[active], delta_gx[active], J_inv[active]
        vT = dx.transpose(-1, -2).bmm(Ji)
        a = dx - Ji.bmm(dg)
        b = vT.bmm(dg)
        b[b >= 0] += eps
        b[b < 0] -= eps
        J_inv[active] += (a / b).bmm(vT)
        update = -J_inv[active].bmm(gx[active])

    return {"result": x, "diff": best_norm, "valid_ids": best_norm < cvg_thresh}


def nabla(decoder_shape, xc, cond, anchors):
    """SDF value and its spatial gradient at xc (iterative_root_finding.py:75-87)."""
    xc.requires_grad_(True)
    sdf, _ = decoder_shape(xc, cond, anchors)
    return sdf, gradient(sdf, xc)


def search(obs, cond, decoder_expr, anchors, multi_corresp=True):
    """Canonical correspondences of the observed points (iterative_root_finding.py:91-168).
    obs [B,N,3]; cond [B,N,L]; anchors [B,N,K,3] or None.  Returns (xc [B,N,3] — [B,N,5,3] with
    ``multi_corresp`` —, result dict of ``broyden`` with 'valid_ids' reshaped alike)."""
    B, N, _ = obs.shape
    n_init = 5
    if multi_corresp:
        # five starts per point: the observation itself and four jittered copies (sigma 0.05)
        xc_init = obs.detach().clone().unsqueeze(2).repeat(1, 1, n_init, 1)
        jitter = torch.randn(xc_init.shape, device=xc_init.device) * 0.05
        jitter[:, :, 0, :] = 0
        xc_init = (xc_init + jitter).reshape(B, N * n_init, 3)
        obs = obs.repeat_interleave(n_init, dim=1)
        cond = cond[:, 0, :].unsqueeze(1).repeat(1, xc_init.shape[1], 1)
        if anchors is not None:
            a0 = anchors[:, 0, :, :] if anchors.dim() == 4 else anchors
            anchors = a0.unsqueeze(1).repeat(1, xc_init.shape[1], 1, 1)
    else:
        xc_init = obs.detach()                 # an alias: nothing below writes into it (the solvers copy / own their iterates)

    # the reference: jac(...) then `.inverse()` (:118).  The fused value+Jacobian launch also yields x_init + F(x_init),
    # i.e. the residual of the solver's iteration 0: handed to the fused solver, which then skips that evaluation
    posed_init = None
    fused = decoder_expr.jacobian(xc_init, cond, anchors, inverse=True) if hasattr(decoder_expr, "jacobian") else None
    if fused is not None:
        posed_init, J0, J_inv_init = fused               # (the inverse from the same launch)
        J_inv_init = J_inv_init.flatten(0, 1)
    else:
        J0 = jac(decoder_expr, xc_init, cond, anchors).detach()
        J_inv_init = inverse3x3(J0.detach()).flatten(0, 1)
    x0 = xc_init.reshape(-1, 3, 1)
    # conditioning may come as one row per batch entry (cond [B,1,L], anchors [B,K,3]: what the mirrored fitting
    # loop passes); the python solver below wants the reference's per-point tensors
    n_pts = xc_init.shape[1]
    cond_full = cond if cond.shape[1] == n_pts else cond.expand(-1, n_pts, -1)
    anchors_full = anchors if anchors is None or anchors.dim() == 4 else anchors.unsqueeze(1).expand(-1, n_pts, -1, -1)

    def residual(xc_flat, mask=None):
        # the field is evaluated for ALL points, the mask is applied afterwards (reference :131-149)
        if multi_corresp:
            xc = xc_flat.reshape(B, -1, 3)
            xd = decoder_expr(xc, cond_full, anchors_full)[0] + xc
        else:
            xc = xc_flat.reshape(1, xc_flat.shape[0], 3)
            if cond.shape[0] != 1:
                xd = decoder_expr(xc, cond_full.reshape(1, -1, cond.shape[2]),
                                  None if anchors is None else anchors_full.reshape(1, -1, anchors_full.shape[2], 3))[0]
            else:
                xd = decoder_expr(xc, cond_full, anchors_full)[0]
            xd = xd + xc
        err = xd - (obs.reshape(1, -1, 3) if obs.shape[0] != 1 else obs)
        return err.flatten(0, 1)[mask].unsqueeze(-1)

    result = None
    if hasattr(decoder_expr, "broyden"):
        # fused solver: same per-point state machine, one launch, no host syncs (None: not applicable)
        if multi_corresp or cond.shape[0] == 1 or cond.shape[1] == 1:
            # batch rows stay batch rows (one conditioning row each): no flattening, nothing to re-discover
            result = decoder_expr.broyden(obs, xc_init, J_inv_init, cond, anchors, max_steps=15, cvg_thresh=1e-6,
                                          dvg_thresh=0.2, **({} if posed_init is None else {"posed_init": posed_init}))
        else:      # the reference flattens the batch into one row of points (:137-139)
            result = decoder_expr.broyden(obs.reshape(1, -1, 3), xc_init.reshape(1, -1, 3), J_inv_init,
                                          cond_full.reshape(1, -1, cond.shape[2]),
                                          None if anchors is None else anchors_full.reshape(1, -1, anchors_full.shape[2], 3),
                                          max_steps=15, cvg_thresh=1e-6, dvg_thresh=0.2)
    if result is None:
        with torch.no_grad():
            result = broyden(residual, x0, J_inv_init, cvg_thresh=1e-6, dvg_thresh=0.2, max_steps=15)

    if multi_corresp:
        xc_opt = result["result"].reshape(B, N, -1, 3)
        result["valid_ids"] = result["valid_ids"].reshape(B, N, n_init)
    else:
        xc_opt = result["result"].reshape(B, N, 3)
        result["valid_ids"] = result["valid_ids"].reshape(B, N)
    return xc_opt, result
