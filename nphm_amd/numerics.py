"""Numerics guard of the NPHM identity field's fast modes (pruning + adaptive precision).

The shipped default drops, per point, the smallest blend weights while they sum to <= 40 * prune_tol and runs
members below ``NPHM_LIGHT_TOL`` (1e-3) normalised weight single-pass; both are validated on seeded
random-init weights, where every member's |f_k| is small.  A trained checkpoint can
carry large far-field member values (the reference's blend, EnsembledDeepSDF.py:129-150, multiplies them by
weights that are tiny but not zero), so ``validate_numerics`` measures, for the weights and latents at hand,

* ``max_abs_diff``: max |default mode - dense exact-fp32 kernel| over a sample of lattice and near-anchor points
  (the in-tree ``precision = "f32"``, ``prune_tol < 0`` kernel evaluates all 40 members with fp32 MFMA),
* ``max_pruned``: the largest sum of w_k |f_k| over the members the pruning rule dropped at a point
  (its exact error contribution), ``max_light``: the largest w_k |f_k| of a single-pass member times the
  bf16 rounding level 2^-8 and ``max_two_pass``: ... of a two-pass member times 2^-9 (the size of the error each
  tier can carry),

and warns (or raises with ``strict=True``) above ``tol`` (default 1e-5, a tenth of the 1e-4 bar).
``NPHM_AMD_VALIDATE=1`` runs it once after every ``load_state_dict`` (at the first HIP evaluation, with that
call's latent)."""
from __future__ import annotations

import warnings
from typing import Optional

import numpy as np
import torch

from . import _lib

LIGHT_TOL = 1e-3
MID_TOL = 1e-2


def _sample_points(anchors: torch.Tensor, n: int, seed: int) -> torch.Tensor:
    """n points: half on a coarse lattice of the reference's extraction box, half Gaussian clouds around the
    anchors (sigma 0.03 and 0.1: inside and at the edge of the blend kernels)."""
    dev = anchors.device
    g = torch.Generator().manual_seed(seed)
    lo, hi = torch.tensor([-.55, -.5, -.95]), torch.tensor([0.55, 0.75, 0.4])
    m = max(2, int(round((n // 2) ** (1 / 3))))
    ax = [torch.linspace(float(lo[d]), float(hi[d]), m) for d in range(3)]
    lattice = torch.stack(torch.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3)
    k = n - lattice.shape[0]
    a = anchors.reshape(-1, 3).cpu()
    pick = a[torch.randint(0, a.shape[0], (k,), generator=g)]
    sigma = torch.where(torch.rand(k, 1, generator=g) < 0.5, torch.tensor(0.03), torch.tensor(0.1))
    near = pick + sigma * torch.randn(k, 3, generator=g)
    return torch.cat([lattice, near], 0).to(dev).contiguous()


def validate_numerics(decoder, latents: Optional[torch.Tensor] = None, n: int = 1 << 16, *, tol: float = 1e-5,
                      strict: bool = False, seed: int = 0, points: Optional[torch.Tensor] = None) -> dict:
    """Compare the decoder's current fast mode with the dense exact-fp32 kernel (see module docstring).
    ``latents`` [R, lat_dim] (default: one zero code = the mean anchors); ``points`` [n, 3] overrides the
    sample.  Returns a dict of the measured quantities; warns (``strict``: raises NphmAmdError) when
    ``max_abs_diff`` or ``max_pruned`` exceed ``tol``."""
    from .ensembled_deepsdf import _member_point_lists
    lib = _lib.load()
    dev = next(decoder.parameters()).device
    if dev.type != "cuda" or not decoder.hip_supported():
        raise _lib.NphmAmdError("validate_numerics needs the HIP-backed NPHM identity field on a ROCm device")
    if latents is None:
        latents = torch.zeros(1, decoder.lat_dim, device=dev)
    latents = latents.reshape(-1, decoder.lat_dim).to(device=dev, dtype=torch.float32)
    A = decoder.num_kps + 1
    saved = (decoder.precision, decoder.prune_tol, getattr(decoder, "_needs_validation", False))
    decoder._needs_validation = False
    worst = {"max_abs_diff": 0.0, "max_pruned": 0.0, "max_light": 0.0, "max_two_pass": 0.0, "max_abs_sdf": 0.0,
             "max_abs_member": 0.0}
    try:
        with torch.no_grad():
            for r in range(latents.shape[0]):
                lat = latents[r:r + 1]
                decoder.precision, decoder.prune_tol = saved[0], saved[1]
                packed, state, anchors = decoder.prepare_latent(lat)
                pts = (points.to(dev).float() if points is not None else _sample_points(anchors[0], n, seed + r))[None]
                N = pts.shape[1]
                stream = torch.cuda.current_stream(dev).cuda_stream

                def run(prec_code, prune):
                    out = torch.empty(1, N, 1, dtype=torch.float32, device=dev)
                    _lib.check(lib.nphm_identity_eval_points(packed.data_ptr(), state.data_ptr(), pts.data_ptr(), 1, N, 0,
                                                             float(prune), prec_code, out.data_ptr(), None, stream),
                               "nphm_identity_eval_points")
                    return out
                fast = run(decoder._precision_code(), decoder.prune_tol)
                dense = run(_lib.NPHM_PREC_F32, -1.0)
                # every member's value at every point (member-centric kernel, all pairs listed)
                what_all, tiles, plist = _member_point_lists(anchors, pts, -1.0, A)
                fmem = torch.zeros(1, N, A, dtype=torch.float32, device=dev)
                _lib.check(lib.nphm_identity_member_forward(packed.data_ptr(), decoder._packed_bwd(dev).data_ptr(),
                                                            state.data_ptr(), pts.data_ptr(), N, tiles.data_ptr(),
                                                            tiles.shape[0], None, plist.data_ptr(), fmem.data_ptr(), stream),
                           "nphm_identity_member_forward")
                contrib = what_all * fmem.abs()                                   # w_k |f_k|, [1,N,A]
                if decoder.prune_tol >= 0:
                    kept = _member_point_lists(anchors, pts, decoder.prune_tol, A)[0] > 0
                    pruned = (contrib * (~kept)).sum(dim=2)
                else:
                    pruned = torch.zeros_like(contrib[..., 0])
                light = contrib * (what_all < LIGHT_TOL) * 2.0 ** -8 if decoder.precision.startswith("bf16x3a") else contrib * 0
                # two-pass members (bf16x3a2): weights rounded to bf16, a 2^-9 relative perturbation of the member
                mid = (contrib * ((what_all >= LIGHT_TOL) & (what_all < MID_TOL)) * 2.0 ** -9
                       if decoder.precision == "bf16x3a2" else contrib * 0)
                worst["max_abs_diff"] = max(worst["max_abs_diff"], float((fast - dense).abs().max()))
                worst["max_pruned"] = max(worst["max_pruned"], float(pruned.max()))
                worst["max_light"] = max(worst["max_light"], float(light.max()))
                worst["max_two_pass"] = max(worst["max_two_pass"], float(mid.max()))
                worst["max_abs_sdf"] = max(worst["max_abs_sdf"], float(dense.abs().max()))
                worst["max_abs_member"] = max(worst["max_abs_member"], float(fmem.abs().max()))
    finally:
        decoder.precision, decoder.prune_tol, decoder._needs_validation = saved
    worst.update(n_points=int(N), n_latents=int(latents.shape[0]), precision=saved[0], prune_tol=saved[1], tol=tol)
    worst["ok"] = bool(worst["max_abs_diff"] <= tol and worst["max_pruned"] <= tol)
    if not worst["ok"]:
        msg = ("nphm_amd.validate_numerics: the fast mode (precision %s, prune_tol %g) deviates from the dense fp32 kernel "
               "by %.3e (pruned mass x |f_k| up to %.3e, largest member value %.3g) - above %.1e; use precision='bf16x3' / "
               "a smaller prune_tol for this checkpoint" % (saved[0], saved[1], worst["max_abs_diff"], worst["max_pruned"],
                                                            worst["max_abs_member"], tol))
        if strict:
            raise _lib.NphmAmdError(msg)
        warnings.warn(msg)
    return worst


def validate_training_numerics(decoder, latents: torch.Tensor, n: int = 2048, *, tol: float = 1e-3, strict: bool = False,
                               seed: int = 0) -> dict:
    """The same guard for the HIP training tier (pruned member lists, split-bf16 sweeps, optionally bf16 operand
    storage): one training-style loss (SDF, eikonal and normal-like terms on the spatial gradient) on ``n`` sample points
    per latent, evaluated through ``decoder.value_and_gradient`` and through the composite PyTorch tier (fp32 autograd
    double backward, all 40 members) for the weights and latents at hand.  Returns the largest deviation of the SDF, of
    its spatial gradient and - relative to each tensor's largest entry - of the latent and parameter gradients; warns
    (raises with ``strict=True``) when a gradient deviates by more than ``tol``.  Leaves ``.grad`` of the parameters
    untouched and needs a ROCm device."""
    from .diff_operators import gradient
    lat0 = latents.detach().reshape(-1, 1, decoder.lat_dim).float()
    params = [p for p in decoder.parameters() if p.requires_grad]
    if not params:
        raise ValueError("validate_training_numerics: the decoder has no trainable parameter")
    was_training, backend0 = decoder.training, decoder.train_backend
    decoder.train()
    try:
        with torch.no_grad():
            anchors = decoder.predict_anchors(lat0)
        xyz = torch.stack([_sample_points(anchors[b], n, seed + b)[:n] for b in range(lat0.shape[0])])
        g = torch.Generator().manual_seed(seed)
        nrm = torch.nn.functional.normalize(torch.randn(xyz.shape, generator=g), dim=-1).to(xyz.device)
        res = {}
        for mode in ("composite", "hip"):
            decoder.train_backend = mode
            lat = lat0.clone().requires_grad_()
            x = xyz.clone().requires_grad_()
            fused = decoder.value_and_gradient(x, lat) if mode == "hip" else None
            if mode == "hip" and fused is None:
                raise _lib.NphmAmdError("validate_training_numerics: the HIP training tier does not serve this decoder / device")
            if fused is not None:
                pred, grad, _ = fused
            else:
                pred, _ = decoder(x, lat, None)
                grad = gradient(pred, x)
            loss = pred.abs().mean() + 0.1 * (grad.norm(dim=-1) - 1).abs().mean() + 0.3 * (grad - nrm).norm(dim=-1).mean()
            grads = torch.autograd.grad(loss, [lat] + params)
            res[mode] = (pred.detach(), grad.detach(), grads)
    finally:
        decoder.train_backend = backend0
        decoder.train(was_training)
    (p_c, g_c, gr_c), (p_h, g_h, gr_h) = res["composite"], res["hip"]
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
    out = {"max_abs_diff_sdf": float((p_h - p_c).abs().max()), "max_abs_diff_gradient": float((g_h - g_c).abs().max()),
           "max_rel_diff_latent_grad": rel(gr_h[0], gr_c[0]),
           "max_rel_diff_param_grad": max(rel(a, b) for a, b in zip(gr_h[1:], gr_c[1:])),
           "n_points": int(xyz.shape[0] * xyz.shape[1]), "operands": decoder.train_operands,
           "prune_tol": decoder.prune_tol if decoder.train_prune_tol is None else decoder.train_prune_tol}
    worst = max(out["max_rel_diff_latent_grad"], out["max_rel_diff_param_grad"])
    if worst > tol or out["max_abs_diff_sdf"] > 1e-4:
        msg = f"nphm_amd training tier deviates from the composite tier: {out}"
        if strict:
            raise _lib.NphmAmdError(msg)
        warnings.warn(msg)
    return out
