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
# tiers of a precision mode: (single-pass threshold, two-pass threshold, rounding level of a single-pass member,
# ... of a two-pass member); the f16 modes carry 3 more significand bits at 8x the thresholds (include/nphm_amd.h)
TIERS = {"bf16x3a": (LIGHT_TOL, None, 2.0 ** -8, 0.0), "bf16x3a2": (LIGHT_TOL, MID_TOL, 2.0 ** -8, 2.0 ** -9),
         "f16x3a2": (8e-3, 8e-2, 2.0 ** -11, 2.0 ** -12)}


def _sample_points(anchors: torch.Tensor, n: int, seed: int) -> torch.Tensor:
    """n points: half on a coarse lattice of the reference's extraction box, half Gaussian clouds around the
    anchors (sigma 0.03, 0.1 and 0.25: inside the blend kernels, at their edge, and in the transition to the far
    field where members with tiny weights predict large values - the worst case of the pruning rule)."""
    dev = anchors.device
    g = torch.Generator().manual_seed(seed)
    lo, hi = torch.tensor([-.55, -.5, -.95]), torch.tensor([0.55, 0.75, 0.4])
    m = max(2, int(round((n // 2) ** (1 / 3))))
    ax = [torch.linspace(float(lo[d]), float(hi[d]), m) for d in range(3)]
    lattice = torch.stack(torch.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3)
    k = n - lattice.shape[0]
    a = anchors.reshape(-1, 3).cpu()
    pick = a[torch.randint(0, a.shape[0], (k,), generator=g)]
    u = torch.rand(k, 1, generator=g)
    sigma = torch.where(u < 0.4, torch.tensor(0.03), torch.where(u < 0.75, torch.tensor(0.1), torch.tensor(0.25)))
    near = pick + sigma * torch.randn(k, 3, generator=g)
    return torch.cat([lattice, near], 0).to(dev).contiguous()


# lattice spacing of the reference's extraction box at 512^3 (the finest BASELINE config): the tiles of the sample
# below are the 4 x 4 x 2-voxel tiles one wavefront of the grid kernels works on.  Pruning and the precision tiers
# are decided per wavefront (a member survives / keeps the full product when ANY of the 32 points asks for it), so
# their error depends on the tile geometry: points scattered over a wavefront (nphm_identity_eval_points on a random
# sample) share far more members than a compact tile and measure a several times smaller error than a real
# lattice extraction shows.
_TILE_SPACING = ((0.55 + 0.55) / 511.0, (0.75 + 0.5) / 511.0, (0.4 + 0.95) / 511.0)


def _sample_tiles(anchors: torch.Tensor, n_tiles: int, seed: int):
    """Lattice-ordered sample for nphm_identity_eval_grid_points: an index lattice [rx, ry, rz] whose 4x4x2 index tiles
    are compact physical tiles (512^3 spacing of the reference box) centred on n_tiles sample positions - half uniform
    in the box, half Gaussian clouds around the anchors (``_sample_points``).  Returns (xyz [rx*ry*rz, 3], (rx, ry, rz))."""
    a = 1
    while a * a * a < n_tiles:
        a *= 2
    b = a
    c = max(1, n_tiles // (a * b))
    while a * b * c < n_tiles:
        c *= 2
    T = a * b * c
    centres = _sample_points(anchors, T, seed)                       # [T, 3], device of the anchors
    dev = centres.device
    g = torch.Generator().manual_seed(seed + 7)
    lo, hi = torch.tensor([-.55, -.5, -.95]), torch.tensor([0.55, 0.75, 0.4])
    centres[: T // 2] = (torch.rand(T // 2, 3, generator=g) * (hi - lo) + lo).to(dev)   # the lattice half: uniform in the box
    h = torch.tensor(_TILE_SPACING, device=dev)
    rx, ry, rz = 4 * a, 4 * b, 2 * c
    ix, iy, iz = torch.meshgrid(torch.arange(rx, device=dev), torch.arange(ry, device=dev), torch.arange(rz, device=dev), indexing="ij")
    tile = ((ix // 4) * b + (iy // 4)) * c + (iz // 2)
    off = torch.stack([ix % 4, iy % 4, iz % 2], dim=-1).float() * h
    xyz = centres[tile.reshape(-1)] + off.reshape(-1, 3)
    return xyz.contiguous(), (rx, ry, rz)


def _eval_tiles(lib, decoder, packed, state, xyz, dims, code, prune, stream):
    rx, ry, rz = dims
    out = torch.empty(rx * ry * rz, dtype=torch.float32, device=xyz.device)
    _lib.check(lib.nphm_identity_eval_grid_points(packed.data_ptr(), state.data_ptr(), xyz.data_ptr(), rx, ry, rz, 0, rx, 0,
                                                  float(prune), code, out.data_ptr(), None, None, 0, stream),
               "nphm_identity_eval_grid_points")
    return out


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
    prune_eff, code_eff = decoder.kernel_knobs(dev)              # pinned, or calibrated (numerics = "auto")
    eff = decoder.calibration if decoder.numerics == "auto" else None
    precision_eff = eff["precision"] if eff else decoder.precision
    light_eff, mid_eff = (eff["light_tol"], eff["mid_tol"]) if eff else (decoder.light_tol, decoder.mid_tol)
    saved = (precision_eff, prune_eff, getattr(decoder, "_needs_validation", False))
    decoder._needs_validation = False
    worst = {"max_abs_diff": 0.0, "max_pruned": 0.0, "max_light": 0.0, "max_two_pass": 0.0, "max_abs_sdf": 0.0,
             "max_abs_member": 0.0}
    try:
        with torch.no_grad():
            for r in range(latents.shape[0]):
                lat = latents[r:r + 1]
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
                if points is None:
                    # the comparison proper on compact lattice tiles (what an extraction evaluates); the per-member
                    # analysis below stays on the scattered sample
                    txyz, tdims = _sample_tiles(anchors[0], max(256, n // 32), seed + r)
                    fast = _eval_tiles(lib, decoder, packed, state, txyz, tdims, code_eff, prune_eff, stream)
                    dense = _eval_tiles(lib, decoder, packed, state, txyz, tdims, _lib.NPHM_PREC_F32, -1.0, stream)
                else:
                    fast = run(code_eff, prune_eff)
                    dense = run(_lib.NPHM_PREC_F32, -1.0)
                # every member's value at every point (member-centric kernel, all pairs listed)
                what_all, tiles, plist = _member_point_lists(anchors, pts, -1.0, A)
                fmem = torch.zeros(1, N, A, dtype=torch.float32, device=dev)
                _lib.check(lib.nphm_identity_member_forward(packed.data_ptr(), decoder._packed_bwd(dev).data_ptr(),
                                                            state.data_ptr(), pts.data_ptr(), N, tiles.data_ptr(),
                                                            tiles.shape[0], None, plist.data_ptr(), fmem.data_ptr(), stream),
                           "nphm_identity_member_forward")
                contrib = what_all * fmem.abs()                                   # w_k |f_k|, [1,N,A]
                if prune_eff >= 0:
                    kept = _member_point_lists(anchors, pts, prune_eff, A)[0] > 0
                    pruned = (contrib * (~kept)).sum(dim=2)
                else:
                    pruned = torch.zeros_like(contrib[..., 0])
                t_light, t_mid, r_light, r_mid = TIERS.get(precision_eff, (None, None, 0.0, 0.0))
                t_light = light_eff if (light_eff is not None and t_light is not None) else t_light
                t_mid = mid_eff if (mid_eff is not None and t_mid is not None) else t_mid
                light = contrib * (what_all < t_light) * r_light if t_light is not None else contrib * 0
                # two-pass members: weights rounded to bf16 / f16, a 2^-9 / 2^-12 relative perturbation of the member
                mid = contrib * ((what_all >= t_light) & (what_all < t_mid)) * r_mid if t_mid is not None else contrib * 0
                worst["max_abs_diff"] = max(worst["max_abs_diff"], float((fast - dense).abs().max()))
                worst["max_pruned"] = max(worst["max_pruned"], float(pruned.max()))
                worst["max_light"] = max(worst["max_light"], float(light.max()))
                worst["max_two_pass"] = max(worst["max_two_pass"], float(mid.max()))
                worst["max_abs_sdf"] = max(worst["max_abs_sdf"], float(dense.abs().max()))
                worst["max_abs_member"] = max(worst["max_abs_member"], float(fmem.abs().max()))
    finally:
        decoder._needs_validation = saved[2]
    worst.update(n_points=int(N), n_latents=int(latents.shape[0]), precision=saved[0], prune_tol=saved[1], tol=tol,
                 light_tol=light_eff, mid_tol=mid_eff, numerics=decoder.numerics)
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


# ---- per-checkpoint calibration (numerics = "auto") -------------------------------------------------------------------
# Two coordinates, searched greedily: the pruning budget with every member on the full three-pass product, then the tier
# thresholds at that budget.  Candidates from the fastest to the safest setting:
PRUNE_LADDER = (1e-7, 3e-8, 1e-8, 3e-9, 1e-9, -1.0)
TIER_LADDER = ((8e-3, 8e-2), (4e-3, 4e-2), (2e-3, 2e-2), (1e-3, 1e-2), (5e-4, 5e-3), (2.5e-4, 2.5e-3), (None, None))


def calibrate_numerics(decoder, latents: Optional[torch.Tensor] = None, n: int = 1 << 19, *, target: float = 5e-6,
                       seed: int = 0, device=None) -> dict:
    """Fastest setting of the inference kernels (pruning tolerance, split-f16 tiers) whose error against the dense
    exact-fp32 kernel stays <= ``target`` on ``n`` sample points per latent - n / 32 compact 4x4x2 lattice tiles at the
    512^3 spacing of the reference box, half of them uniform in the box, half in clouds around the anchors
    (``_sample_tiles``) - for the weights the decoder holds NOW.  ``target`` defaults to 5e-6: half of a tenth of the 1e-4 bar (margin for the points
    the sample does not hold; DESIGN.md section 3 compares the sample maximum with full 256^3 volumes).  ``latents``
    [R, lat_dim]: the codes to calibrate with (``kernel_knobs`` passes the code of the call that triggers it; default:
    the zero code = mean anchors).  Returns {"precision", "light_tol", "mid_tol", "prune_tol",
    "error", "searched": [(setting, error)]}; what ``decoder.kernel_knobs`` uses when ``decoder.numerics == "auto"``."""
    lib = _lib.load()
    dev = torch.device(device) if device is not None else next(decoder.parameters()).device
    if dev.type != "cuda" or not decoder.hip_supported():
        raise _lib.NphmAmdError("calibrate_numerics needs the HIP-backed NPHM identity field on a ROCm device")
    if latents is None:
        latents = torch.zeros(1, decoder.lat_dim, device=dev)
    latents = latents.reshape(-1, decoder.lat_dim).to(device=dev, dtype=torch.float32)
    searched = []
    with torch.no_grad():
        cases = []
        for r in range(latents.shape[0]):
            packed, state, anchors = decoder.prepare_latent(latents[r:r + 1])
            xyz, dims = _sample_tiles(anchors[0], max(256, n // 32), seed + r)
            cases.append((packed, state, xyz, dims))
        stream = torch.cuda.current_stream(dev).cuda_stream

        def run(case, code, prune):
            packed, state, xyz, dims = case
            return _eval_tiles(lib, decoder, packed, state, xyz, dims, code, prune, stream)
        dense = [run(c, _lib.NPHM_PREC_F32, -1.0) for c in cases]

        def err_of(precision, light, mid, prune):
            code = decoder.precision_code(precision, light, mid)
            e = max(float((run(c, code, prune) - d).abs().max()) for c, d in zip(cases, dense))
            searched.append(({"precision": precision, "light_tol": light, "mid_tol": mid, "prune_tol": prune}, e))
            return e
        # 1. pruning budget, three-pass product everywhere: half of the target
        prune, e_prune = PRUNE_LADDER[-1], None
        for cand in PRUNE_LADDER:
            e = err_of("f16x3", None, None, cand)
            if e <= 0.5 * target or cand == PRUNE_LADDER[-1]:
                prune, e_prune = cand, e
                break
        # 2. tier thresholds at that budget: the whole target
        choice, e_choice = ("f16x3", None, None), e_prune
        for light, mid in TIER_LADDER[:-1]:
            e = err_of("f16x3a2", light, mid, prune)
            if e <= target:
                choice, e_choice = ("f16x3a2", light, mid), e
                break
    return {"precision": choice[0], "light_tol": choice[1], "mid_tol": choice[2], "prune_tol": prune, "error": e_choice,
            "target": target, "n_points": int(n), "n_latents": int(latents.shape[0]), "searched": searched}


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
