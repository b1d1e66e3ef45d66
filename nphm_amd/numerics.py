"""Numerics guard of the NPHM identity field's fast modes (pruning + adaptive precision).

The shipped default drops, per point, the smallest blend weights while they sum to <= 40 * prune_tol and runs
members below ``NPHM_LIGHT_TOL`` (1e-3) normalised weight single-pass; both are validated on seeded
random-init weights (profiles/NOTES.md section 3), where every member's |f_k| is small.  A trained checkpoint can
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


def _tile_surface_weights(dense: torch.Tensor, dims):
    """(mask of the sample points within ~1.5 lattice spacings of the zero level set, 1 / |grad f|) from the dense values
    on the compact tiles of ``_sample_tiles``: one-sided differences inside every 4x4x2 tile.  An SDF error e moves the
    extracted surface by e / |grad f| there - what the north star's mesh criterion (Chamfer within 1e-5) measures; a
    trained field has |grad f| ~ 1, a seeded one ~ 0.1."""
    rx, ry, rz = dims
    f = dense.view(rx // 4, 4, ry // 4, 4, rz // 2, 2)
    hx, hy, hz = _TILE_SPACING

    def one_sided(t, axis, h):
        d = torch.diff(t, dim=axis) / h
        last = d.narrow(axis, d.shape[axis] - 1, 1)
        return torch.cat([d, last], dim=axis)
    g = torch.sqrt(one_sided(f, 1, hx) ** 2 + one_sided(f, 3, hy) ** 2 + one_sided(f, 5, hz) ** 2).reshape(-1)
    near = dense.abs() <= 1.5 * max(hx, hy, hz) * g
    return near, 1.0 / g.clamp_min(1e-3)


def _eval_tiles(lib, decoder, packed, state, xyz, dims, code, prune, stream):
    rx, ry, rz = dims
    out = torch.empty(rx * ry * rz, dtype=torch.float32, device=xyz.device)
    _lib.check(lib.nphm_identity_eval_grid_points(packed.data_ptr(), state.data_ptr(), xyz.data_ptr(), rx, ry, rz, 0, rx, 0,
                                                  float(prune), code, out.data_ptr(), None, None, 0, stream),
               "nphm_identity_eval_grid_points")
    return out


def _kept_mask(rule: torch.Tensor, prune_tol: float) -> torch.Tensor:
    """The pruning rule of the kernels (eval_kernel.hip: blend_masks) per point on normalised rule weights [..., A]: the
    cut is the largest of the candidates {1, 2, 4, 8, 16, 40} x prune_tol whose not-larger weights sum to <= 40 prune_tol."""
    budget = 40.0 * prune_tol
    cut = torch.full(rule.shape[:-1], prune_tol, dtype=rule.dtype, device=rule.device)
    for c in (2.0, 4.0, 8.0, 16.0, 40.0):
        ok = torch.where(rule <= c * prune_tol, rule, torch.zeros_like(rule)).sum(dim=-1) <= budget
        cut = torch.where(ok, torch.full_like(cut, c * prune_tol), cut)
    return rule > cut[..., None]


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
    bounds_eff = eff.get("bounds") if eff else None
    saved = (precision_eff, prune_eff, getattr(decoder, "_needs_validation", False))
    decoder._needs_validation = False
    worst = {"max_abs_diff": 0.0, "max_pruned": 0.0, "max_light": 0.0, "max_two_pass": 0.0, "max_abs_sdf": 0.0,
             "max_abs_member": 0.0}
    try:
        with torch.no_grad():
            for r in range(latents.shape[0]):
                lat = latents[r:r + 1]
                # (the knobs under examination are (prune_eff, code_eff) above: no second decision - calibration, per-latent
                # verification, guard of a pinned mode - inside the prologue)
                packed, state, anchors = decoder.prepare_latent(lat, bounds=bounds_eff)
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
                # what the rules of the kernels see: the normalised weight times the member's magnitude bound
                rule = what_all
                if bounds_eff is not None:
                    dk = torch.cat([(pts[0][:, None, :] - anchors[0][None]).norm(dim=-1) + 1e-5,
                                    torch.zeros(N, 1, device=dev)], dim=1)[None]
                    rule = what_all * (bounds_eff[:, 0] + bounds_eff[:, 1] * dk + bounds_eff[:, 2] * dk * dk)
                if prune_eff >= 0:
                    pruned = (contrib * (~_kept_mask(rule, prune_eff))).sum(dim=2)
                else:
                    pruned = torch.zeros_like(contrib[..., 0])
                t_light, t_mid, r_light, r_mid = TIERS.get(precision_eff, (None, None, 0.0, 0.0))
                t_light = light_eff if (light_eff is not None and t_light is not None) else t_light
                t_mid = mid_eff if (mid_eff is not None and t_mid is not None) else t_mid
                light = contrib * (rule < t_light) * r_light if t_light is not None else contrib * 0
                # two-pass members: weights rounded to bf16 / f16, a 2^-9 / 2^-12 relative perturbation of the member
                mid = contrib * ((rule >= t_light) & (rule < t_mid)) * r_mid if t_mid is not None else contrib * 0
                worst["max_abs_diff"] = max(worst["max_abs_diff"], float((fast - dense).abs().max()))
                worst["max_pruned"] = max(worst["max_pruned"], float(pruned.max()))
                worst["max_light"] = max(worst["max_light"], float(light.max()))
                worst["max_two_pass"] = max(worst["max_two_pass"], float(mid.max()))
                worst["max_abs_sdf"] = max(worst["max_abs_sdf"], float(dense.abs().max()))
                worst["max_abs_member"] = max(worst["max_abs_member"], float(fmem.abs().max()))
    finally:
        decoder._needs_validation = saved[2]
    worst.update(n_points=int(N), n_latents=int(latents.shape[0]), precision=saved[0], prune_tol=saved[1], tol=tol,
                 light_tol=light_eff, mid_tol=mid_eff, numerics=decoder.numerics, member_bounds=bounds_eff is not None)
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
# 1. member magnitude bounds B_k(d) >= |f_k| (quadratic upper envelopes in the distance to the anchor, fitted on a sample of
#    member values, x BOUND_SAFETY): the pruning rule and the tiers then act on w_k B_k(d_k) - the size of a member's
#    term in SDF units - so that 40 * prune_tol bounds the pruning error of ANY checkpoint, whatever its members do far
#    from their anchors (a trained member predicts tens of SDF units there, a seeded one 0.1);
# 2. two coordinates searched greedily from the fastest to the safest candidate: the pruning budget with every member on
#    the three-pass product, then the tier thresholds at that budget.
BOUND_SAFETY = 2.0
# Rungs of the calibration search, coarsest (fastest) first.  Pruning budgets in 1-2-5 steps; tier thresholds in the kernel's
# own half-octave codes (FastEnsembleDeepSDFMirrored._tier_code), light = 2^-5 ... 2^-14.5, mid = 8 x light to start with
# (MID_WIDEN: then widened on its own while the error allows - round 4 measured a x4 wider two-pass tier at NO change of the
# full-volume error, and a 1-2-5 pruning rung between the old 3e-7 / 1e-7 at 4.7e-6 instead of 3.7e-6: 6 % of throughput
# between two rungs of the coarse ladders).
PRUNE_LADDER = (1e-6, 5e-7, 2e-7, 1e-7, 5e-8, 2e-8, 1e-8, 5e-9, 2e-9, 1e-9, -1.0)
TIER_LADDER = tuple((2.0 ** (-5 - 0.5 * i), 2.0 ** (-2 - 0.5 * i)) for i in range(20)) + ((None, None),)
MID_WIDEN = (2.0, 4.0, 8.0)
PRUNE_SHARE_MAX = 0.8      # a pruning rung is a candidate while its error alone stays below this share of the target


def fit_member_bounds(decoder, lat: torch.Tensor, n: int = 1 << 16, seed: int = 0) -> torch.Tensor:
    """[40,4] device tensor (b0, b1, b2, 0) per member: b0 + b1 d + b2 d^2 >= BOUND_SAFETY * |f_k(x)| for the sampled
    points x at distance d from anchor k (background member: constant), non-negative coefficients.  Member values by
    the member-centric kernel on lattice + near-anchor sample points of the latent ``lat`` [1, lat_dim]."""
    from .ensembled_deepsdf import _member_point_lists
    lib = _lib.load()
    dev = lat.device
    A = decoder.num_kps + 1
    with torch.no_grad():
        packed, state, anchors = decoder.prepare_latent(lat, bounds=None)
        pts = _sample_points(anchors[0], n, seed)[None]
        N = pts.shape[1]
        stream = torch.cuda.current_stream(dev).cuda_stream
        _, tiles, plist = _member_point_lists(anchors, pts, -1.0, A)
        fmem = torch.zeros(1, N, A, dtype=torch.float32, device=dev)
        _lib.check(lib.nphm_identity_member_forward(packed.data_ptr(), decoder._packed_bwd(dev).data_ptr(), state.data_ptr(),
                                                    pts.data_ptr(), N, tiles.data_ptr(), tiles.shape[0], None, plist.data_ptr(),
                                                    fmem.data_ptr(), stream), "nphm_identity_member_forward")
        f = fmem[0].abs()                                                             # [N, 40]
        d = (pts[0][:, None, :] - anchors[0][None]).norm(dim=-1)                      # [N, 39]
        # maxima of |f_k| over 24 distance bins of width 0.1, on the device (39 x np.maximum.at over 65 536 points was
        # most of the 0.13 s this function took): only the [39, 24] table travels
        bins = torch.clamp((d.double() / 0.1).floor().long(), 0, 23)                  # = np.digitize(d, linspace(0, 2.4, 25)) - 1
        table = torch.zeros(A - 1, 24, dtype=torch.float32, device=dev)
        table.scatter_reduce_(1, bins.t().contiguous(), f[:, :A - 1].t().contiguous(), reduce="amax", include_self=True)
        table = table.double().cpu().numpy()
        f_bg = float(f[:, A - 1].max())
    out = np.zeros((A, 4), np.float32)
    edges = np.linspace(0.0, 2.4, 25)
    for k in range(A - 1):
        m = np.maximum.accumulate(table[k])                  # non-decreasing envelope of the bin maxima
        x = edges[1:]                                        # evaluated at the far edge of a bin
        # non-negative least squares on (1, d, d^2), then lifted onto the envelope
        from scipy.optimize import nnls
        Amat = np.stack([np.ones_like(x), x, x * x], 1)
        c, _ = nnls(Amat, m)
        lift = float(np.max(m - Amat @ c))
        c[0] += max(lift, 0.0)
        out[k, :3] = BOUND_SAFETY * c
    out[A - 1, 0] = BOUND_SAFETY * f_bg
    out[:, 0] = np.maximum(out[:, 0], 1e-6)
    return torch.from_numpy(out).to(dev)


def calibrate_numerics(decoder, latents: Optional[torch.Tensor] = None, n: int = 1 << 19, *, target: float = 5e-6,
                       target_surface: float = 5e-6, seed: int = 0, device=None, member_bounds: bool = True,
                       refine_band: Optional[float] = None) -> dict:
    """Fastest setting of the inference kernels (member magnitude bounds, pruning budget, split-f16 tiers) whose error
    against the dense exact-fp32 kernel stays <= ``target`` on ``n`` sample points per latent - n / 32 compact 4x4x2
    lattice tiles at the 512^3 spacing of the reference box, half of them uniform in the box, half in clouds around the
    anchors (``_sample_tiles``: the tile geometry of an extraction at its finest BASELINE resolution, where a
    wavefront's points share the fewest members - the worst case of the per-wavefront rules) - for the weights the
    decoder holds NOW.  ``target`` defaults to 5e-6: full 256^3 extractions then stay below 1e-5, a decade inside the 1e-4
    bar (their maximum over 16.7 M voxels is up to twice the sample's, profiles/NOTES.md section 3a).  ``target_surface`` bounds the
    MEAN displacement of the zero level set, |error| / |grad f| over the sample points next to it (the mesh criterion of
    the north star: Chamfer within 1e-5; only binding for fields with small gradients, e.g. seeded weights).
    ``refine_band`` (default: target): the sign-safe refinement band every candidate runs with - values that close to
    zero are re-evaluated at full precision, so the fast setting cannot flip the sign of a voxel (mesh topology).
    ``latents`` [R, lat_dim]: the codes to
    calibrate with (``kernel_knobs`` passes the code of the call that triggers it; default: the zero code = mean
    anchors).  Returns {"precision", "light_tol", "mid_tol", "prune_tol", "bounds" [40,4] or None, "error",
    "terms_per_point", "searched": [(setting, error, product terms per sample point)]}; what the inference entry points use when ``decoder.numerics == "auto"``."""
    lib = _lib.load()
    dev = torch.device(device) if device is not None else next(decoder.parameters()).device
    if dev.type != "cuda" or not decoder.hip_supported():
        raise _lib.NphmAmdError("calibrate_numerics needs the HIP-backed NPHM identity field on a ROCm device")
    if latents is None:
        latents = torch.zeros(1, decoder.lat_dim, device=dev)
    latents = latents.reshape(-1, decoder.lat_dim).to(device=dev, dtype=torch.float32)
    searched = []
    with torch.no_grad():
        bounds = None
        if member_bounds:
            fits = torch.stack([fit_member_bounds(decoder, latents[r:r + 1], seed=seed + r) for r in range(latents.shape[0])])
            bounds = fits.max(dim=0).values.contiguous()
        cases = []
        for r in range(latents.shape[0]):
            packed, state, anchors = decoder.prepare_latent(latents[r:r + 1], bounds=bounds)
            xyz, dims = _sample_tiles(anchors[0], max(256, n // 32), seed + r)
            cases.append((packed, state, xyz, dims))
        stream = torch.cuda.current_stream(dev).cuda_stream

        stats = torch.zeros(16, dtype=torch.int64, device=dev)

        def run(case, code, prune, count=False):
            packed, state, xyz, dims = case
            rx, ry, rz = dims
            out = torch.empty(rx * ry * rz, dtype=torch.float32, device=xyz.device)
            _lib.check(lib.nphm_identity_eval_grid_points(packed.data_ptr(), state.data_ptr(), xyz.data_ptr(), rx, ry, rz, 0, rx, 0,
                                                          float(prune), code, out.data_ptr(), stats.data_ptr() if count else None,
                                                          None, 0, stream), "nphm_identity_eval_grid_points")
            return out
        dense = [run(c, _lib.NPHM_PREC_F32, -1.0) for c in cases]
        surf = [_tile_surface_weights(d, c[3]) for c, d in zip(cases, dense)]
        scale = target / target_surface          # errors are compared in units of `target`

        band = target if refine_band is None else refine_band

        def err_of(precision, light, mid, prune):
            """(error, cost) of a setting; cost = product terms per sample point from the kernel's own counters: 3 x three-pass +
            2 x two-pass + 1 x one-pass (point, member) pairs, + 1 per pair for its fixed work"""
            code = decoder.precision_code(precision, light, mid, band)
            e = 0.0
            stats.zero_()
            for c, d, (near, inv_g) in zip(cases, dense, surf):
                diff = (run(c, code, prune, count=True) - d).abs()
                e = max(e, float(diff.max()))
                if int(near.sum()) >= 64:
                    e = max(e, scale * float((diff * inv_g)[near].mean()))
            st = stats.cpu().numpy()
            total, pts, two, one = int(st[0]), max(int(st[1]), 1), int(st[14]), int(st[15])
            cost = (3 * (total - two - one) + 2 * two + one + total) / pts      # (+ one term per pair: its weight stream and epilogue)
            searched.append(({"precision": precision, "light_tol": light, "mid_tol": mid, "prune_tol": prune}, e, cost))
            return e, cost

        def tiers_at(prune, e_prune, cost_prune):
            """tier thresholds at a pruning budget: the coarsest rung inside the whole target, then the two-pass tier alone,
            wider (its members stop running the third pass), under the same bound"""
            choice, e_choice, cost = ("f16x3", None, None), e_prune, cost_prune
            for light, mid in TIER_LADDER[:-1]:
                e, c = err_of("f16x3a2", light, mid, prune)
                if e <= target:
                    choice, e_choice, cost = ("f16x3a2", light, mid), e, c
                    break
            if choice[0] == "f16x3a2":
                light, mid0 = choice[1], choice[2]
                for f in MID_WIDEN:
                    e, c = err_of("f16x3a2", light, mid0 * f, prune)
                    if e > target:
                        break
                    choice, e_choice, cost = ("f16x3a2", light, mid0 * f), e, c
            return choice, e_choice, cost
        # 1. candidates for the pruning budget (three-pass product everywhere): the coarsest rungs that leave the tiers a fifth
        #    of the target, down to the first that leaves them half of it (round 4: the greedy "half" rule passed over a rung
        #    that is faster AND more accurate in the end - a 2e-7 budget with a half-octave smaller one-pass threshold)
        rungs = []
        for cand in PRUNE_LADDER:
            e, c = err_of("f16x3", None, None, cand)
            if e <= PRUNE_SHARE_MAX * target or cand == PRUNE_LADDER[-1]:
                rungs.append((cand, e, c))
                if e <= 0.5 * target or len(rungs) == 3 or cand == PRUNE_LADDER[-1]:
                    break
        # 2. the tiers at every candidate; the cheapest setting (product terms per sample point) wins
        best = None
        for cand, e, c in rungs:
            choice, e_choice, cost = tiers_at(cand, e, c)
            if best is None or cost < best[3]:
                best = (cand, choice, e_choice, cost)
        prune, choice, e_choice, cost_choice = best
    return {"precision": choice[0], "light_tol": choice[1], "mid_tol": choice[2], "prune_tol": prune, "bounds": bounds,
            "refine_band": band, "latents": latents.detach().clone(),
            "error": e_choice, "terms_per_point": cost_choice, "target": target, "n_points": int(n), "n_latents": int(latents.shape[0]),
            "searched": searched}


def sample_error(decoder, lat: torch.Tensor, *, precision, light_tol, mid_tol, prune_tol, refine_band=None, bounds=None,
                 n_tiles: int = 512, seed: int = 0, exact: bool = False) -> float:
    """max |fast setting - reference| on ``n_tiles`` compact 4x4x2 lattice tiles (``_sample_tiles``) of the latent ``lat``
    [1, lat_dim] for the decoder's CURRENT weights.  Reference: all 40 members on the three-pass split-f16 product (within
    1e-6 of fp32 on every checkpoint measured, 16x the fp32 MFMA rate: 0.3 ms for 512 tiles), or with ``exact`` the
    fp32-MFMA kernel.  One synchronisation.  The cheap check behind the per-latent verification of numerics = "auto" and the
    guard of pinned fast modes (FastEnsembleDeepSDFMirrored.kernel_knobs)."""
    lib = _lib.load()
    dev = lat.device
    with torch.no_grad():
        packed, state, anchors = decoder.prepare_latent(lat.reshape(1, -1), bounds=bounds)
        xyz, dims = _sample_tiles(anchors[0], n_tiles, seed)
        stream = torch.cuda.current_stream(dev).cuda_stream
        code = decoder.precision_code(precision, light_tol, mid_tol, refine_band)
        fast = _eval_tiles(lib, decoder, packed, state, xyz, dims, code, prune_tol, stream)
        ref = _eval_tiles(lib, decoder, packed, state, xyz, dims, _lib.NPHM_PREC_F32 if exact else _lib.NPHM_PREC_F16X3, -1.0, stream)
        return float((fast - ref).abs().max())


_BASE_MODE = {"bf16x3a": "bf16x3", "bf16x3a2": "bf16x3", "f16x3a2": "f16x3"}


def clamp_pinned(decoder, lat: torch.Tensor, *, precision, light_tol, mid_tol, prune_tol, refine_band=None,
                 guard: float = 2e-5, n_tiles: int = 512) -> dict:
    """Guard of a PINNED fast mode (numerics = "fixed" with a tiered precision and / or a pruning budget): its error on the
    checkpoint at hand, measured once per weight version with the first call's latent; above ``guard`` the setting is
    tightened - tier thresholds / 4 per step down to the mode without tiers, then the pruning budget / 10 per step down to
    no pruning - until it is inside, and a warning names what was asked and what runs instead.  (Seeded-weight thresholds
    on a trained checkpoint: f16x3a2 at 8e-3 / 8e-2 measured 2.1e-4, twice the 1e-4 bar, in round 3.)"""
    light0, mid0 = TIERS[precision][:2] if precision in TIERS else (None, None)
    light = light_tol if light_tol is not None else light0
    mid = mid_tol if mid_tol is not None else mid0
    asked = {"precision": precision, "light_tol": light, "mid_tol": mid, "prune_tol": prune_tol}
    cur = dict(asked)
    steps = []
    for _ in range(24):
        e = sample_error(decoder, lat, precision=cur["precision"], light_tol=cur["light_tol"], mid_tol=cur["mid_tol"],
                         prune_tol=cur["prune_tol"], refine_band=refine_band, n_tiles=n_tiles)
        steps.append((dict(cur), e))
        if e <= guard:
            break
        if cur["precision"] in _BASE_MODE:
            if cur["light_tol"] is not None and cur["light_tol"] > 2e-5:
                cur["light_tol"] = cur["light_tol"] / 4.0
                cur["mid_tol"] = None if cur["mid_tol"] is None else cur["mid_tol"] / 4.0
            else:
                cur.update(precision=_BASE_MODE[cur["precision"]], light_tol=None, mid_tol=None)
        elif cur["prune_tol"] >= 0:
            cur["prune_tol"] = cur["prune_tol"] / 10.0 if cur["prune_tol"] > 2e-10 else -1.0
        else:
            break                                            # all members, full products: nothing left to tighten
    clamped = cur != asked
    out = dict(cur, error=steps[-1][1], asked=asked, asked_error=steps[0][1], clamped=clamped, guard=guard, steps=steps)
    if clamped:
        warnings.warn("nphm_amd: the pinned fast mode %s deviates from the fp32-equivalent kernel by %.2e on this checkpoint "
                      "(guard %.0e); running %s instead (%.2e).  numerics = 'auto' calibrates the knobs per checkpoint; "
                      "decoder.pinned_guard = None disables this check."
                      % (asked, steps[0][1], guard, {k: cur[k] for k in asked}, steps[-1][1]))
    return out


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
