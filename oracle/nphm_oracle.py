"""CPU oracle for the NPHM neural-field evaluation hot path (TEST INFRASTRUCTURE ONLY).

This file is a plain-numpy fp32 restatement of the reference's arithmetic for the
path named by BASELINE.json:north_star.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product (``nphm_amd``) never does.

Parity status: PINNED.  The reference ships no golden vectors, so
the oracle is pinned against outputs of the reference's own PyTorch modules
executed in the build container (``tests/golden/make_golden.py`` imports
``/root/reference/src`` and writes ``tests/golden/*.npz``);
``tests/test_oracle_golden.py`` checks this file against those fixtures.

Every function cites the reference file:line it restates (paths relative to
``/root/reference``).
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32
SQRT2 = F32(np.sqrt(2))  # reference divides by np.sqrt(2) (float64 scalar -> weak type)


# ----------------------------------------------------------------------------
# activations
# ----------------------------------------------------------------------------
def softplus100(x: np.ndarray, beta: float = 100.0, threshold: float = 20.0) -> np.ndarray:
    """nn.Softplus(beta=100) with PyTorch's default threshold=20.

    src/NPHM/models/EnsembledDeepSDF.py:99, src/NPHM/models/deepSDF.py:58.
    PyTorch returns ``x`` exactly when ``x*beta > threshold``.
    """
    x = x.astype(F32, copy=False)
    z = x * F32(beta)
    with np.errstate(over="ignore"):
        soft = np.log1p(np.exp(np.minimum(z, F32(threshold)))) / F32(beta)
    return np.where(z > F32(threshold), x, soft).astype(F32)


def relu(x: np.ndarray) -> np.ndarray:
    return np.maximum(x, F32(0))


# ----------------------------------------------------------------------------
# EnsembledLinear / EnsembledDeepSDF   (EnsembledDeepSDF.py:8-126)
# ----------------------------------------------------------------------------
def member_to_set(k: int, n_symm: int) -> int:
    """Weight set used by ensemble member k (EnsembledDeepSDF.py:43-45):
    the first n_symm sets are each repeated twice (repeat_interleave(2)), the
    rest are used once."""
    return k // 2 if k < 2 * n_symm else n_symm + (k - 2 * n_symm)


def ensembled_linear(weight: np.ndarray, bias: np.ndarray, x: np.ndarray, n_symm: int) -> np.ndarray:
    """EnsembledLinear.forward (EnsembledDeepSDF.py:37-55).

    weight [A-n_symm, out, in], bias [A-n_symm, out], x [A, P, in] -> [A, P, out].
    """
    A = x.shape[0]
    out = np.empty((A, x.shape[1], weight.shape[1]), F32)
    for k in range(A):
        s = member_to_set(k, n_symm)
        out[k] = x[k] @ weight[s].T + bias[s][None, :]
    return out


def ensembled_deepsdf(params: dict, prefix: str, xyz: np.ndarray, lat: np.ndarray,
                      n_symm: int, nlayers: int) -> np.ndarray:
    """EnsembledDeepSDF.forward (EnsembledDeepSDF.py:101-126).

    xyz [A,B,N,3], lat [A,B,N,F] -> [A,B,N,out].  dims=[d_in]+[hidden]*nlayers+[out],
    skip_in=[nlayers//2] (:80-84); at the skip layer x=[x ‖ inp]/sqrt(2) (:115-116);
    Softplus(100) after every layer but the last (:120-121).
    """
    A, B, N, _ = xyz.shape
    inp = np.concatenate([xyz, lat], axis=-1).reshape(A, B * N, -1).astype(F32)
    x = inp
    num_layers = nlayers + 2
    skip_in = [nlayers // 2]
    for layer in range(num_layers - 1):
        W = params[f"{prefix}lin{layer}.weight"]
        b = params[f"{prefix}lin{layer}.bias"]
        if layer in skip_in:
            x = (np.concatenate([x, inp], -1) / SQRT2).astype(F32)
        x = ensembled_linear(W, b, x, n_symm)
        if layer < num_layers - 2:
            x = softplus100(x)
    return x.reshape(A, B, N, -1)


# ----------------------------------------------------------------------------
# Gaussian blend   (EnsembledDeepSDF.py:129-150, duplicate deepSDF.py:92-115)
# ----------------------------------------------------------------------------
def sample_point_feature(q: np.ndarray, p: np.ndarray, fea: np.ndarray,
                         var: float = 0.1 ** 2, background: bool = False) -> np.ndarray:
    """q [B,N,3], p [B,K,3], fea [B,N,K(+1),C] -> [B,N,C]."""
    diff = (p[:, None, :, :] - q[:, :, None, :]).astype(F32)
    nrm = np.sqrt((diff * diff).sum(-1, dtype=F32)).astype(F32)
    dist = -((nrm + F32(10e-6)) ** 2)
    if background:
        const = np.full_like(dist[:, :, :1], F32(-0.2))
        dist = np.concatenate([dist, const], axis=-1)
    weight = np.exp((dist / F32(var)).astype(F32)).astype(F32)
    weight = weight / (weight.sum(axis=2, dtype=F32)[..., None] + F32(1e-6))
    return (weight[..., None] * fea).sum(axis=2, dtype=F32).astype(F32)


def blend_weights(q: np.ndarray, anchors: np.ndarray) -> np.ndarray:
    """Normalised blend weights [N,40] for q [N,3], anchors [39,3]
    (same arithmetic as sample_point_feature with background=True)."""
    diff = (anchors[None, :, :] - q[:, None, :]).astype(F32)
    nrm = np.sqrt((diff * diff).sum(-1, dtype=F32)).astype(F32)
    dist = -((nrm + F32(10e-6)) ** 2)
    dist = np.concatenate([dist, np.full_like(dist[:, :1], F32(-0.2))], axis=-1)
    w = np.exp((dist / F32(0.1 ** 2)).astype(F32)).astype(F32)
    return (w / (w.sum(axis=1, dtype=F32)[:, None] + F32(1e-6))).astype(F32)


# ----------------------------------------------------------------------------
# FastEnsembleDeepSDFMirrored.forward   (EnsembledDeepSDF.py:203-267)
# ----------------------------------------------------------------------------
def mlp_pos(params: dict, z_glob: np.ndarray) -> np.ndarray:
    """nn.Sequential(Linear, ReLU, Linear, ReLU, Linear) (EnsembledDeepSDF.py:194-200)."""
    h = relu(z_glob @ params["mlp_pos.0.weight"].T + params["mlp_pos.0.bias"])
    h = relu(h @ params["mlp_pos.2.weight"].T + params["mlp_pos.2.bias"])
    return (h @ params["mlp_pos.4.weight"].T + params["mlp_pos.4.bias"]).astype(F32)


def nphm_identity_forward(params: dict, anchors_mean: np.ndarray, xyz: np.ndarray,
                          lat_rep: np.ndarray, *, training: bool = False,
                          lat_dim_glob: int = 64, lat_dim_loc: int = 32, n_loc: int = 39,
                          n_symm: int = 16, nlayers: int = 4):
    """Returns (sdf [B,N,1], anchors [B,n_loc,3]).

    params: numpy state_dict of the reference module; anchors_mean [n_loc,3] fp32
    (the module's ``self.anchors.squeeze()``).
    """
    xyz = np.asarray(xyz, F32)
    lat_rep = np.asarray(lat_rep, F32)
    if xyz.ndim < 3:                                   # :218-219
        xyz = xyz[None]
    B, N, _ = xyz.shape
    if lat_rep.shape[1] == 1:                          # :222-223
        lat_rep = np.repeat(lat_rep, N, axis=1)
    A = n_loc + 1
    assert lat_rep.shape[-1] == lat_dim_glob + A * lat_dim_loc   # :225

    anchors = mlp_pos(params, lat_rep[:, 0, :lat_dim_glob]).reshape(B, n_loc, 3)   # :228
    anchors = (anchors + anchors_mean.reshape(1, n_loc, 3)).astype(F32)            # :229

    # local coordinates; last member uses global coords (:240-241)
    off = np.concatenate([anchors, np.zeros((B, 1, 3), F32)], axis=1)             # [B,A,3]
    coords = (xyz[:, :, None, :] - off[:, None, :, :]).astype(F32)                # [B,N,A,3]
    coords[:, :, 1:2 * n_symm:2, 0] *= F32(-1)                                    # :244

    t1 = np.broadcast_to(lat_rep[:, :, None, :lat_dim_glob], (B, N, A, lat_dim_glob))
    t2 = lat_rep[:, :, lat_dim_glob:].reshape(B, N, A, lat_dim_loc)
    cond = np.concatenate([t1, t2], axis=-1)                                      # :247-252

    sdf = np.empty((A, B, N, 1), F32)
    # member-by-member to bound memory (the arithmetic per member is independent)
    for k in range(A):
        sdf[k:k + 1] = _ensemble_single(params, k, coords[:, :, k], cond[:, :, k], n_symm, nlayers)
    if not training:                                                              # :260-261
        sdf[:, :, -1, 0] = F32(1)
    sdf = np.transpose(sdf, (1, 2, 0, 3))                                         # [B,N,A,1]
    pred = sample_point_feature(xyz[..., :3], anchors, sdf, background=True, var=0.1 ** 2)  # :265
    return pred, anchors


def _ensemble_single(params, k, coords_k, cond_k, n_symm, nlayers):
    """One member of EnsembledDeepSDF (same maths as ensembled_deepsdf restricted
    to member k)."""
    s = member_to_set(k, n_symm)
    B, N, _ = coords_k.shape
    inp = np.concatenate([coords_k, cond_k], -1).reshape(B * N, -1).astype(F32)
    x = inp
    num_layers = nlayers + 2
    skip_in = [nlayers // 2]
    for layer in range(num_layers - 1):
        W = params[f"ensembled_deep_sdf.lin{layer}.weight"][s]
        b = params[f"ensembled_deep_sdf.lin{layer}.bias"][s]
        if layer in skip_in:
            x = (np.concatenate([x, inp], -1) / SQRT2).astype(F32)
        x = (x @ W.T + b[None, :]).astype(F32)
        if layer < num_layers - 2:
            x = softplus100(x)
    return x.reshape(1, B, N, -1)


# ----------------------------------------------------------------------------
# DeepSDF / DeformationNetwork   (deepSDF.py:6-89, :118-239)
# ----------------------------------------------------------------------------
def deepsdf_forward(params: dict, prefix: str, xyz: np.ndarray, lat_rep: np.ndarray,
                    nlayers: int = 8, beta: float = 100.0) -> np.ndarray:
    """DeepSDF.forward without positional encoding (deepSDF.py:64-89).
    dims=[d_in]+[H]*nlayers+[out]; skip_in=[nlayers//2]; layer skip-1 emits H-d_in."""
    inp = np.concatenate([np.asarray(xyz, F32), np.asarray(lat_rep, F32)], axis=-1)
    x = inp
    num_layers = nlayers + 2
    skip_in = [nlayers // 2]
    for layer in range(num_layers - 1):
        W = params[f"{prefix}lin{layer}.weight"]
        b = params[f"{prefix}lin{layer}.bias"]
        if layer in skip_in:
            x = (np.concatenate([x, inp], -1) / SQRT2).astype(F32)
        x = (x @ W.T + b).astype(F32)
        if layer < num_layers - 2:
            x = softplus100(x, beta) if beta > 0 else relu(x)
    return x


def deformation_forward(params: dict, xyz: np.ndarray, lat_rep: np.ndarray, anchors: np.ndarray,
                        *, lat_dim_expr: int = 200, nlayers: int = 6):
    """DeformationNetwork.forward, mode 'compress', eval mode (deepSDF.py:212-223,237-239).

    xyz [B,N,3]; lat_rep [B,N or 1,L]; anchors [B,K,3] or [B,N,K,3].
    Returns (pred[..., :3], pred[..., -1:]).
    """
    xyz = np.asarray(xyz, F32)
    if xyz.ndim < 3:
        xyz = xyz[None]
    B, N, _ = xyz.shape
    lat_rep = np.asarray(lat_rep, F32)
    if lat_rep.shape[1] == 1:
        lat_rep = np.repeat(lat_rep, N, axis=1)
    anchors = np.asarray(anchors, F32)
    a0 = anchors[:, 0] if anchors.ndim == 4 else anchors              # row 0 only (:218-219)
    concat0 = np.concatenate([lat_rep[:, 0, :-lat_dim_expr], a0.reshape(B, -1)], axis=-1)
    comp = (concat0 @ params["compressor.0.weight"].T + params["compressor.0.bias"]).astype(F32)
    comp = np.repeat(comp[:, None, :], N, axis=1)
    cond = np.concatenate([comp, lat_rep[..., -lat_dim_expr:]], axis=-1)
    pred = deepsdf_forward(params, "defDeepSDF.", xyz, cond, nlayers=nlayers)
    return pred[..., :3], pred[..., -1:]


# ----------------------------------------------------------------------------
# grid + chunked evaluation   (utils/reconstruction.py:5-20, models/reconstruction.py:6-56)
# ----------------------------------------------------------------------------
def create_grid_points_from_bounds(minimun, maximum, res) -> np.ndarray:
    """float64 [res^3,3], 'ij' order: x slowest, z fastest (utils/reconstruction.py:10-18)."""
    x = np.linspace(minimun[0], maximum[0], res)
    y = np.linspace(minimun[1], maximum[1], res)
    z = np.linspace(minimun[2], maximum[2], res)
    X, Y, Z = np.meshgrid(x, y, z, indexing="ij")
    return np.column_stack((X.reshape(-1), Y.reshape(-1), Z.reshape(-1)))


def get_logits(forward, encoding: np.ndarray, grid_points: np.ndarray, nbatch_points: int = 100000):
    """models/reconstruction.py:6-25.  ``forward(points[1,n,3], lat[1,n,L]) -> (sdf, anchors)``;
    encoding 1-D [L] or [1,1,L]; each chunk is a separate forward() call (so the
    eval-mode last-point overwrite happens per chunk)."""
    enc = np.asarray(encoding, F32).reshape(1, 1, -1)
    out = []
    n = grid_points.shape[1]
    for s in range(0, n, nbatch_points):
        pts = grid_points[:, s:s + nbatch_points]
        sdf, _ = forward(pts, np.repeat(enc, pts.shape[1], axis=1))
        out.append(np.asarray(sdf, F32).reshape(-1))
    return np.concatenate(out)


def get_logits_backward(forward_shape, forward_expr, enc_shape, enc_expr, grid_points, nbatch_points=100000):
    """models/reconstruction.py:28-56: x_can = x + F_ex(x, z_ex); F_id(x_can, z_id)."""
    enc_s = np.asarray(enc_shape, F32).reshape(1, 1, -1)
    out = []
    n = grid_points.shape[1]
    for s in range(0, n, nbatch_points):
        pts = grid_points[:, s:s + nbatch_points]
        if enc_expr is not None:
            enc_e = np.asarray(enc_expr, F32).reshape(1, 1, -1)
            off, _ = forward_expr(pts, np.repeat(enc_e, pts.shape[1], axis=1))
            pts_can = (pts + off).astype(F32)
        else:
            pts_can = pts
        sdf, _ = forward_shape(pts_can, np.repeat(enc_s, pts.shape[1], axis=1))
        out.append(np.asarray(sdf, F32).reshape(-1))
    return np.concatenate(out)


def hack_indices(n_points: int, nbatch_points: int) -> np.ndarray:
    """Flat indices whose value is overwritten by the eval-mode hack when a volume is
    extracted with get_logits(chunk=nbatch_points): last point of every chunk."""
    idx = list(range(nbatch_points - 1, n_points, nbatch_points))
    if not idx or idx[-1] != n_points - 1:
        idx.append(n_points - 1)
    return np.asarray(idx, np.int64)


def flops_per_point(kind: str) -> int:
    """Dense algorithmic FLOPs/point used for roofline.achieved."""
    return {"nphm_identity": 9_616_000, "deformation": 2_624_512, "npm": 14_682_112}[kind]
