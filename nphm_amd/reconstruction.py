"""Dense-grid / chunked field evaluation with the reference's call signatures
(src/NPHM/models/reconstruction.py:6-88, src/NPHM/utils/reconstruction.py:5-20), plus the fused
grid evaluator and its multi-GPU (x-slab + RCCL all-gather) form.

``get_logits`` returns exactly what the reference returns — a float32 numpy array of
``decoder(points, lat)`` evaluated chunk by chunk, INCLUDING the eval-mode overwrite of the last
point of every chunk — but when the decoder is the HIP-backed NPHM identity field the whole
volume is produced by one kernel launch with no per-chunk latent repeat and no per-chunk
device->host copy.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from .deepsdf import DeepSDF, DeformationNetwork
from .ensembled_deepsdf import FastEnsembleDeepSDFMirrored


# ----------------------------------------------------------------------------------------------
# grids
# ----------------------------------------------------------------------------------------------
def create_grid_points_from_bounds(minimun, maximum, res, scale=None):
    """float64 [res^3, 3] lattice in 'ij' order — x slowest, z fastest
    (utils/reconstruction.py:5-20)."""
    if scale is not None:
        res = int(scale * res)
        minimun = scale * minimun
        maximum = scale * maximum
    axes = [np.linspace(minimun[d], maximum[d], res) for d in range(3)]
    gx, gy, gz = np.meshgrid(*axes, indexing="ij")
    return np.stack([gx.ravel(), gy.ravel(), gz.ravel()], axis=1)


def grid_axes(minimun, maximum, res):
    """The three fp32 axis vectors of the lattice above, cast exactly like the reference casts the
    full lattice (float64 linspace -> float32, fitting_pointclouds.py:168-169)."""
    if np.isscalar(res):
        res = (res, res, res)
    return tuple(np.linspace(minimun[d], maximum[d], res[d]).astype(np.float32) for d in range(3))


def _as_lat_row(encoding: torch.Tensor, lat_dim: int) -> torch.Tensor:
    enc = encoding.reshape(-1, encoding.shape[-1])
    if enc.shape[0] != 1 or enc.shape[1] != lat_dim:
        raise ValueError(f"expected a single latent code of width {lat_dim}, got {tuple(encoding.shape)}")
    return enc


def _detect_lattice(points: torch.Tensor):
    """If ``points`` [1,n,3] is a full 'ij' lattice (any per-axis resolutions) return its axis
    tensors, else None.  One fused compare on the device + one host sync."""
    if points.dim() != 3 or points.shape[0] != 1 or points.shape[2] != 3 or points.shape[1] < 8:
        return None
    p = points[0]
    n = p.shape[0]
    head = p[: min(n, 1 << 16)]
    # z is fastest: rz = first index where z wraps back to its first value
    z0 = head[0, 2]
    wrap = (head[1:, 2] == z0).nonzero()
    if wrap.numel() == 0:
        return None
    rz = int(wrap[0, 0]) + 1
    if n % rz:
        return None
    y0 = p[0, 1]
    ycol = p[::rz, 1]
    wrap = (ycol[1:] == y0).nonzero()
    ry = int(wrap[0, 0]) + 1 if wrap.numel() else ycol.shape[0]
    if n % (rz * ry):
        return None
    rx = n // (rz * ry)
    ax = p[:: rz * ry, 0].contiguous()
    ay = p[: rz * ry: rz, 1].contiguous()
    az = p[:rz, 2].contiguous()
    g = p.view(rx, ry, rz, 3)
    ok = (g[..., 0] == ax[:, None, None]).all() & (g[..., 1] == ay[None, :, None]).all() & \
         (g[..., 2] == az[None, None, :]).all()
    return (ax, ay, az) if bool(ok) else None


def marching_cubes(volume: np.ndarray, isovalue: float = 0.0, *, negate: bool = False, n_threads: int = 0):
    """Iso-surface of a host fp32 volume [nx,ny,nz] -> (vertices float64 [nv,3] in index space,
    triangles int64 [nf,3]); same contract as the third-party ``mcubes.marching_cubes`` the
    reference calls (utils/reconstruction.py:30).  Native, slab-parallel over the host cores."""
    import ctypes
    lib = _lib.load()
    vol = np.ascontiguousarray(volume, dtype=np.float32)
    if vol.ndim != 3:
        raise ValueError("marching_cubes expects a 3-D volume")
    handle, nv, nf = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64()
    rc = lib.nphm_mc_extract(vol.ctypes.data, vol.shape[0], vol.shape[1], vol.shape[2], float(isovalue),
                             int(bool(negate)), int(n_threads), ctypes.byref(handle), ctypes.byref(nv),
                             ctypes.byref(nf))
    if rc != 0:
        raise _lib.NphmAmdError(f"nphm_mc_extract failed ({rc})")
    try:
        verts = np.empty((nv.value, 3), np.float64)
        faces = np.empty((nf.value, 3), np.int64)
        lib.nphm_mc_fetch(handle, verts.ctypes.data, faces.ctypes.data)
    finally:
        lib.nphm_mc_free(handle)
    return verts, faces


def marching_cubes_device(volume: torch.Tensor, isovalue: float = 0.0, *, negate: bool = False):
    """The same extraction for a DEVICE-resident fp32 volume [nx,ny,nz], on the GPU (count, hipCUB
    scan, emit): (vertices float64 [nv,3], triangles int64 [nf,3]) as device tensors, bit-identical to
    ``marching_cubes`` of the same volume.  One host sync (the mesh size)."""
    import ctypes
    lib = _lib.load()
    if not volume.is_cuda or volume.dtype != torch.float32 or volume.dim() != 3:
        raise ValueError("marching_cubes_device expects a 3-D fp32 tensor on a ROCm device")
    vol = volume.contiguous()
    nx, ny, nz = vol.shape
    dev = vol.device
    ws = torch.empty(lib.nphm_mc_device_workspace_bytes(nx, ny, nz), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    nv, nf = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(lib.nphm_mc_device_count(vol.data_ptr(), nx, ny, nz, float(isovalue), int(bool(negate)), ws.data_ptr(),
                                        ctypes.byref(nv), ctypes.byref(nf), stream), "nphm_mc_device_count")
    verts = torch.empty(nv.value, 3, dtype=torch.float64, device=dev)
    faces = torch.empty(nf.value, 3, dtype=torch.int64, device=dev)
    _lib.check(lib.nphm_mc_device_emit(vol.data_ptr(), nx, ny, nz, float(isovalue), int(bool(negate)), ws.data_ptr(),
                                       verts.data_ptr() if nv.value else None, faces.data_ptr() if nf.value else None,
                                       stream), "nphm_mc_device_emit")
    return verts, faces


def to_host(t: torch.Tensor) -> np.ndarray:
    """Device tensor -> numpy through page-locked memory of torch's caching host allocator: one DMA at
    link speed and no first-touch page faults (``tensor.cpu()`` lands in fresh pageable memory: 1-3 ms for
    the 18 MB of a 256^3 mesh, with 25 ms outliers).  The array keeps the pinned block alive."""
    if not t.is_cuda:
        return t.numpy()
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return host.numpy()


def extract_mesh(decoder, encoding, mini, maxi, resolution, nbatch_points=25000):
    """latent -> mesh without leaving the device until the mesh exists: the NPHM identity SDF on the
    reference lattice (= get_logits, incl. the eval-mode chunk overwrite) followed by the GPU marching
    cubes (= mesh_from_logits); only vertices / triangles travel to the host.  Same mesh as
    ``mesh_from_logits(get_logits(decoder, encoding, grid, nbatch_points), mini, maxi, resolution)``."""
    axes = grid_axes(mini, maxi, resolution)
    hack = 0 if decoder.training else int(nbatch_points)
    vol = evaluate_grid(decoder, encoding, axes, hack_chunk=hack)
    verts, faces = marching_cubes_device(vol.view(resolution, resolution, resolution), 0.0, negate=True)
    step = (np.array(maxi) - np.array(mini)) / (resolution - 1)
    vertices = to_host(verts) * np.expand_dims(step, axis=0)
    vertices += [mini[0], mini[1], mini[2]]
    triangles = to_host(faces)
    try:
        import trimesh
        return trimesh.Trimesh(vertices, triangles)
    except ImportError:
        return SimpleNamespace(vertices=vertices, faces=triangles)


def mesh_from_logits(logits, mini, maxi, resolution, n_threads: int = 0):
    """utils/reconstruction.py:22-37: SDF volume -> mesh in world coordinates.  Like the reference
    it negates ``logits`` IN PLACE (``logits *= -1`` on a reshape view) and extracts the zero level
    set; returns a ``trimesh.Trimesh`` when trimesh is installed, else a namespace with
    ``vertices`` / ``faces``."""
    logits = np.reshape(logits, (resolution,) * 3)
    logits *= -1
    if torch.cuda.is_available() and logits.dtype == np.float32 and logits.size >= 64 ** 3:
        # a ROCm device is there: upload (4 B/voxel) + GPU extraction beats the host pass ~8x at 256^3;
        # the two extractors return bit-identical meshes
        v, f = marching_cubes_device(torch.from_numpy(np.ascontiguousarray(logits)).cuda(), 0.0)
        vertices, triangles = to_host(v), to_host(f)
    else:
        vertices, triangles = marching_cubes(logits, 0.0, n_threads=n_threads)
    step = (np.array(maxi) - np.array(mini)) / (resolution - 1)
    vertices = vertices * np.expand_dims(step, axis=0)
    vertices += [mini[0], mini[1], mini[2]]
    try:
        import trimesh
        return trimesh.Trimesh(vertices, triangles)
    except ImportError:
        return SimpleNamespace(vertices=vertices, faces=triangles)


# ----------------------------------------------------------------------------------------------
# fused grid evaluation (NPHM identity field)
# ----------------------------------------------------------------------------------------------
def _hip_ready(decoder, device) -> bool:
    return (isinstance(decoder, FastEnsembleDeepSDFMirrored) and decoder.backend == "hip"
            and device.type == "cuda" and decoder.hip_supported())


def grid_workspace(device, n_x_local: int, ry: int, rz: int) -> torch.Tensor:
    """Scratch buffer of the binned grid traversal (nphm_identity_grid_workspace_bytes: ~9.5 B per voxel),
    allocated per call from torch's caching allocator: stream-ordered like every other temporary, returned
    to the cache when the caller drops it (``torch.cuda.empty_cache()`` can reclaim it), nothing is kept
    alive by this module."""
    need = int(_lib.load().nphm_identity_grid_workspace_bytes(int(n_x_local), int(ry), int(rz)))
    if need == 0:
        raise _lib.NphmAmdError("grid too large for one launch")
    return torch.empty(need, dtype=torch.uint8, device=device)


def _identity_state(decoder, lat, n_points, numerics):
    """(packed, state, anchors) of an inference launch: the decoder's own decision for ``n_points`` points, or - ``numerics``
    = ((prune_tol, precision code), bounds) - a decision taken elsewhere (rank 0's, ``shared_identity_numerics``)."""
    if numerics is None:
        return decoder.prepare_latent(lat, inference=True, n_points=n_points)
    knobs, bounds = numerics
    packed, state, anchors = decoder.prepare_latent(lat, inference=False, bounds=bounds)
    state.nphm_knobs = (float(knobs[0]), int(knobs[1]))
    return packed, state, anchors


def evaluate_grid(decoder: FastEnsembleDeepSDFMirrored, encoding: torch.Tensor, axes: Sequence,
                  *, hack_chunk: Optional[int] = None, x_range=None, x_planes=None,
                  out: Optional[torch.Tensor] = None, return_anchors: bool = False, stats=None,
                  binned: bool = True, numerics=None):
    """SDF of the NPHM identity field on the 'ij' lattice spanned by ``axes`` (three fp32 vectors),
    restricted to the x-planes ``x_range = (ix0, ix1)`` or to an ascending list ``x_planes``;
    returns a device tensor [n_planes*ry*rz] in the flattened order of the reference lattice.

    hack_chunk: chunk length whose last point get_logits would overwrite in eval mode
    (None -> off when decoder.training else whole volume as one chunk; 0 -> off).
    binned: traverse the lattice tile by tile in the order of the tiles' active-member sets (a scratch
    buffer per device; bitwise the same values as the brick-order traversal, faster).
    numerics: ((prune_tol, precision code), member bounds) decided elsewhere - the sharded evaluation passes rank 0's.
    """
    lib = _lib.load()
    device = encoding.device
    if not _hip_ready(decoder, device):
        raise _lib.NphmAmdError("evaluate_grid needs the HIP-backed NPHM identity field on a ROCm device")
    ax, ay, az = [torch.as_tensor(a, dtype=torch.float32, device=device).contiguous() for a in axes]
    rx, ry, rz = ax.numel(), ay.numel(), az.numel()
    ix0, ix1 = (0, rx) if x_range is None else x_range
    if hack_chunk is None:
        hack_chunk = 0 if decoder.training else rx * ry * rz
    lat = _as_lat_row(encoding.to(device=device, dtype=torch.float32), decoder.lat_dim)
    planes_dev = None
    if x_planes is not None:
        planes_dev = torch.as_tensor(np.ascontiguousarray(x_planes, dtype=np.int32)).to(device) \
            if not torch.is_tensor(x_planes) else x_planes.to(device=device, dtype=torch.int32).contiguous()
        n_planes = planes_dev.numel()
    else:
        n_planes = ix1 - ix0
    n = n_planes * ry * rz
    # the knobs are decided for the WHOLE lattice (rx * ry * rz points), not for this rank's share of it: every rank of a
    # sharded extraction must run the same setting (the shards of a volume are slices of one evaluation)
    packed, state, anchors = _identity_state(decoder, lat, rx * ry * rz, numerics)
    if n == 0:                                  # a rank without planes (more ranks than brick slabs)
        empty = torch.empty(0, dtype=torch.float32, device=device)
        return (empty, anchors) if return_anchors else empty
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=device)
    elif out.numel() != n or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("out must be a contiguous fp32 tensor with n_planes*ry*rz elements")
    stream = torch.cuda.current_stream(device).cuda_stream
    stats_ptr = None if stats is None else stats.data_ptr()
    ws = grid_workspace(device, n_planes, ry, rz) if binned else None
    ws_ptr, ws_bytes = (ws.data_ptr(), ws.numel()) if binned else (None, 0)
    if planes_dev is not None:
        _lib.check(lib.nphm_identity_eval_grid_planes(
            packed.data_ptr(), state.data_ptr(), ax.data_ptr(), ay.data_ptr(), az.data_ptr(), rx, ry, rz,
            planes_dev.data_ptr(), n_planes, int(hack_chunk), *state.nphm_knobs,
            out.data_ptr(), stats_ptr, ws_ptr, ws_bytes, stream), "nphm_identity_eval_grid_planes")
    else:
        _lib.check(lib.nphm_identity_eval_grid(
            packed.data_ptr(), state.data_ptr(), ax.data_ptr(), ay.data_ptr(), az.data_ptr(), rx, ry, rz,
            ix0, ix1, int(hack_chunk), *state.nphm_knobs,
            out.data_ptr(), stats_ptr, ws_ptr, ws_bytes, stream), "nphm_identity_eval_grid")
    return (out, anchors) if return_anchors else out


def _mlp_hip_ready(decoder, device) -> bool:
    return (isinstance(decoder, DeepSDF) and decoder.backend == "hip" and device.type == "cuda"
            and decoder.hip_supported())


def _mlp_lattice_code(mlp, packed, state, ax, ay, az) -> int:
    """`numerics` argument of a lattice evaluation of the dense MLP kernel, decided for the WHOLE lattice."""
    rx, ry, rz = ax.numel(), ay.numel(), az.numel()
    device = ax.device

    def sample():
        # 16 x 16 x 16 lattice points spread over the WHOLE lattice: what the two-term layers are verified on
        # (DeepSDF._numerics_code).  The decision is taken for the lattice, not for this slab of it: every rank of a sharded
        # evaluation, and a slab evaluated on its own, runs the setting of the full volume (slabs are exact slices of it)
        pick = lambda a, hi: a[torch.linspace(0, hi - 1, min(16, hi), device=device).round().long()]
        sx, sy, sz = pick(ax, rx), pick(ay, ry), pick(az, rz)
        return torch.stack(torch.meshgrid(sx, sy, sz, indexing="ij"), dim=-1).reshape(1, -1, 3).contiguous()

    return int(mlp._numerics_code(packed, state, rx * ry * rz, sample))


def evaluate_grid_mlp(mlp: DeepSDF, cond_row: torch.Tensor, axes: Sequence, *, x_range=None,
                      add_input: bool = False, out: Optional[torch.Tensor] = None, code: Optional[int] = None):
    """A DeepSDF skip-MLP (NPM SDF / deformation backbone) on the x-slab ``x_range`` of the 'ij'
    lattice spanned by ``axes``: device tensor [(ix1-ix0)*ry*rz, out_dim] in flattened lattice
    order, one fused launch.  ``cond_row`` [1, lat_dim] is the conditioning vector;
    ``add_input`` adds the lattice coordinates to the first three outputs.  ``code``: the kernel's `numerics` argument
    decided elsewhere (rank 0's per-layer tiers in a sharded evaluation) instead of this module's own calibration."""
    lib = _lib.load()
    device = cond_row.device
    if not _mlp_hip_ready(mlp, device):
        raise _lib.NphmAmdError("evaluate_grid_mlp needs a HIP-covered DeepSDF on a ROCm device")
    ax, ay, az = [torch.as_tensor(a, dtype=torch.float32, device=device).contiguous() for a in axes]
    rx, ry, rz = ax.numel(), ay.numel(), az.numel()
    ix0, ix1 = (0, rx) if x_range is None else x_range
    packed, state = mlp.prepare_latent(_as_lat_row(cond_row.to(device=device, dtype=torch.float32), mlp.lat_dim))
    n = (ix1 - ix0) * ry * rz
    if out is None:
        out = torch.empty(n, mlp.n_out, dtype=torch.float32, device=device)
    elif out.numel() != n * mlp.n_out or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("out must be a contiguous fp32 tensor with (ix1-ix0)*ry*rz*out_dim elements")
    stream = torch.cuda.current_stream(device).cuda_stream
    if code is None:
        code = _mlp_lattice_code(mlp, packed, state, ax, ay, az)
    ws = mlp.eval_workspace(int(code), device)
    _lib.check(lib.nphm_mlp_eval_grid_ws(*mlp._arch(), packed.data_ptr(), state.data_ptr(), ax.data_ptr(),
                                         ay.data_ptr(), az.data_ptr(), rx, ry, rz, ix0, ix1, int(bool(add_input)), int(code),
                                         out.data_ptr(), ws.data_ptr() if ws is not None else None,
                                         ws.numel() if ws is not None else 0, stream), "nphm_mlp_eval_grid")
    return out


def _expr_condition(decoder_expr, encoding_expr, anchors, device):
    """Conditioning row [1, lat_dim] of the expression decoder for a row-constant latent."""
    enc = encoding_expr.reshape(1, 1, -1).to(device=device, dtype=torch.float32)
    if isinstance(decoder_expr, DeformationNetwork):
        dummy = torch.zeros(1, 1, 3, device=device)
        was_training = decoder_expr.training
        decoder_expr.eval()                      # no conditioning noise (deepSDF.py:220-221)
        try:
            with torch.no_grad():
                cond = decoder_expr._condition(dummy, enc, anchors)
        finally:
            decoder_expr.train(was_training)
        return decoder_expr.defDeepSDF, cond[:, 0, :]
    return decoder_expr, enc[:, 0, :]


def evaluate_grid_two_stage(decoder_shape: FastEnsembleDeepSDFMirrored, decoder_expr, encoding_shape,
                            encoding_expr, axes: Sequence, *, anchors=None, hack_chunk: Optional[int] = None,
                            x_range=None, out: Optional[torch.Tensor] = None, return_canonical: bool = False,
                            numerics=None, mlp_code: Optional[int] = None):
    """Two-stage lattice evaluation (get_logits_backward, models/reconstruction.py:28-56) entirely on
    the device: canonical points x + F_ex(x, z_ex) by the fused deformation kernel, then the identity
    field at those points by the fused ensemble kernel (same brick traversal as evaluate_grid).
    ``anchors`` [1,39,3] conditions an NPHM 'compress' deformation net (default: the identity
    net's predicted anchors); ``encoding_expr`` is the expression decoder's full latent
    ([z_id | z_ex] for DeformationNetwork)."""
    lib = _lib.load()
    device = encoding_shape.device
    if not _hip_ready(decoder_shape, device):
        raise _lib.NphmAmdError("evaluate_grid_two_stage needs the HIP-backed NPHM identity field on a ROCm device")
    ax, ay, az = [torch.as_tensor(a, dtype=torch.float32, device=device).contiguous() for a in axes]
    rx, ry, rz = ax.numel(), ay.numel(), az.numel()
    ix0, ix1 = (0, rx) if x_range is None else x_range
    if hack_chunk is None:
        hack_chunk = 0 if decoder_shape.training else rx * ry * rz
    lat = _as_lat_row(encoding_shape.to(device=device, dtype=torch.float32), decoder_shape.lat_dim)
    packed, state, anchors_pred = _identity_state(decoder_shape, lat, rx * ry * rz, numerics)
    if anchors is None:
        anchors = anchors_pred
    mlp, cond = _expr_condition(decoder_expr, encoding_expr, anchors, device)
    canonical = evaluate_grid_mlp(mlp, cond, (ax, ay, az), x_range=(ix0, ix1), add_input=True, code=mlp_code)
    if canonical.shape[1] != 3:
        canonical = canonical[:, :3].contiguous()
    n = (ix1 - ix0) * ry * rz
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=device)
    elif out.numel() != n or out.dtype != torch.float32 or not out.is_contiguous() or out.device != canonical.device:
        raise ValueError("out must be a contiguous fp32 device tensor with (ix1-ix0)*ry*rz elements")
    stream = torch.cuda.current_stream(device).cuda_stream
    ws = grid_workspace(device, ix1 - ix0, ry, rz)
    _lib.check(lib.nphm_identity_eval_grid_points(
        packed.data_ptr(), state.data_ptr(), canonical.data_ptr(), rx, ry, rz, ix0, ix1, int(hack_chunk),
        *state.nphm_knobs, out.data_ptr(), None,
        ws.data_ptr(), ws.numel(), stream), "nphm_identity_eval_grid_points")
    return (out, canonical) if return_canonical else out


def slab_bounds(rx: int, world_size: int, rank: int):
    """Contiguous x-slab of rank ``rank``: ceil(rx/world) planes each, last ranks may be short."""
    per = (rx + world_size - 1) // world_size
    return min(rank * per, rx), min((rank + 1) * per, rx)


def cyclic_planes(rx: int, world_size: int, rank: int, unit: int = 8) -> np.ndarray:
    """x-planes of rank ``rank`` in the work-balancing multi-GPU partition: the lattice is cut into
    brick slabs of ``unit`` planes (the kernel's brick depth) and rank r takes slabs r, r + world,
    r + 2 world, ...  The ensemble is spatially sparse — planes through the face cost ~2x the border
    planes — so contiguous equal slabs leave the middle ranks 35 % slower than the mean at 8 GPUs
    (measured); the cyclic assignment samples the whole volume on every rank (max/mean 1.02)."""
    planes = [np.arange(u * unit, min((u + 1) * unit, rx)) for u in range(rank, (rx + unit - 1) // unit, world_size)]
    return np.concatenate(planes).astype(np.int32) if planes else np.zeros(0, np.int32)


def gather_planes(local: torch.Tensor, rx: int, plane: int, group=None, unit: int = 8) -> torch.Tensor:
    """Reassemble the volume from the ranks' cyclic plane sets: ONE all_gather_into_tensor (RCCL
    over xGMI on ROCm) of shards padded to the largest plane count, then one index_select that
    restores the flattened 'ij' order.  ``local``: this rank's planes [n_planes*plane] in
    ``cyclic_planes`` order, or already the padded shard [shard_depth*plane] (what
    ``evaluate_grid_sharded`` hands over: the kernel wrote straight into it, no staging copy).
    Returns the full volume [rx*plane] on every rank."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    depth = shard_depth(rx, world, unit)
    if local.numel() == depth * plane and local.is_contiguous():
        shard = local.reshape(-1)
    else:
        shard = torch.empty(depth * plane, dtype=local.dtype, device=local.device)
        shard[: local.numel()] = local.reshape(-1)
        shard[local.numel():].zero_()
    gathered = torch.empty(world * depth * plane, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, shard, group=group)
    return reorder_gathered(gathered, rx, plane, world, unit)


def shard_depth(rx: int, world_size: int, unit: int = 8) -> int:
    """Planes per (padded) shard of the cyclic partition."""
    return max(len(cyclic_planes(rx, world_size, r, unit)) for r in range(world_size))


_REORDER_INDEX = {}


def reorder_index(rx: int, world_size: int, unit: int, device) -> torch.Tensor:
    """Row of the gathered [world * depth, plane] buffer that holds global x-plane i, as a device index tensor
    (cached per partition and device: built and uploaded once, not per step)."""
    key = (rx, world_size, unit, str(device))
    index = _REORDER_INDEX.get(key)
    if index is None:
        depth = shard_depth(rx, world_size, unit)
        src = np.empty(rx, np.int64)
        for r in range(world_size):
            p = cyclic_planes(rx, world_size, r, unit)
            src[p] = r * depth + np.arange(len(p))
        index = torch.from_numpy(src).to(device)
        _REORDER_INDEX[key] = index
    return index


def reorder_gathered(gathered: torch.Tensor, rx: int, plane: int, world_size: int, unit: int = 8) -> torch.Tensor:
    """[world * depth * plane] rank-major shards of the cyclic partition -> volume in 'ij' order."""
    depth = shard_depth(rx, world_size, unit)
    index = reorder_index(rx, world_size, unit, gathered.device)
    return gathered.view(world_size * depth, plane).index_select(0, index).reshape(-1)


def _takes_output(fn) -> bool:
    """does the injected per-rank evaluator have the ``(planes, out)`` form (>= 2 positional parameters)?"""
    import inspect
    try:
        params = list(inspect.signature(fn).parameters.values())
    except (TypeError, ValueError):
        return True
    if any(p.kind == p.VAR_POSITIONAL for p in params):
        return True
    return sum(p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD) for p in params) >= 2


# ---- numerics of a sharded evaluation: ONE rank decides, every rank runs that decision ------------------------------
# The knobs of numerics = "auto" come out of measurements (calibration of a weight version, verification of a latent,
# the per-layer tiers of the dense MLP): deterministic kernels on identical inputs, so ranks that calibrate on their own
# agree in practice - but nothing enforced it, and the gathered volume is bit-identical to the single-GPU one only if they
# do (verdict round 5).  Now rank ``src`` decides and broadcasts ~1.3 KB (knob pair, the [40][4] member bounds, the MLP's
# tier code); the other ranks never calibrate.  Every rank caches the decision per (weights, latent digest, lattice size)
# while ``src`` reports it stable (a calibration exists for these weights); a new calibration serial in any broadcast
# voids the cache.  All ranks make the same calls (SPMD), so they hit and miss together: a steady-state step sends nothing.
_N_SHARED = 8 + 160


def _broadcast_from(values, device, group, src):
    """float64 vector of ``_N_SHARED`` entries from rank ``src`` (``values`` is ignored elsewhere) -> numpy, on every rank"""
    import torch.distributed as dist
    on_device = dist.get_backend(group) == "nccl"
    buf = torch.zeros(_N_SHARED, dtype=torch.float64)
    if dist.get_rank(group) == src:
        buf[: len(values)] = torch.as_tensor(values, dtype=torch.float64)
    if on_device:
        buf = buf.to(device)
    dist.broadcast(buf, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    return buf.cpu().numpy()


def shared_numerics(decoder, lat, n_points, group=None, src=0, mlp=None, mlp_key=None, mlp_code_fn=None):
    """(((prune_tol, precision code), bounds), mlp_code) of rank ``src`` for an evaluation of ``n_points`` lattice points
    with the latent row ``lat`` [1, lat_dim], on every rank of ``group``.  ``mlp`` (a DeepSDF), ``mlp_key`` (what its
    conditioning depends on, hashable) and ``mlp_code_fn() -> int`` (called on ``src`` only, on a cache miss): also the dense
    MLP kernel's `numerics` code of the two-stage evaluation (None otherwise)."""
    import torch.distributed as dist
    device = lat.device
    me = dist.get_rank(group)
    key = (decoder._weights_key(device), decoder._latent_digest(lat.detach().reshape(-1, decoder.lat_dim)[:8]), int(n_points),
           None if mlp is None else (tuple((t.data_ptr(), t._version) for t in sum(mlp._lin_params(), [])),
                                     mlp.precision, float(mlp.numerics_target), mlp_key))
    cache = decoder.__dict__.setdefault("_shared_numerics", {"serial": None, "entries": {}})
    hit = cache["entries"].get(key)
    if hit is not None:
        return hit
    values = None
    if me == src:
        knobs, bounds, serial = decoder.inference_numerics(device, lat, n_points)
        code = -1 if mlp is None else int(mlp_code_fn())
        serial = serial * 65536 + (0 if mlp is None else getattr(mlp, "_numerics_serial", 0) % 65536)
        stable = float(serial != 0 or decoder.numerics != "auto")
        values = [float(serial), stable, float(knobs[0]), float(knobs[1]), float(bounds is not None), float(code), 0.0, 0.0]
        if bounds is not None:
            values += bounds.detach().reshape(-1).double().cpu().tolist()
    got = _broadcast_from(values, device, group, src)
    serial, stable, code = int(got[0]), bool(got[1]), int(got[5])
    if cache["serial"] != serial:
        cache["entries"].clear()
        cache["serial"] = serial
    bounds = torch.from_numpy(got[8:8 + 160].astype(np.float32)).reshape(40, 4).to(device) if got[4] else None
    decision = (((float(got[2]), int(got[3])), bounds), None if code < 0 else code)
    if stable:
        if len(cache["entries"]) >= 64:
            cache["entries"].pop(next(iter(cache["entries"])))
        cache["entries"][key] = decision
    return decision


def evaluate_grid_sharded(decoder, encoding, axes, *, hack_chunk: Optional[int] = None, group=None,
                          evaluate=None, unit: int = 8):
    """Multi-GPU lattice evaluation: every rank evaluates its cyclic set of x-planes
    (``cyclic_planes``; one kernel launch that writes straight into the rank's padded shard) and one
    all-gather + reorder reassembles the full volume on every rank (``gather_planes``).  The chunk
    overwrite uses global indices, so the result is bit-identical to the single-GPU volume.
    ``evaluate(planes: int32 ndarray, out: tensor)`` can be injected (CPU/gloo tests, other fields): it fills
    ``out`` [len(planes)*ry*rz] with the values of those planes.  The round-1 form ``evaluate(planes) -> tensor``
    is still accepted (its result is copied into the shard)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    rx, ry, rz = (len(a) for a in axes)
    plane = ry * rz
    planes = cyclic_planes(rx, world, rank, unit)
    if evaluate is None:
        if hack_chunk is None:
            hack_chunk = 0 if decoder.training else rx * ry * rz
        numerics = None
        if world > 1 and _hip_ready(decoder, encoding.device):
            # rank 0's knobs and member bounds (shared_numerics): no rank decides for itself
            lat = _as_lat_row(encoding.to(dtype=torch.float32), decoder.lat_dim)
            numerics = shared_numerics(decoder, lat, rx * ry * rz, group)[0]
        evaluate = lambda pl, out: evaluate_grid(decoder, encoding, axes, hack_chunk=hack_chunk, x_planes=pl, out=out,
                                                 numerics=numerics)
    shard = torch.empty(shard_depth(rx, world, unit) * plane, dtype=torch.float32, device=encoding.device)
    n = len(planes) * plane
    if n:
        res = evaluate(planes, shard[:n]) if _takes_output(evaluate) else evaluate(planes)
        if res is not None and torch.is_tensor(res) and res.data_ptr() != shard.data_ptr():
            shard[:n].copy_(res.reshape(-1))
    shard[n:].zero_()                           # padding of the ranks with fewer planes
    return gather_planes(shard, rx, plane, group, unit)


def evaluate_grid_two_stage_sharded(decoder_shape, decoder_expr, encoding_shape, encoding_expr, axes, *,
                                    anchors=None, hack_chunk: Optional[int] = None, group=None, unit: int = 8):
    """Multi-GPU form of ``evaluate_grid_two_stage`` (configs[2] sharded like configs[3]): the same cyclic
    partition; a rank runs the deformation + identity kernels once per contiguous ``unit``-plane slab of its
    set (256^3 on 8 ranks: 4 slabs), writing into its padded shard; one all-gather + reorder."""
    import torch.distributed as dist
    rx, ry, rz = (len(a) for a in axes)
    plane = ry * rz
    if hack_chunk is None:
        hack_chunk = 0 if decoder_shape.training else rx * ry * rz
    numerics, mlp_code = None, None
    device = encoding_shape.device
    if dist.get_world_size(group) > 1 and _hip_ready(decoder_shape, device):
        # rank 0 decides the identity field's knobs AND the deformation backbone's per-layer tiers
        lat = _as_lat_row(encoding_shape.to(device=device, dtype=torch.float32), decoder_shape.lat_dim)
        mlp = decoder_expr.defDeepSDF if isinstance(decoder_expr, DeformationNetwork) else decoder_expr

        def mlp_code():
            anch = anchors if anchors is not None else decoder_shape.prepare_latent(lat)[2]
            _, cond = _expr_condition(decoder_expr, encoding_expr, anch, device)
            packed, state = mlp.prepare_latent(_as_lat_row(cond.to(device=device, dtype=torch.float32), mlp.lat_dim))
            ax, ay, az = [torch.as_tensor(a, dtype=torch.float32, device=device).contiguous() for a in axes]
            return _mlp_lattice_code(mlp, packed, state, ax, ay, az)

        digest = lambda t: None if t is None else decoder_shape._latent_digest(t.detach().reshape(1, -1))
        numerics, mlp_code = shared_numerics(decoder_shape, lat, rx * ry * rz, group, mlp=mlp, mlp_code_fn=mlp_code,
                                             mlp_key=(digest(encoding_expr), digest(anchors), tuple(len(a) for a in axes)))

    def evaluate(planes, out):
        planes = np.asarray(planes)
        starts = np.flatnonzero(np.diff(planes, prepend=planes[0] - 2) != 1)      # first plane of every contiguous run
        for i, s in enumerate(starts):
            e = starts[i + 1] if i + 1 < len(starts) else len(planes)
            evaluate_grid_two_stage(decoder_shape, decoder_expr, encoding_shape, encoding_expr, axes, anchors=anchors,
                                    hack_chunk=hack_chunk, x_range=(int(planes[s]), int(planes[e - 1]) + 1),
                                    out=out[s * plane:e * plane], numerics=numerics, mlp_code=mlp_code)

    return evaluate_grid_sharded(decoder_shape, encoding_shape, axes, hack_chunk=hack_chunk, group=group,
                                 evaluate=evaluate, unit=unit)


# ----------------------------------------------------------------------------------------------
# reference-signature entry points
# ----------------------------------------------------------------------------------------------
def get_logits(decoder, encoding, grid_points, nbatch_points=100000, return_anchors=False):
    """models/reconstruction.py:6-25."""
    device = grid_points.device
    if _hip_ready(decoder, device) and grid_points.dtype == torch.float32 and grid_points.shape[0] == 1:
        lat = _as_lat_row(encoding, decoder.lat_dim)
        hack = 0 if decoder.training else int(nbatch_points)
        lattice = _detect_lattice(grid_points)
        if lattice is not None:
            vol, anchors = evaluate_grid(decoder, lat, lattice, hack_chunk=hack, return_anchors=True)
        else:
            lib = _lib.load()
            packed, state, anchors = decoder.prepare_latent(lat.to(device), inference=True, n_points=grid_points.shape[1])
            pts = grid_points.contiguous()
            vol = torch.empty(pts.shape[1], dtype=torch.float32, device=device)
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(lib.nphm_identity_eval_points(
                packed.data_ptr(), state.data_ptr(), pts.data_ptr(), 1, pts.shape[1], hack,
                *state.nphm_knobs, vol.data_ptr(), None, stream),
                "nphm_identity_eval_points")
        logits = to_host(vol)
        return (logits, anchors) if return_anchors else logits

    if _mlp_hip_ready(decoder, device) and grid_points.dtype == torch.float32 and grid_points.shape[0] == 1:
        # NPM global SDF: one fused launch over the whole point set (no eval-mode overwrite in DeepSDF)
        lat = _as_lat_row(encoding.to(device), decoder.lat_dim)
        lattice = _detect_lattice(grid_points)
        if lattice is not None:
            vol = evaluate_grid_mlp(decoder, lat, lattice)
        else:
            vol = decoder.forward_hip(grid_points, lat)
        logits = to_host(vol.reshape(-1))
        return (logits, None) if return_anchors else logits

    # generic decoders: the reference's chunk loop (latent broadcast instead of repeat)
    enc = encoding.reshape(1, 1, -1)
    chunks = []
    anchors = None
    for points in torch.split(grid_points, nbatch_points, dim=1):
        with torch.no_grad():
            logits, anchors = decoder(points, enc.expand(1, points.shape[1], -1), None)
            chunks.append(logits.reshape(-1).detach().cpu())
    logits = torch.cat(chunks, dim=0).numpy()
    return (logits, anchors) if return_anchors else logits


def get_logits_backward(decoder_shape, decoder_expr, encoding_shape, encoding_expr, grid_points,
                        nbatch_points=100000, return_anchors=False, anchors=None):
    """models/reconstruction.py:28-56: canonicalise with the deformation field, then query the
    identity field (two-stage evaluation).  ``anchors`` (extension, default None = the reference's
    call) conditions an NPHM 'compress' deformation net, which the reference's own signature cannot
    serve (it passes None, :44)."""
    device = grid_points.device
    expr_hip = encoding_expr is None or _mlp_hip_ready(
        decoder_expr.defDeepSDF if isinstance(decoder_expr, DeformationNetwork) else decoder_expr, device)
    expr_needs_anchors = isinstance(decoder_expr, DeformationNetwork) and decoder_expr.mode in (
        "compress", "interpolate", "GNN")
    if (_hip_ready(decoder_shape, device) and expr_hip and grid_points.dtype == torch.float32
            and grid_points.shape[0] == 1 and not (expr_needs_anchors and anchors is None)
            and not (isinstance(decoder_expr, DeformationNetwork) and decoder_expr.mode == "interpolate")):
        lib = _lib.load()
        lat = _as_lat_row(encoding_shape.to(device), decoder_shape.lat_dim)
        hack = 0 if decoder_shape.training else int(nbatch_points)
        lattice = _detect_lattice(grid_points)
        if encoding_expr is None:
            return get_logits(decoder_shape, encoding_shape, grid_points, nbatch_points, return_anchors)
        if lattice is not None:
            vol = evaluate_grid_two_stage(decoder_shape, decoder_expr, lat, encoding_expr, lattice,
                                          anchors=anchors, hack_chunk=hack)
            anchors_pred = decoder_shape.prepare_latent(lat)[2] if return_anchors else None
        else:
            mlp, cond = _expr_condition(decoder_expr, encoding_expr, anchors, device)
            canonical = mlp.forward_hip(grid_points, cond, add_input=True)[..., :3].contiguous()
            packed, state, anchors_pred = decoder_shape.prepare_latent(lat, inference=True, n_points=canonical.shape[1])
            vol = torch.empty(canonical.shape[1], dtype=torch.float32, device=device)
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(lib.nphm_identity_eval_points(
                packed.data_ptr(), state.data_ptr(), canonical.data_ptr(), 1, canonical.shape[1], hack,
                *state.nphm_knobs, vol.data_ptr(), None, stream),
                "nphm_identity_eval_points")
        logits = to_host(vol)
        return (logits, anchors_pred) if return_anchors else logits

    enc_s = encoding_shape.reshape(1, 1, -1)
    chunks = []
    anchors_out = None
    for points in torch.split(grid_points, nbatch_points, dim=1):
        with torch.no_grad():
            if encoding_expr is not None:
                enc_e = encoding_expr.reshape(1, 1, -1)
                offsets, _ = decoder_expr(points, enc_e.expand(1, points.shape[1], -1), anchors)
                points_can = points + offsets
            else:
                points_can = points
            logits, anchors_out = decoder_shape(points_can, enc_s.expand(1, points.shape[1], -1), None)
            chunks.append(logits.reshape(-1).detach().cpu())
    logits = torch.cat(chunks, dim=0).numpy()
    return (logits, anchors_out) if return_anchors else logits


def deform_mesh(mesh, deformer, lat_rep, anchors, lat_rep_shape=None):
    """models/reconstruction.py:59-88: displace mesh vertices by the deformation field."""
    verts = torch.from_numpy(np.asarray(mesh.vertices)).float().unsqueeze(0).to(lat_rep.device)
    cond = lat_rep if lat_rep_shape is None else torch.cat([lat_rep_shape, lat_rep], dim=-1)
    with torch.no_grad():
        if isinstance(deformer, DeformationNetwork):
            # one fused launch for all vertices (HIP tier; composite otherwise)
            posed = deformer.canonical_points(verts, cond, anchors).squeeze(0).cpu().numpy()
        else:
            parts = []
            for pts in torch.split(verts, 1 << 16, dim=1):
                d, _ = deformer(pts, cond, anchors)
                parts.append(d)
            posed = (verts[:, :, :3] + torch.cat(parts, dim=1)).squeeze(0).cpu().numpy()
    try:
        import trimesh
        return trimesh.Trimesh(posed, mesh.faces, process=False)
    except ImportError:
        return SimpleNamespace(vertices=posed, faces=np.asarray(mesh.faces))
