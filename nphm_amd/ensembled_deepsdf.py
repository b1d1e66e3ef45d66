"""Host-side mirror of the reference's NPHM identity field
(src/NPHM/models/EnsembledDeepSDF.py): same class names, constructor signatures, attributes,
``forward`` contracts and ``state_dict`` layout, so checkpoints load with ``strict=True`` and
the reference's callers (get_logits, fitting loop, trainers) run unchanged.

Execution tiers of ``FastEnsembleDeepSDFMirrored.forward``:

* **HIP** (``libnphm_amd.so``, gfx950): used whenever no autograd graph is needed, the tensors
  live on a ROCm device, the architecture is the NPHM one (nphm.yaml) and the latent is constant
  along the point axis (what get_logits / the fitting loop pass).  One fused kernel evaluates the
  40-member ensemble and the Gaussian blend; if the library is missing this tier raises.
* **HIP autograd** (first order): when a graph IS needed but no parameter requires grad (latent
  fitting: only xyz / the latent are optimised), in training mode: forward = the same fused kernel,
  backward = ``nphm_identity_backward`` (member-centric MFMA kernel) for d/dxyz, d/danchors and
  d/d(folded biases), chained through ``mlp_pos`` and the latent columns by ordinary autograd.  Not
  double-differentiable (training gets the HIP training tier below: its parameters require grad).
* **HIP training** (twice differentiable in xyz, every parameter trainable): when parameters require grad,
  in training mode (``compute_loss``: decoder -> gradient(pred, x, create_graph=True) -> loss.backward()).  The
  member MLPs and their first / second-order backward run on ``nphm_identity_train_forward/backward``
  (ident_train_kernel.hip), the weight gradients on ``nphm_identity_train_weight_grads`` over operands the backward
  kernel stores; ``module.train_backend = "composite"`` (or NPHM_AMD_TRAIN_TIER=composite) opts out.
* **composite**: a differentiable PyTorch formulation (latent columns of lin0 / the skip layer are
  applied once per latent instead of once per point).  Used for per-point latents, other architectures,
  eval mode under autograd and by explicit opt-in (first- and second-order autograd).
  It is never chosen silently on a CPU tensor: that needs ``module.backend = "composite"``.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib

_CAL_SERIAL = __import__("itertools").count(1)      # numbers the calibrations of a process (inference_numerics)

_SQRT2 = float(np.sqrt(2))


def _member_sets(ensemble_size: int, n_symm: int) -> torch.Tensor:
    """member k -> index of its weight set: the first n_symm sets serve two members each
    (reference: repeat_interleave(2) + cat, EnsembledDeepSDF.py:43-45)."""
    idx = [k // 2 if k < 2 * n_symm else n_symm + (k - 2 * n_symm) for k in range(ensemble_size)]
    return torch.tensor(idx, dtype=torch.long)


class EnsembledLinear(nn.Module):
    """``ensemble_size`` linear layers evaluated at once; the first ``n_symm`` parameter sets are
    shared by two (mirror-symmetric) members each.  Parameters: ``weight``
    [ensemble_size - n_symm, out, in], ``bias`` [ensemble_size - n_symm, out]
    (EnsembledDeepSDF.py:8-55)."""

    def __init__(self, ensemble_size, n_symm, in_features, out_features, bias=True):
        super().__init__()
        self.ensemble_size = ensemble_size
        self.n_symm = n_symm
        self.in_features = in_features
        self.out_features = out_features
        n_sets = ensemble_size - n_symm
        self.weight = nn.Parameter(torch.empty(n_sets, out_features, in_features))
        if bias:
            self.bias = nn.Parameter(torch.empty(n_sets, out_features))
        else:
            self.register_parameter("bias", None)
        self.register_buffer("_sets", _member_sets(ensemble_size, n_symm), persistent=False)
        self.reset_parameters()

    def reset_parameters(self):
        # every parameter set gets nn.Linear's default init on its own (same RNG consumption
        # order as the reference, so a seeded construction yields identical weights)
        bound_b = 1.0 / math.sqrt(self.in_features) if self.in_features > 0 else 0.0
        with torch.no_grad():
            for s in range(self.weight.shape[0]):
                nn.init.kaiming_uniform_(self.weight[s], a=math.sqrt(5))
                if self.bias is not None:
                    nn.init.uniform_(self.bias[s], -bound_b, bound_b)

    def member_weight(self):
        return self.weight.index_select(0, self._sets)

    def member_bias(self):
        return None if self.bias is None else self.bias.index_select(0, self._sets)

    def forward(self, input):
        # input [A, P, in] -> [A, P, out]
        w = self.member_weight()
        if self.bias is None:
            return torch.bmm(input, w.transpose(1, 2))
        return torch.baddbmm(self.member_bias().unsqueeze(1), input, w.transpose(1, 2))


class EnsembledDeepSDF(nn.Module):
    """A bank of DeepSDF MLPs with a skip connection into layer ``nlayers//2``
    (EnsembledDeepSDF.py:58-126)."""

    def __init__(self, ensemble_size, n_symm, lat_dim, hidden_dim, nlayers, out_dim=1, input_dim=3):
        super().__init__()
        d_in = input_dim + lat_dim
        self.ensemble_size = ensemble_size
        self.n_symm = n_symm
        self.lat_dim = lat_dim
        self.input_dim = input_dim
        dims = [d_in] + [hidden_dim] * nlayers + [out_dim]
        self.num_layers = len(dims)
        self.skip_in = [nlayers // 2]
        for layer in range(self.num_layers - 1):
            fan_out = dims[layer + 1] - d_in if (layer + 1) in self.skip_in else dims[layer + 1]
            setattr(self, f"lin{layer}", EnsembledLinear(ensemble_size, n_symm, dims[layer], fan_out))
        self.activation = nn.Softplus(beta=100)

    def _lin(self, i) -> EnsembledLinear:
        return getattr(self, f"lin{i}")

    def forward(self, xyz, lat_rep):
        # xyz [A,B,N,3], lat_rep [A,B,N,F] -> [A,B,N,out]
        return self.evaluate(xyz, lat_rep)

    def evaluate(self, coords, cond):
        """coords [A,B,N,D]; cond [A,B,Lr,F] with Lr in {1, N}.  The latent columns of the first
        and of the skip layer are multiplied once per latent row and broadcast over the points."""
        A, B, N, D = coords.shape
        Lr = cond.shape[2]
        last = self.num_layers - 2
        x = None
        for layer in range(self.num_layers - 1):
            lin = self._lin(layer)
            W = lin.member_weight()                       # [A,out,in]
            b = lin.member_bias()                         # [A,out]
            if layer == 0 or layer in self.skip_in:
                n_prev = 0 if layer == 0 else x.shape[-1]
                scale = 1.0 if layer == 0 else 1.0 / _SQRT2
                Wc = W[:, :, n_prev:n_prev + D]
                Wl = W[:, :, n_prev + D:]
                lat_term = torch.einsum("ablf,aof->ablo", cond, Wl) * scale + b[:, None, None, :]
                y = torch.einsum("abnd,aod->abno", coords, Wc) * scale
                if layer != 0:
                    y = y + torch.einsum("abnk,aok->abno", x, W[:, :, :n_prev]) * scale
                x = y + (lat_term if Lr == N else lat_term.expand(A, B, N, -1))
            else:
                x = torch.einsum("abnk,aok->abno", x, W) + b[:, None, None, :]
            if layer < last:
                x = self.activation(x)
        return x


def sample_point_feature(q, p, fea, var=0.1 ** 2, background=False):
    """Gaussian-of-distance blend of per-anchor features (EnsembledDeepSDF.py:129-150).
    q [B,N,3], p [B,K,3], fea [B,N,K(+1),C] -> [B,N,C]."""
    d = (p[:, None, :, :] - q[:, :, None, :]).norm(dim=3)
    logit = -((d + 10e-6) ** 2)
    if background:
        logit = torch.cat([logit, torch.full_like(logit[:, :, :1], -0.2)], dim=-1)
    w = (logit / var).exp()
    w = w / (w.sum(dim=2, keepdim=True) + 1e-6)
    return (w.unsqueeze(-1) * fea).sum(dim=2)


def _blend_mask(anchors, xyz, prune_tol, n_members):
    """Normalised blend weights [B,N,A] (EnsembledDeepSDF.py:129-150) and the members the fused kernel's pruning rule
    keeps (per point the smallest weights are dropped while they sum to <= A * prune_tol; all if negative)."""
    A = n_members
    d = (anchors[:, None, :, :] - xyz[:, :, None, :]).norm(dim=3) + 1e-5
    w = torch.exp(-(d * d) / 0.01)
    w_bg = float(np.exp(-20.0))
    denom = w.sum(dim=2, keepdim=True) + w_bg + 1e-6
    what = torch.cat([w, torch.full_like(w[:, :, :1], w_bg)], dim=2) / denom
    if prune_tol >= 0:
        tol = float(prune_tol)
        cut = torch.full_like(what[..., :1], tol)
        for mult in (2.0, 4.0, 8.0, 16.0, 40.0):
            below = (what * (what <= mult * tol)).sum(dim=2, keepdim=True)
            cut = torch.where(below <= A * tol, torch.full_like(cut, mult * tol), cut)
        mask = what > cut
    else:
        mask = torch.ones_like(what, dtype=torch.bool)
    return what, mask


def _member_point_lists(anchors, xyz, prune_tol, n_members):
    """(row, member) -> points that keep the member under the pruning rule of the fused kernel (per
    point the smallest normalised blend weights are dropped while they sum to <= 40 * prune_tol; all
    points if negative): blend weights [B,N,A] (zero where pruned), tile table int32 [T,4] = (row, member,
    offset into the point list, count <= 64) and the point list int32 [P] (sorted by row, member,
    point).  One host sync (the per-pair counts size the launch)."""
    B, N, _ = xyz.shape
    A = n_members
    what, mask = _blend_mask(anchors, xyz, prune_tol, A)
    idx = mask.permute(0, 2, 1).nonzero()                      # sorted by (row, member, point)
    counts = torch.bincount(idx[:, 0] * A + idx[:, 1], minlength=B * A).cpu().numpy()
    offs = np.concatenate([[0], np.cumsum(counts)])
    n_t = (counts + 63) // 64                                  # 64-point tiles per (row, member)
    bk = np.repeat(np.arange(B * A), n_t)
    within = np.arange(int(n_t.sum())) - np.repeat(np.cumsum(n_t) - n_t, n_t)
    tiles = np.stack([bk // A, bk % A, offs[bk] + 64 * within, np.minimum(64, counts[bk] - 64 * within)],
                     axis=1).astype(np.int32)
    return what * mask, torch.from_numpy(tiles).to(xyz.device), idx[:, 2].to(torch.int32).contiguous()


_PINNED_COUNTS = {}           # (kind, device index, stream id) -> [pinned int32 tensor, event recorded behind its last asynchronous use]


def _pinned_staging(kind, dev, n):
    """A pinned int32 staging buffer of >= n elements for (device, current stream), kept across calls (pinning is a driver
    call).  The buffer may still be the source / target of the asynchronous copy of the previous call on this stream's key:
    its event is waited for before the buffer is handed out again; another device, stream or thread gets another buffer
    (advisor, round 5: one process-global buffer was safe only for one stream)."""
    import threading
    key = (kind, dev.index, torch.cuda.current_stream(dev).cuda_stream, threading.get_ident())
    slot = _PINNED_COUNTS.get(key)
    if slot is None or slot[0].numel() < n:
        if slot is not None and slot[1] is not None:
            slot[1].synchronize()
        slot = _PINNED_COUNTS[key] = [torch.empty(max(n, 1 << 10) * (2 if kind == "tables" else 1), dtype=torch.int32).pin_memory(), None]
    elif slot[1] is not None:
        slot[1].synchronize()
    return slot


def _train_member_lists(mask, sets):
    """Point lists of the training kernels (ident_train_kernel.hip) from ``mask`` [B,N,A] (the members the pruning
    rule keeps per point: a bool mask, or the blend weights, > 0 where kept): tiles ordered by (member, row), so the tiles of one weight set are contiguous (``sets``
    [A] = member -> set, non-decreasing).  Returns the forward
    kernel's tile table (64-point tiles), the backward kernel's int32 [T,4] = (row, member, offset, count <= 32)
    over the same point list, the point list int32 and the work list of the backward pass:
    the tile table is cut into pieces of <= _TRAIN_RING_TILES tiles (the backward kernel's stored operands of one
    piece live in a ring buffer that fits the Infinity Cache), every piece into chunks of <= _WGRAD_CHUNK tiles of
    ONE weight set for the weight-gradient kernel: ``pieces`` = list of (first tile, tiles, first chunk, chunks),
    chunk table int32 [C,4] = (weight set, first tile RELATIVE to its piece, tiles, piece), and the tables of
    nphm_identity_train_reduce_grads: int32 [sets + 1 | A * B + 1] = first chunk of every weight set (the chunk table is ordered
    by tile, hence by set) | first backward tile of every (member, row) pair, and the tiles per piece.  One host sync."""
    B, N, A = mask.shape
    lib = _lib.load() if mask.is_cuda else None
    plist = None
    if mask.is_cuda and mask.dtype == torch.float32 and mask.is_contiguous() and A == 40:
        # ``mask`` = the blend weights themselves (> 0 where listed): counts and point list by two launches of our own - the host
        # waits for the 40 B counts only, the list is written while it builds the tables (torch.nonzero + bincount: ~25
        # launches and two synchronisations)
        dev = mask.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        counts_dev = torch.empty(A * B, dtype=torch.int32, device=dev)
        _lib.check(lib.nphm_identity_train_pair_counts(mask.data_ptr(), B, N, counts_dev.data_ptr(), stream), "nphm_identity_train_pair_counts")
        c_slot = _pinned_staging("counts", dev, A * B)
        counts_host = c_slot[0][:A * B]
        counts_host.copy_(counts_dev, non_blocking=True)
        ready = torch.cuda.Event()
        ready.record()
        c_slot[1] = ready
        plist = torch.empty(B * N * A, dtype=torch.int32, device=dev)
        _lib.check(lib.nphm_identity_train_point_list(mask.data_ptr(), B, N, counts_dev.data_ptr(), plist.data_ptr(), stream),
                   "nphm_identity_train_point_list")
        ready.synchronize()                                    # the one sync: the counts size five launches
        counts = counts_host.numpy().astype(np.int64)
    else:
        with torch.no_grad():
            idx = (mask if mask.dtype == torch.bool else mask > 0).permute(2, 0, 1).nonzero()        # sorted by (member, row, point)
            counts = torch.bincount(idx[:, 0] * B + idx[:, 1], minlength=A * B).cpu().numpy().astype(np.int64)   # the one sync
    sets_np = _sets_on_host(sets)
    n_sets = int(sets_np.max()) + 1
    # the tables on the host in C (nphm_identity_train_tables; numpy took 0.5 ms per step with the GPU idle), in ONE buffer
    # that travels to the device in one copy: [forward tiles | backward tiles | chunks | first chunk per set | first tile per pair]
    lib = _lib.load()
    total = int(counts.sum())
    cap64, cap32 = total // 64 + A * B, total // 32 + A * B
    o_fwd, o_bwd, o_ch = 0, 4 * cap64, 4 * cap64 + 4 * cap32
    o_set = o_ch + 4 * cap32
    o_pair = o_set + n_sets + 1
    need = o_pair + A * B + 1
    pinned = None
    if mask.is_cuda:
        # (a pinned staging buffer, kept per device / stream / thread: the tables travel by an asynchronous copy - a pageable
        # source is a blocking, staged one; _pinned_staging waits for the previous copy out of it)
        t_slot = _pinned_staging("tables", mask.device, max(need, 1 << 16))
        pinned = t_slot[0]
        buf = pinned.numpy()[:need]
    else:
        buf = np.empty(need, np.int32)
    sizes = np.zeros(4, np.int32)
    base = buf.ctypes.data
    _lib.check(lib.nphm_identity_train_tables(counts.ctypes.data, B, sets_np.ctypes.data, n_sets, int(_TRAIN_RING_TILES), int(_WGRAD_CHUNK),
                                              base + 4 * o_fwd, base + 4 * o_bwd, base + 4 * o_ch, base + 4 * o_set, base + 4 * o_pair,
                                              sizes.ctypes.data), "nphm_identity_train_tables")
    t64, T, C, ring = (int(v) for v in sizes)
    piece_of_chunk = buf[o_ch:o_ch + 4 * C].reshape(C, 4)[:, 3]
    pieces = []
    if C:
        c_first = np.searchsorted(piece_of_chunk, np.arange(int(piece_of_chunk[-1]) + 1), side="left")
        c_count = np.diff(np.r_[c_first, C])
        pieces = [(int(pi * ring), int(min(ring, T - pi * ring)), int(c0), int(nc)) for pi, (c0, nc) in enumerate(zip(c_first, c_count))]
    dev = mask.device
    if pinned is not None:
        d = torch.empty(need, dtype=torch.int32, device=dev)
        d.copy_(pinned[:need], non_blocking=True)
        t_slot[1] = torch.cuda.Event()
        t_slot[1].record()                                     # the buffer is free again when this copy has been consumed
    else:
        d = torch.from_numpy(buf).to(dev)
    tiles_fwd = d[o_fwd:o_fwd + 4 * t64].view(t64, 4)
    tiles = d[o_bwd:o_bwd + 4 * T].view(T, 4)
    chunks = d[o_ch:o_ch + 4 * C].view(C, 4)
    edge_tabs = d[o_set:]                                      # [n_sets + 1 | A * B + 1], contiguous
    if plist is None:
        plist = idx[:, 2].to(torch.int32).contiguous()
    return tiles_fwd, tiles, plist, chunks, pieces, (edge_tabs, n_sets, ring)


_SETS_HOST = {}


def _sets_on_host(sets):
    """member -> weight set as a contiguous int32 host array (a constant of the architecture: read from the device once)"""
    key = (sets.data_ptr(), sets._version, sets.numel())
    hit = _SETS_HOST.get(key)
    if hit is None:
        if len(_SETS_HOST) > 8:
            _SETS_HOST.clear()
        # (the entry holds the tensor: an address that is still referenced cannot be handed to another decoder's table)
        hit = _SETS_HOST[key] = (sets, np.ascontiguousarray(sets.detach().cpu().numpy().astype(np.int32)))
    return hit[1]


def _member_point_lists_device(state, xyz, prune_tol, n_members, stream):
    """The same lists built on the device with fixed capacity (nphm_identity_build_lists): no host sync, static
    shapes - what the autograd tier uses (the fitting step can be captured in a hipGraph).  Returns
    (blend weights [B,N,A] zero where pruned, tile table int32 [B*A*T,4] with the used tiles at the front, their
    number as a device int32 [1], point list)."""
    lib = _lib.load()
    B, N, _ = xyz.shape
    T = int(lib.nphm_identity_list_tiles(N))
    dev = xyz.device
    what = torch.empty(B, N, n_members, dtype=torch.float32, device=dev)
    tiles = torch.empty(2 * B * n_members * T, 4, dtype=torch.int32, device=dev)     # used tiles first | slot scratch
    n_used = torch.empty(1, dtype=torch.int32, device=dev)
    plist = torch.empty(B * n_members * T * 64, dtype=torch.int32, device=dev)
    _lib.check(lib.nphm_identity_build_lists(state.data_ptr(), xyz.data_ptr(), B, N, float(prune_tol), what.data_ptr(),
                                             tiles.data_ptr(), n_used.data_ptr(), plist.data_ptr(), stream),
               "nphm_identity_build_lists")
    return what, tiles[: B * n_members * T], n_used, plist


def _blend_weights_device(state, xyz, prune_tol, n_members, stream):
    """Normalised blend weights [B,N,A], zero where the pruning rule drops the member (the first launch of
    nphm_identity_build_lists alone: the training tier cuts its own member-ordered lists from them)."""
    lib = _lib.load()
    B, N, _ = xyz.shape
    what = torch.empty(B, N, n_members, dtype=torch.float32, device=xyz.device)
    _lib.check(lib.nphm_identity_build_lists(state.data_ptr(), xyz.data_ptr(), B, N, float(prune_tol), what.data_ptr(),
                                             None, None, None, stream), "nphm_identity_build_lists")
    return what


class _FrozenHeadFn(torch.autograd.Function):
    """y = head(x) for a small nn.Sequential of Linear / ReLU layers whose parameters are CONSTANTS (latent fitting:
    ``mlp_pos``, the deformation field's compressor): the whole chain in one launch, the gradient w.r.t. x in one launch
    (``nphm_head_forward / _backward``) - the PyTorch formulation is a GEMM + bias + ReLU launch per layer and direction."""

    @staticmethod
    def forward(ctx, x, n_layers, y_add, twice, *wb):
        """x [rows, >= in_features]: the head reads the first in_features columns of every row (a latent row goes in whole -
        no slice, whose backward would be a zero-fill plus a copy); y_add [out] or None is added to every output row.
        ``twice``: -> (y, an alias of y) for an output with two uses - their gradients meet in the backward launch instead of
        in an add launch of autograd's"""
        lib = _lib.load()
        ws, bs = list(wb[0::2]), list(wb[1::2])
        rows = x.shape[0]
        dims = [ws[0].shape[1]] + [w.shape[0] for w in ws]
        xc = x.detach().contiguous().float()
        y = torch.empty(rows, dims[-1], dtype=torch.float32, device=x.device)
        hidden = torch.empty(rows, max(1, sum(dims[1:-1])), dtype=torch.float32, device=x.device)
        import ctypes
        cdims = (ctypes.c_int * 4)(*(dims + [0] * (4 - len(dims))))
        pad = lambda ts: _lib.ptr_array3([t.detach() for t in ts] + [ts[0].detach()] * (3 - len(ts)))
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(lib.nphm_head_forward(pad(ws), pad(bs), cdims, n_layers, xc.data_ptr(), xc.shape[1],
                                         None if y_add is None else y_add.data_ptr(), rows, y.data_ptr(), hidden.data_ptr(), stream),
                   "nphm_head_forward")
        ctx.save_for_backward(hidden, *[t.detach() for t in wb])
        ctx.meta = (n_layers, dims, xc.shape[1])
        if twice:
            ctx.set_materialize_grads(False)
            return y, y.detach()
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_y, g_y2=None):
        lib = _lib.load()
        hidden, *wb = ctx.saved_tensors
        n_layers, dims, width = ctx.meta
        ws, bs = list(wb[0::2]), list(wb[1::2])
        if g_y is None:
            g_y, g_y2 = g_y2, None
        if g_y is None:
            return (None,) * (4 + len(wb))
        g = g_y.contiguous().float()
        g2 = None if g_y2 is None else g_y2.contiguous().float()
        rows = g.shape[0]
        g_x = torch.empty(rows, width, dtype=torch.float32, device=g.device)        # written in full: zeros beyond the input columns
        import ctypes
        cdims = (ctypes.c_int * 4)(*(dims + [0] * (4 - len(dims))))
        pad = lambda ts: _lib.ptr_array3(list(ts) + [ts[0]] * (3 - len(ts)))
        stream = torch.cuda.current_stream(g.device).cuda_stream
        _lib.check(lib.nphm_head_backward(pad(ws), pad(bs), cdims, n_layers, hidden.data_ptr(), g.data_ptr(),
                                          None if g2 is None else g2.data_ptr(), rows, g_x.data_ptr(), width, stream), "nphm_head_backward")
        return (g_x, None, None, None) + (None,) * len(wb)


def _row0(lat_rep):
    """lat_rep[:, 0, :]; for a [B,1,L] code as a reshape - the backward of a select is a zero-fill plus a copy (two launches on
    the fitting step's serial chain), the backward of a view is nothing"""
    return lat_rep.reshape(lat_rep.shape[0], lat_rep.shape[2]) if lat_rep.shape[1] == 1 else lat_rep[:, 0, :]


def frozen_head(seq, x, frozen: bool, add=None, twice=False):
    """``seq(x[:, :in_features]) (+ add)`` for an nn.Sequential of Linear (+ ReLU between) layers; on a ROCm device, with
    parameters that do not require grad (or ``frozen``), fp32 rows and <= 3 linear layers of width <= 1536 through the fused
    kernels.  ``x`` may be wider than the head's input (a whole latent row); ``add`` [out_features] is a constant.
    ``twice``: -> (y, y2) with y2 an alias of y for its second use (fused path; else the same tensor twice)."""
    lins = [m for m in seq if isinstance(m, nn.Linear)]
    others = [m for m in seq if not isinstance(m, nn.Linear)]
    ok = (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and 1 <= len(lins) <= 3 and len(others) == len(lins) - 1
          and x.shape[1] >= lins[0].in_features
          and all(isinstance(m, nn.ReLU) for m in others)
          and all(max(l.in_features, l.out_features) <= 1536 and l.bias is not None for l in lins)
          and (frozen or not any(p.requires_grad for l in lins for p in l.parameters()))
          and os.environ.get("NPHM_AMD_FIT_FUSED", "1") not in ("0", ""))
    if not ok:
        y = seq(x[:, :lins[0].in_features] if x.shape[1] != lins[0].in_features else x)
        y = y if add is None else y + add.reshape(1, -1).to(y)
        return (y, y) if twice else y
    wb = [t for l in lins for t in (l.weight, l.bias)]
    if add is not None:
        add = add.detach().reshape(-1).to(device=x.device, dtype=torch.float32).contiguous()
    return _FrozenHeadFn.apply(x, len(lins), add, bool(twice), *wb)


class _IdentityFieldFn(torch.autograd.Function):
    """sdf = field(xyz; latent rows, anchors) with hand-written forward and first-order backward kernels
    (member-centric: one workgroup = one member x 64 of the points that member matters for).  The kernels work on
    the HIP prologue's state (built from the latent rows); the backward kernel returns d/dxyz, d/danchors and the
    gradients of the folded biases of lin0 / the skip layer, which the latent blocks of those two layers map onto
    the latent rows right here (one batched matrix product) - the latent enters the field through nothing else;
    ``anchors`` is the differentiable ``mlp_pos`` output and only receives its gradient (chained by autograd).
    The decoder's parameters are constants on this tier."""

    @staticmethod
    def forward(ctx, module, xyz, lat_rows, anchors):
        lib = _lib.load()
        B, N, _ = xyz.shape
        dev = xyz.device
        A = module.num_kps + 1
        packed, state, _ = module.prepare_latent(lat_rows.detach(), anchors=anchors)     # the field of exactly these anchors
        xyz_c = xyz.detach().contiguous().float()
        stream = torch.cuda.current_stream(dev).cuda_stream
        what, tiles, n_used, plist = _member_point_lists_device(state, xyz_c, module._autograd_tol(packed, state, xyz_c, stream),
                                                                A, stream)
        fmem = torch.empty(B, N, A, dtype=torch.float32, device=dev)      # written for the listed (point, member) pairs only
        _lib.check(lib.nphm_identity_member_forward(
            packed.data_ptr(), module._packed_bwd(dev).data_ptr(), state.data_ptr(), xyz_c.data_ptr(), N,
            tiles.data_ptr(), tiles.shape[0], n_used.data_ptr(), plist.data_ptr(), fmem.data_ptr(), stream),
            "nphm_identity_member_forward")
        out = torch.empty(B, N, 1, dtype=torch.float32, device=dev)
        _lib.check(lib.nphm_identity_blend_members(what.data_ptr(), fmem.data_ptr(), B * N, out.data_ptr(), stream),
                   "nphm_identity_blend_members")                        # reads fmem only where the blend weight is not 0
        ctx.module = module
        ctx.save_for_backward(xyz_c, out, packed, state, tiles, n_used, plist, what)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        # first order only: differentiating these gradients again (eikonal / normal losses built with
        # create_graph=True) raises instead of silently dropping the second-order term; such losses run
        # on the composite tier (module.backend = "composite", or parameters that require grad)
        lib = _lib.load()
        module = ctx.module
        xyz, out, packed, state, tiles, n_used, plist, what = ctx.saved_tensors
        B, N, _ = xyz.shape
        dev = xyz.device
        A = module.num_kps + 1
        # all four are WRITTEN by the kernels (per-tile records added in fixed order: no zero fill, no atomics)
        H, K = module.hidden_dim, module.num_kps
        sizes = [B * N * 3, B * K * 3, B * A * H, B * A * H]
        parts = torch.empty(sum(sizes), dtype=torch.float32, device=dev).split(sizes)
        gx, ga = parts[0].view(B, N, 3), parts[1].view(B, K, 3)
        gb0, gb2 = parts[2].view(B, A, H), parts[3].view(B, A, H)
        g = grad_out.detach().reshape(B, N).contiguous().float()
        stream = torch.cuda.current_stream(dev).cuda_stream
        scratch = torch.empty(lib.nphm_identity_backward_scratch_bytes(B, N, tiles.shape[0]), dtype=torch.uint8, device=dev)
        _lib.check(lib.nphm_identity_backward(
            packed.data_ptr(), module._packed_bwd(dev).data_ptr(), state.data_ptr(), xyz.data_ptr(),
            out.data_ptr(), g.data_ptr(), B, N, tiles.data_ptr(), tiles.shape[0], n_used.data_ptr(), plist.data_ptr(),
            what.data_ptr(), scratch.data_ptr(), gx.data_ptr(), ga.data_ptr(), gb0.data_ptr(), gb2.data_ptr(), stream),
            "nphm_identity_backward")
        # folded bias of member k: W0[k][:, lat] cond_k + b (lin0), W2[k][:, lat] cond_k / sqrt2 + b (skip layer), with
        # cond_k = [z_glob | z_k]: d/dcond_k = gb0_k W0_lat[k] + gb2_k W2_lat[k] / sqrt2, then back onto the latent layout
        # (one launch through the latent columns of lin0 / lin2 in place of a batched GEMM over the 40 members + cat / sum)
        e = module.ensembled_deep_sdf
        w0, w2 = e.lin0.weight.detach(), e.lin2.weight.detach()
        g_lat = torch.empty(B, module.lat_dim, dtype=torch.float32, device=dev)
        scratch = torch.empty(lib.nphm_identity_latent_grad_scratch_bytes(B), dtype=torch.uint8, device=dev)
        _lib.check(lib.nphm_identity_latent_grad(w0.data_ptr(), w2.data_ptr(), gb0.data_ptr(), gb2.data_ptr(), B, g_lat.data_ptr(),
                                                 scratch.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                   "nphm_identity_latent_grad")
        return None, gx, g_lat, ga


_WGRAD_CHUNK = int(os.environ.get("NPHM_AMD_WGRAD_CHUNK", "32"))     # tiles per workgroup of the weight-gradient kernel
# tiles per piece of the backward pass (257 KiB of stored operands each; 0 = one piece): the reverse kernel writes the
# operands of a piece into a buffer the weight-gradient kernel consumes before the next piece reuses it - a cap on the
# memory of large batches (pieces of 512 tiles that would stay in the Infinity Cache measured 15-25 % SLOWER than
# one piece: launch gaps and tail effects outweigh the saved HBM traffic)
# (round 4: 65 536 tiles = 16 GiB of 257 KiB records - the nphm.yaml batch, 28.6 k tiles, is ONE piece: 9.64 -> 9.47 ms per step
# against two pieces of 16 384; the cap only matters for batches several times that size on a 288 GB device)
_TRAIN_RING_TILES = int(os.environ.get("NPHM_AMD_TRAIN_RING_TILES", "65536"))


class _MemberFieldFn(torch.autograd.Function):
    """Training tier, kernel half: (f_k, d f_k / d xyz) of the 40 member MLPs for every (point, member) the pruning
    rule keeps (zeros elsewhere), by ``nphm_identity_train_forward``; backward = ``nphm_identity_train_backward``
    (value + one tangent stream, reverse sweep with sigma'') followed by ``nphm_identity_train_weight_grads`` (the
    stored operands contracted over the point axis on the MFMA path).  Differentiable inputs: xyz, anchors, the folded biases of lin0 / the skip layer
    (``b0f``, ``b2f`` [B,40,200]: graph handles - the kernels read the same quantities from the HIP prologue's state;
    autograd chains them to the latent and to the latent columns of lin0 / lin2), the remaining weights and biases.

    Contract with ``_AttachGradientFn`` (always used together): the FIRST-order term  dL/df_k * d f_k/d xyz  of
    d/dxyz and d/danchors is returned
      * by ``_AttachGradientFn.backward`` as differentiable PyTorch ops when the backward pass records a graph
        (``gradient(pred, x)`` with create_graph=True) - this function then returns nothing (no parameter gradients
        are produced by a graph-recording backward pass);
      * by this function's kernel (which computes the total) otherwise (``loss.backward()``)."""

    @staticmethod
    def forward(ctx, module, xyz, anchors, lat_rows, b0f, b2f, W0, W1, W2, W3, W4, b1, b3, b4):
        lib = _lib.load()
        B, N, _ = xyz.shape
        dev = xyz.device
        A = module.num_kps + 1
        packed, state, anchors_k = module.prepare_latent(lat_rows.detach())
        packed_bwd = module._packed_bwd(dev)
        xyz_c = xyz.detach().contiguous().float()
        stream = torch.cuda.current_stream(dev).cuda_stream
        # members the pruning rule keeps per point: the list kernel's normalised blend weights (0 where pruned)
        tol = module._train_tol()
        what = _blend_weights_device(state, xyz_c, tol, A, stream)
        # (the zero fills in front of the list building: its wait for the counts then has them queued behind it)
        S = torch.zeros(B, N, A, dtype=torch.float32, device=dev)
        G = torch.zeros(B, N, A, 3, dtype=torch.float32, device=dev)
        tiles_fwd, tiles, plist, chunks, pieces, (edge_tabs, n_sets, ring) = _train_member_lists(what, module.ensembled_deep_sdf.lin0._sets)
        _lib.check(lib.nphm_identity_train_forward(
            packed.data_ptr(), packed_bwd.data_ptr(), state.data_ptr(), xyz_c.data_ptr(), N, tiles_fwd.data_ptr(),
            tiles_fwd.shape[0], plist.data_ptr(), S.data_ptr(), G.data_ptr(), stream), "nphm_identity_train_forward")
        if module.train_prune_tol is None:
            # size of the kept member values (their RMS; 0 marks a pruned pair): next step's pruning budget (_train_tol)
            module._train_mag = (torch.linalg.vector_norm(S.detach()), torch.count_nonzero(S.detach()))
        ctx.module = module
        ctx.pieces = pieces
        ctx.shapes = [t.shape for t in (W0, W1, W2, W3, W4, b1, b3, b4)]
        ctx.edge_meta = (n_sets, ring)
        ctx.save_for_backward(xyz_c, packed, packed_bwd, state, tiles, plist, chunks, edge_tabs)
        ctx.set_materialize_grads(False)
        return S, G

    @staticmethod
    def backward(ctx, gS, gG):
        n_in = 14
        if torch.is_grad_enabled():
            # graph-recording pass (gradient(pred, x), create_graph=True): _AttachGradientFn supplies d/dxyz
            # - and only d/dxyz, d/danchors.  A graph-recording pass that ALSO asks for latent or parameter gradients
            # (torch.autograd.grad(pred, [x, lat], create_graph=True), loss.backward(create_graph=True)) would have to
            # return them as differentiable ops, which this function cannot: refuse instead of returning nothing for them.
            # ctx is this function's node: next_functions[i] is where the gradient of input i would flow.
            # Where the engine can tell that such a gradient is wanted this raises; where it cannot (leaf inputs under
            # torch.autograd.grad) the gradient is returned NaN-filled: discarded by the engine when nobody asked for it,
            # unmistakable otherwise.
            # next_functions holds one edge per TENSOR input (13: the non-tensor `module` argument has none), whereas
            # needs_input_grad, `shapes` and the returned tuple are indexed by ARGUMENT position (14): edge e <-> argument e + 1.
            wanted, unknown = [], []
            assert len(ctx.next_functions) == n_in - 1, "one autograd edge per tensor argument of _MemberFieldFn.forward"
            for e, (node, _) in enumerate(ctx.next_functions):
                i = e + 1                                                          # argument index of this edge
                if i in (1, 2) or node is None or not ctx.needs_input_grad[i]:     # xyz, anchors: _AttachGradientFn
                    continue
                try:
                    if torch._C._will_engine_execute_node(node):
                        wanted.append(i)
                except Exception:                         # noqa: BLE001 - leaf under autograd.grad / API missing
                    unknown.append(i)
            if wanted:
                raise RuntimeError("nphm_amd training tier: a graph-recording backward pass (create_graph=True) can only "
                                   "deliver the spatial gradient d/dxyz (diff_operators.gradient); latent / parameter "
                                   "gradients of such a pass need module.train_backend = 'composite' "
                                   f"(requested inputs {wanted} of _MemberFieldFn)")
            if gG is not None:
                raise RuntimeError("nphm_amd training tier: third-order derivatives are not implemented "
                                   "(set module.train_backend = 'composite')")
            out = [None] * n_in
            if unknown:
                xyz_s = ctx.saved_tensors[0]
                B = xyz_s.shape[0]
                A, H = ctx.module.num_kps + 1, ctx.module.hidden_dim
                shapes = {3: (B, ctx.module.lat_dim), 4: (B, A, H), 5: (B, A, H)}
                shapes.update({6 + j: tuple(sh) for j, sh in enumerate(ctx.shapes)})
                for i in unknown:
                    out[i] = torch.full(shapes[i], float("nan"), dtype=torch.float32, device=xyz_s.device)
            return tuple(out)
        lib = _lib.load()
        module = ctx.module
        xyz, packed, packed_bwd, state, tiles, plist, chunks, edge_tabs = ctx.saved_tensors
        B, N, _ = xyz.shape
        dev = xyz.device
        A, H, K = module.num_kps + 1, module.hidden_dim, module.num_kps
        T = tiles.shape[0]
        # every gradient is accumulated into by the kernels: ONE zero-fill, views
        shapes = [torch.Size(sh) for sh in ([B, N, 3], [B, K, 3], [B, A, H], [B, A, H])] + list(ctx.shapes)
        sizes = [sh.numel() for sh in shapes]
        parts = [t.view(sh) for t, sh in zip(torch.zeros(sum(sizes), dtype=torch.float32, device=dev).split(sizes), shapes)]
        gx, ga, gb0, gb2, gW0, gW1, gW2, gW3, gW4, gb1, gb3, gb4 = parts
        if T and (gS is not None or gG is not None):
            gS_c = torch.zeros(B, N, A, dtype=torch.float32, device=dev) if gS is None else gS.detach().contiguous().float()
            gG_c = None if gG is None else gG.detach().contiguous().float()
            mode = module.train_operands
            if mode == "auto":            # binary16 storage where its rounding averages out (>= TRAIN_F16_MIN_POINTS points)
                mode = "f16" if B * N >= module.TRAIN_F16_MIN_POINTS else "f32"
            o16 = {"f32": 0, "bf16": 1, "f16": 2}[mode]
            scales = None
            if o16 == 2:
                # per-stream power-of-two scales of the binary16 operand storage, from the seeds of THIS step (one launch)
                sw = torch.zeros(8, dtype=torch.float32, device=dev)            # [4] scales | [3] work words
                _lib.check(lib.nphm_identity_train_operand_scales(
                    gS_c.data_ptr(), gS_c.numel(), None if gG_c is None else gG_c.data_ptr(), 0 if gG_c is None else gG_c.numel(),
                    sw.data_ptr() + 16, sw.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "nphm_identity_train_operand_scales")
                scales = sw
                shift = float(getattr(module, "train_scale_shift", 0.0))
                if shift:                 # extra octaves on S_d (and S_t = S_d / S_u): shrinks the margin - what the tests of the clamp detector use
                    f = 2.0 ** shift
                    sw[0:4].mul_(torch.tensor([f, 1.0, f, 1.0 / f], device=dev))
            # per tile: its share of the lin0 / lin4 gradients, summed over the tile's columns in the reverse kernel
            edge = torch.empty(lib.nphm_identity_train_edge_bytes(T), dtype=torch.uint8, device=dev)
            edge_tile = lib.nphm_identity_train_edge_bytes(1)
            stream = torch.cuda.current_stream(dev).cuda_stream
            gws = _lib.ptr_array5([gW0, gW1, gW2, gW3, gW4])
            timing = getattr(module, "_train_backward_events", None)       # bench.py: HIP events around the two kernels
            if timing is not None:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                ev[0].record()
            n_sets, ring = ctx.edge_meta
            C = chunks.shape[0]
            # every chunk's share of lin1 .. lin3 (written by the weight-gradient kernel, summed per weight set afterwards)
            wpart = torch.empty(lib.nphm_identity_train_wpart_bytes(C), dtype=torch.uint8, device=dev)
            wpart_chunk = lib.nphm_identity_train_wpart_bytes(1)

            def sweep(o16, scales):
                saved = torch.empty(lib.nphm_identity_train_saved_bytes(max(n for _, n, _, _ in ctx.pieces), o16),
                                    dtype=torch.uint8, device=dev)
                for t0, nt, c0, nc in ctx.pieces:
                    _lib.check(lib.nphm_identity_train_backward(
                        packed.data_ptr(), packed_bwd.data_ptr(), state.data_ptr(), xyz.data_ptr(), N,
                        tiles.data_ptr() + 16 * t0, nt, plist.data_ptr(), gS_c.data_ptr(),
                        None if gG_c is None else gG_c.data_ptr(), gx.data_ptr(), saved.data_ptr(),
                        edge.data_ptr() + edge_tile * t0, o16, None if scales is None else scales.data_ptr(), stream),
                        "nphm_identity_train_backward")
                    _lib.check(lib.nphm_identity_train_weight_grads(
                        saved.data_ptr(), o16, None if scales is None else scales.data_ptr(), chunks.data_ptr() + 16 * c0, nc,
                        wpart.data_ptr() + wpart_chunk * c0, stream), "nphm_identity_train_weight_grads")
                return saved

            saved = sweep(o16, scales)
            if o16 == 2:
                # Did a stored operand leave the binary16 range?  The scales follow the seeds' maxima through ratios measured on
                # two weight sets (operand_scales_kernel, 30 x margin); the kernel clamps silently and counts (ABI 11).  The
                # count is read - one synchronisation - on the first two binary16 steps of a module and on every
                # TRAIN_SAT_CHECK_EVERY-th after them: a step that clamped is REPEATED with fp32 storage and the module stays
                # there (train_operands = "f32"), with a warning.
                n16 = module.__dict__.get("_train_f16_steps", 0)
                module.__dict__["_train_f16_steps"] = n16 + 1
                if n16 < 2 or n16 % module.TRAIN_SAT_CHECK_EVERY == 0:
                    clamped = int(scales[7:8].view(torch.int32).item())
                    if clamped:
                        import warnings
                        warnings.warn(f"nphm_amd: {clamped} wavefronts of the training step's reverse sweep clamped a weight-gradient "
                                      "operand to the binary16 range (seed / adjoint ratios outside the measured ones): the step is "
                                      "repeated with fp32 operand storage and train_operands is set to 'f32' for this module")
                        module.train_operands = "f32"
                        module.__dict__["train_clamped_steps"] = module.__dict__.get("train_clamped_steps", 0) + 1
                        gx.zero_()                                   # (the only output the first sweep ACCUMULATED into)
                        o16, scales = 0, None
                        saved = sweep(0, None)
            if getattr(module, "_keep_train_operands", False):          # (tools/train_operand_stats.py: the last piece's stored operands)
                module._last_train_operands = saved
                module._last_train_seeds = (float(gS_c.abs().max()), 0.0 if gG_c is None else float(gG_c.abs().max()))
            scratch = torch.empty(C * edge_tile, dtype=torch.uint8, device=dev)
            _lib.check(lib.nphm_identity_train_reduce_grads(
                edge.data_ptr(), T, wpart.data_ptr(), chunks.data_ptr(), C, ring, edge_tabs.data_ptr(),
                edge_tabs.data_ptr() + 4 * (n_sets + 1), B, scratch.data_ptr(), gws, gb1.data_ptr(), gb3.data_ptr(), gb4.data_ptr(),
                gb0.data_ptr(), gb2.data_ptr(), ga.data_ptr(), stream), "nphm_identity_train_reduce_grads")
            if timing is not None:
                ev[1].record()
                timing.append((ev[0], ev[1], 2 * T * lib.nphm_identity_train_saved_bytes(1, o16)))   # bytes written + read back
        return (None, gx, ga, None, gb0, gb2, gW0, gW1, gW2, gW3, gW4, gb1, gb3, gb4)


class _AttachGradientFn(torch.autograd.Function):
    """Identity on the member values S that knows their spatial gradients G = dS/dxyz (an output of the same
    kernel): in a graph-recording backward pass it returns dL/dxyz and dL/danchors as PyTorch ops on G, which makes
    ``gradient(pred, x, create_graph=True)`` differentiable again (the second-order seed reaches ``_MemberFieldFn``
    through G); otherwise ``_MemberFieldFn``'s kernel returns these two."""

    @staticmethod
    def forward(ctx, xyz, anchors, S, G):
        ctx.save_for_backward(G)
        ctx.n_kps = anchors.shape[1]
        return S.clone()

    @staticmethod
    def backward(ctx, gS):
        if not torch.is_grad_enabled():
            return None, None, gS, None
        (G,) = ctx.saved_tensors
        gc = gS.unsqueeze(-1) * G                                    # [B,N,A,3]
        return gc.sum(dim=2), -gc[:, :, :ctx.n_kps].sum(dim=1), gS, None


class _BlendFn(torch.autograd.Function):
    """(pred, d pred / d xyz) of the Gaussian blend from the member values S and gradients G = dS/dxyz of
    ``_MemberFieldFn`` - one kernel forward, one backward (``nphm_identity_blend_forward/backward``).  With the spatial
    gradient an OUTPUT, a loss on it needs no graph-recording backward pass: ``loss.backward()`` reaches the member
    kernels with both seeds (dL/dS, dL/dG) through this function's first-order backward."""

    @staticmethod
    def forward(ctx, xyz, anchors, S, G):
        lib = _lib.load()
        B, N, _ = xyz.shape
        dev = xyz.device
        xyz_c, anchors_c = xyz.detach().contiguous().float(), anchors.detach().contiguous().float()
        S_c, G_c = S.detach().contiguous(), G.detach().contiguous()
        pred = torch.empty(B, N, dtype=torch.float32, device=dev)
        grad = torch.empty(B, N, 3, dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.nphm_identity_blend_forward(xyz_c.data_ptr(), anchors_c.data_ptr(), S_c.data_ptr(), G_c.data_ptr(), B, N,
                                                   pred.data_ptr(), grad.data_ptr(), stream), "nphm_identity_blend_forward")
        ctx.save_for_backward(xyz_c, anchors_c, S_c, G_c)
        ctx.set_materialize_grads(False)
        return pred, grad

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_pred, g_grad):
        lib = _lib.load()
        xyz, anchors, S, G = ctx.saved_tensors
        B, N, _ = xyz.shape
        dev = xyz.device
        gS, gG = torch.empty_like(S), torch.empty_like(G)
        sizes = [B * N * 3, anchors.numel()]
        gx, ga = torch.zeros(sum(sizes), dtype=torch.float32, device=dev).split(sizes)
        gp = torch.zeros(B, N, dtype=torch.float32, device=dev) if g_pred is None else g_pred.contiguous().float()
        gg = None if g_grad is None else g_grad.contiguous().float()
        stream = torch.cuda.current_stream(dev).cuda_stream
        part = torch.empty(lib.nphm_identity_blend_partial_bytes(B, N), dtype=torch.uint8, device=dev)
        _lib.check(lib.nphm_identity_blend_backward(
            xyz.data_ptr(), anchors.data_ptr(), S.data_ptr(), G.data_ptr(), gp.data_ptr(), None if gg is None else gg.data_ptr(),
            B, N, gS.data_ptr(), gG.data_ptr(), gx.data_ptr(), ga.data_ptr(), part.data_ptr(), stream), "nphm_identity_blend_backward")
        return gx.view(B, N, 3), ga.view_as(anchors), gS, gG


class FastEnsembleDeepSDFMirrored(nn.Module):
    """NPHM identity SDF: one small MLP per facial anchor (+ one background MLP), evaluated in
    anchor-local coordinates (odd member of each symmetric pair mirrored in x) and blended by a
    Gaussian of the anchor distance (EnsembledDeepSDF.py:153-267)."""

    def __init__(self, lat_dim_glob: int, lat_dim_loc: int, n_loc: int, n_symm_pairs: int,
                 anchors: torch.Tensor, hidden_dim: int, n_layers: int, pos_mlp_dim: int = 256,
                 out_dim: int = 1, input_dim: int = 3):
        super().__init__()
        self.lat_dim_glob = lat_dim_glob
        self.lat_dim_loc = lat_dim_loc
        self.lat_dim = lat_dim_glob + (n_loc + 1) * lat_dim_loc
        self.input_dim = input_dim
        self.out_dim = out_dim
        self.pos_mlp_dim = pos_mlp_dim
        self.num_kps = n_loc
        self.num_symm_pairs = n_symm_pairs
        self.hidden_dim = hidden_dim
        self.n_layers = n_layers

        self.ensembled_deep_sdf = EnsembledDeepSDF(ensemble_size=n_loc + 1, n_symm=n_symm_pairs,
                                                   lat_dim=lat_dim_glob + lat_dim_loc,
                                                   hidden_dim=hidden_dim, nlayers=n_layers,
                                                   out_dim=out_dim, input_dim=input_dim).float()
        # plain attribute like the reference: not a buffer, absent from state_dict
        self.anchors = anchors
        self.mlp_pos = nn.Sequential(
            nn.Linear(lat_dim_glob, pos_mlp_dim), nn.ReLU(),
            nn.Linear(pos_mlp_dim, pos_mlp_dim), nn.ReLU(),
            nn.Linear(pos_mlp_dim, n_loc * 3))

        # ---- execution knobs (defaults keep reference numerics within 1e-4) -------------------
        self.backend = "hip"            # "hip" | "composite"
        # Knobs of the inference kernels (eval_kernel.hip).  numerics = "auto" (default): pruning tolerance, precision
        # mode and tier thresholds are CALIBRATED per checkpoint (numerics.calibrate_numerics: the fastest setting whose
        # error against the dense exact-fp32 kernel stays a decade inside the 1e-4 bar for the weights at hand; once per
        # weight version, a few ms).  Assigning ``precision`` / ``prune_tol`` / a tier threshold (or NPHM_AMD_PRECISION /
        # NPHM_AMD_PRUNE_TOL) pins them: numerics = "fixed".
        # "bf16x3a2": per wavefront, single-pass bf16 for members below 1e-3 normalised blend weight, two passes (weights
        # rounded to bf16) below 1e-2, the full three-pass split-bf16 product from there on | "bf16x3a": the same without
        # the two-pass tier | "bf16x3": split-bf16 everywhere | "f16x3a2" / "f16x3": the same on binary16 halves (thresholds
        # 8e-3 / 8e-2) | "f32": exact products.  ``prune_tol`` is also the tolerance of the autograd / training tiers.
        self._prune_tol = float(os.environ.get("NPHM_AMD_PRUNE_TOL", "1e-7"))
        self._precision = os.environ.get("NPHM_AMD_PRECISION", "bf16x3a2")
        self._light_tol = None          # tier thresholds of the adaptive modes (None: the mode's default)
        self._mid_tol = None
        self._refine_band = None        # sign-safe refinement band (None / 0: off)
        pinned = "NPHM_AMD_PRECISION" in os.environ or "NPHM_AMD_PRUNE_TOL" in os.environ
        self.numerics = os.environ.get("NPHM_AMD_NUMERICS", "fixed" if pinned else "auto")
        self._calibration = None        # (weights key, preset dict) of numerics = "auto"
        self._verified_latents = {}     # digest of a latent -> sample error of the calibrated knobs on it (per-latent verification)
        # guard of pinned approximate modes: max |pinned - fp32-equivalent| allowed on a sample of the first call's latent
        # before the setting is tightened (numerics.clamp_pinned); None / 0 switches the check off
        self.pinned_guard = float(os.environ.get("NPHM_AMD_PINNED_GUARD", "2e-5")) or None
        self._pinned_check = None       # ((weights key, pinned knobs), clamp report)
        self._pack_cache = None         # (key, packed tensor)
        self._pack_bwd_cache = None     # (key, transposed pack for the backward kernel)
        # latent fitting with the REFERENCE's unchanged fitting.py: it leaves the decoder's parameters trainable and
        # lets autograd fill their .grad, which nothing reads.  True (or NPHM_AMD_ASSUME_FROZEN=1) treats the
        # parameters as frozen when choosing the tier, so that loop reaches the HIP autograd tier too (their .grad
        # stays None).  nphm_amd.fitting freezes them itself and does not need it.
        self.assume_frozen_parameters = os.environ.get("NPHM_AMD_ASSUME_FROZEN", "0") not in ("", "0")
        # tier of a training forward (parameters require grad, train mode): "hip" = member MLPs and their double
        # backward by the kernels of ident_train_kernel.hip | "composite" = PyTorch formulation
        self.train_backend = os.environ.get("NPHM_AMD_TRAIN_TIER", "hip")
        # pruning tolerance of the training tier (None: prune_tol).  Work is proportional to the kept (point, member)
        # pairs: 16.6 per point at 1e-7 on near-surface training samples
        self.train_prune_tol = (float(os.environ["NPHM_AMD_TRAIN_PRUNE_TOL"])
                                if "NPHM_AMD_TRAIN_PRUNE_TOL" in os.environ else None)
        # storage of the weight-gradient operands between the reverse and the weight-gradient kernel: "f16" (round 5:
        # binary16 with per-stream power-of-two scales derived from the step's seeds - half the traffic that bounds both
        # kernels, one-pass contraction) | "f32" (rounds 2-4: fp32-equivalent products end to end; 17 % slower) | "bf16"
        # (round 2: 8-bit mantissas) | "auto" (default): "f16" for batches of >= TRAIN_F16_MIN_POINTS points, "f32" below.
        # The storage rounding is random and averages over the columns of a weight set: parameter gradients against the
        # composite tier (largest entry per tensor) 5.2e-5 at 4 x 1000 points, ~1.5e-5 at the nphm.yaml batch (32 x 1693),
        # but 3e-4 on a 2 x 400-point batch, where fp32 storage gives 5e-6 .. 2e-5 and the traffic does not matter.
        # Values, spatial gradients and the lin0 / lin4 / bias gradients are the same bits in all modes.
        self.train_operands = os.environ.get("NPHM_AMD_TRAIN_OPERANDS", "auto")

    # ------------------------------------------------------------------------------------------
    def invalidate_pack(self):
        """Drop the cached MFMA-fragment copies of the weights.  The caches are keyed on the parameters'
        (address, version counter, device), which covers optimizer steps, ``load_state_dict`` and
        ``.to()``; writes that bypass the version counter (``param.data.copy_(...)``, raw pointer writes)
        need this call.  ``load_state_dict`` and ``_apply`` (``.to()/.float()/.cuda()``) call it themselves."""
        self._pack_cache = None
        self._pack_bwd_cache = None
        self._lat_blocks_cache = None

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_pack()
        # NPHM_AMD_VALIDATE=1: check the fast mode against the dense fp32 kernel once per loaded checkpoint
        # (numerics.validate_numerics, at the first HIP evaluation, with that call's latent)
        self._needs_validation = os.environ.get("NPHM_AMD_VALIDATE", "0") not in ("", "0")
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_pack()
        return out

    def hip_supported(self) -> bool:
        lib = _lib.load()
        return bool(lib.nphm_identity_supported(self.lat_dim_glob, self.lat_dim_loc, self.num_kps,
                                                self.num_symm_pairs, self.hidden_dim, self.n_layers,
                                                self.out_dim, self.input_dim)) and self.pos_mlp_dim <= 256

    def _lin_params(self):
        e = self.ensembled_deep_sdf
        ws = [e._lin(i).weight for i in range(5)]
        bs = [e._lin(i).bias for i in range(5)]
        return ws, bs

    def _packed(self, device):
        """Packed (MFMA-fragment order) copy of the ensemble weights on ``device``; rebuilt when
        any parameter changed (optimizer step, load_state_dict, .to())."""
        ws, bs = self._lin_params()
        key = tuple((t.data_ptr(), t._version, str(t.device)) for t in ws + bs) + (str(device),)
        if self._pack_cache is not None and self._pack_cache[0] == key:
            return self._pack_cache[1]
        lib = _lib.load()
        for t in ws + bs:
            if t.device != device or t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.NphmAmdError("ensemble parameters must be contiguous fp32 on the query device")
        packed = torch.empty(lib.nphm_identity_packed_bytes(), dtype=torch.uint8, device=device)
        stream = torch.cuda.current_stream(device).cuda_stream
        _lib.check(lib.nphm_identity_pack(_lib.ptr_array5(ws), _lib.ptr_array5(bs), packed.data_ptr(), stream),
                   "nphm_identity_pack")
        self._pack_cache = (key, packed)
        return packed

    def _latent_blocks(self, device):
        """[A, 2H, 96]: per member the latent columns of lin0 stacked on those of the skip layer (/ sqrt 2) - what maps
        the backward kernel's bias gradients onto the latent rows (cached like _packed)."""
        e = self.ensembled_deep_sdf
        ws = [e.lin0.weight, e.lin2.weight]
        key = tuple((t.data_ptr(), t._version, str(t.device)) for t in ws) + (str(device),)
        cache = getattr(self, "_lat_blocks_cache", None)
        if cache is not None and cache[0] == key:
            return cache[1]
        with torch.no_grad():
            d_in, n1 = self.input_dim, e.lin1.out_features
            blocks = torch.cat([e.lin0.member_weight()[:, :, d_in:], e.lin2.member_weight()[:, :, n1 + d_in:] / _SQRT2],
                               dim=1).to(device).contiguous()
        self._lat_blocks_cache = (key, blocks)
        return blocks

    def _packed_bwd(self, device):
        """Transposed split-bf16 pack of lin0..lin3 for nphm_identity_backward (cached like _packed)."""
        ws, _ = self._lin_params()
        key = tuple((t.data_ptr(), t._version, str(t.device)) for t in ws) + (str(device),)
        if self._pack_bwd_cache is not None and self._pack_bwd_cache[0] == key:
            return self._pack_bwd_cache[1]
        lib = _lib.load()
        packed = torch.empty(lib.nphm_identity_bwd_packed_bytes(), dtype=torch.uint8, device=device)
        stream = torch.cuda.current_stream(device).cuda_stream
        _lib.check(lib.nphm_identity_pack_bwd(_lib.ptr_array5(ws), packed.data_ptr(), stream), "nphm_identity_pack_bwd")
        self._pack_bwd_cache = (key, packed)
        return packed

    def prepare_latent(self, lat_rows: torch.Tensor, inference: bool = False, bounds="auto", anchors=None, n_points=None):
        """lat_rows [B, lat_dim] -> (packed weights, latent_state, anchors [B,n_loc,3]) via the HIP prologue kernel.
        ``inference``: the state feeds the inference kernels (eval_kernel.hip) - with numerics = "auto" the knobs are
        calibrated for the current weights first (once per weight version) and the fitted member magnitude bounds are
        installed into the state.  ``bounds``: "auto" (the calibrated bounds when they belong to the current weights,
        else the plain-weight rule), None, or a [40,4] device tensor.  ``anchors`` [B,n_loc,3]: the anchors of these rows when
        the caller has evaluated ``mlp_pos`` already (the autograd tier's differentiable head) - the prologue then skips it."""
        lib = _lib.load()
        device = lat_rows.device
        scope = getattr(self, "_anchor_scope", None)
        if scope is not None and anchors is not None and not inference and isinstance(bounds, str):
            hit = scope.get(self._state_key(lat_rows, anchors))
            if hit is not None:                 # run ahead on another stream (prefetch_state): this stream waits for it
                torch.cuda.current_stream(device).wait_event(hit[1])
                return hit[0]
        if getattr(self, "_needs_validation", False):
            from .numerics import validate_numerics
            self._needs_validation = False
            validate_numerics(self, lat_rows[:1].detach(), n=1 << 14)
        knobs = None
        if inference:
            # numerics = "auto": calibrate for these weights if needed, verify the knobs on this latent (n_points given);
            # "fixed": the guard of a pinned approximate mode.  The decision travels with the state: state.nphm_knobs
            knobs = self.kernel_knobs(device, lat_rows, n_points)
        if isinstance(bounds, str):
            # the bounds only matter to the inference kernels; the autograd / training tiers build their member lists with
            # the plain rule
            c = self._calibration
            bounds = c[1].get("bounds") if (inference and self.numerics == "auto" and c is not None
                                            and c[0] == self._weights_key(device)) else None
        packed = self._packed(device)
        B = lat_rows.shape[0]
        lat_rows = lat_rows.contiguous().float()
        state = torch.empty(lib.nphm_identity_latent_state_bytes(B), dtype=torch.uint8, device=device)
        ws, bs = self._lin_params()
        stream = torch.cuda.current_stream(device).cuda_stream
        if anchors is not None and bounds is None and tuple(anchors.shape) == (B, self.num_kps, 3):
            given = anchors.detach().contiguous().float()
            _lib.check(lib.nphm_identity_prepare_latent_anchors(_lib.ptr_array5(ws), _lib.ptr_array5(bs), lat_rows.data_ptr(),
                                                                given.data_ptr(), B, state.data_ptr(), stream),
                       "nphm_identity_prepare_latent_anchors")
            if knobs is not None:
                state.nphm_knobs = knobs      # (every inference launch site unpacks it)
            return packed, state, given
        anchors = torch.empty(B, self.num_kps, 3, dtype=torch.float32, device=device)
        mean = self.anchors.reshape(self.num_kps, 3).to(device=device, dtype=torch.float32).contiguous()
        pw = [self.mlp_pos[i].weight for i in (0, 2, 4)]
        pb = [self.mlp_pos[i].bias for i in (0, 2, 4)]
        for t in pw + pb:
            if t.device != device or t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.NphmAmdError("mlp_pos parameters must be contiguous fp32 on the query device")
        stream = torch.cuda.current_stream(device).cuda_stream
        _lib.check(lib.nphm_identity_prepare_latent(
            packed.data_ptr(), _lib.ptr_array5(ws), _lib.ptr_array5(bs), _lib.ptr_array3(pw),
            _lib.ptr_array3(pb), self.pos_mlp_dim, mean.data_ptr(), lat_rows.data_ptr(), B,
            state.data_ptr(), anchors.data_ptr(), stream), "nphm_identity_prepare_latent")
        if bounds is not None:
            _lib.check(lib.nphm_identity_set_member_bounds(state.data_ptr(), B, bounds.data_ptr(), stream),
                       "nphm_identity_set_member_bounds")
        if knobs is not None:
            state.nphm_knobs = knobs          # (prune_tol, precision code) the inference kernels run this state with
        return packed, state, anchors

    @staticmethod
    def _state_key(lat_rows, anchors):
        return ("state", lat_rows.data_ptr(), lat_rows._version, tuple(lat_rows.shape), anchors.data_ptr(), anchors._version, tuple(anchors.shape))

    def prefetch_state(self, lat_rep, anchors, stream):
        """Inside an ``anchor_scope``: the field's prologue for (row 0 of ``lat_rep`` [1,1,L], ``anchors`` [1,K,3]) launched NOW
        on ``stream``; the next ``prepare_latent`` call of the autograd tier with these tensors takes the result after making
        ITS stream wait.  The fitting step knows code and anchors four launches before it needs the field: the prologue then
        runs beside the correspondence search instead of in front of the member lists.  No-op outside a scope / on the CPU."""
        scope = getattr(self, "_anchor_scope", None)
        if scope is None or not lat_rep.is_cuda or anchors is None or lat_rep.shape[0] != 1:
            return
        rows = _row0(lat_rep).detach()
        anchors = anchors.detach()
        cur = torch.cuda.current_stream(rows.device)
        stream.wait_stream(cur)
        with torch.cuda.stream(stream):
            res = self.prepare_latent(rows, anchors=anchors)
            done = torch.cuda.Event()
            done.record(stream)
        if not torch.cuda.is_current_stream_capturing():
            for t in res:
                if isinstance(t, torch.Tensor):
                    t.record_stream(cur)                 # (allocated under `stream`, read by the caller's)
        scope[self._state_key(rows, anchors)] = (res, done, rows, anchors)

    def _weights_key(self, device):
        ws, bs = self._lin_params()
        return tuple((t.data_ptr(), t._version, str(t.device)) for t in ws + bs) + (str(device),)

    # ---- knobs: plain attributes to the caller; assigning one pins the numerics ----------------------------
    def _pin(self, name, value):
        object.__setattr__(self, name, value)
        object.__setattr__(self, "numerics", "fixed")

    prune_tol = property(lambda self: self._prune_tol, lambda self, v: self._pin("_prune_tol", float(v)))
    precision = property(lambda self: self._precision, lambda self, v: self._pin("_precision", v))
    light_tol = property(lambda self: self._light_tol, lambda self, v: self._pin("_light_tol", v))
    mid_tol = property(lambda self: self._mid_tol, lambda self, v: self._pin("_mid_tol", v))
    refine_band = property(lambda self: self._refine_band, lambda self, v: self._pin("_refine_band", v))

    _MODES = {"f32": _lib.NPHM_PREC_F32, "bf16x3": _lib.NPHM_PREC_BF16X3, "bf16x3a": _lib.NPHM_PREC_BF16X3_ADAPTIVE,
              "bf16x3a2": _lib.NPHM_PREC_BF16X3_ADAPTIVE2, "f16x3": _lib.NPHM_PREC_F16X3,
              "f16x3a2": _lib.NPHM_PREC_F16X3_ADAPTIVE2}
    _MODE_NAMES = {v: k for k, v in _MODES.items()}

    @staticmethod
    def _tier_code(tol):
        """half-octave code of a tier threshold (include/nphm_amd.h: NPHM_PREC_WITH_TIERS): the largest representable
        threshold <= tol; None -> 0 (the mode's default), tol <= 0 -> 255 (tier off)"""
        if tol is None:
            return 0
        if tol <= 0:
            return 255
        import math
        return int(min(254, max(1, math.ceil(2.0 * (1.0 - math.log2(tol)) - 1e-9))))

    @staticmethod
    def precision_code(precision, light_tol=None, mid_tol=None, refine_band=None):
        """``refine_band``: sign-safe refinement (eval_kernel.hip) - tiles with a value within the band of zero are
        re-evaluated with every member on the three-pass product at 1/32 of the pruning budget; coded in bits 24..30 as the
        smallest half-octave step >= the band"""
        T = FastEnsembleDeepSDFMirrored
        rc = 0
        if refine_band is not None and refine_band > 0:
            import math
            rc = int(min(127, max(1, math.floor(2.0 * (1.0 - math.log2(refine_band)) + 1e-9))))
        return T._MODES[precision] | (T._tier_code(light_tol) << 8) | (T._tier_code(mid_tol) << 16) | (rc << 24)

    def _precision_code(self):
        return self.precision_code(self._precision, self._light_tol, self._mid_tol, self._refine_band)

    # evaluations below this many points never use the fast tiers of numerics = "auto" (they run every member on the
    # three-pass split-f16 product: fp32-equivalent, and cheap at that size); larger ones use the calibrated knobs after they
    # have been VERIFIED on a sample of that call's latent
    AUTO_MIN_POINTS = 1 << 16
    _EXACT_KNOBS = (-1.0, "f16x3")
    CHURN_USES = 3          # large evaluations a calibration must have served for the next weight version to be calibrated at once
    TRAIN_F16_MIN_POINTS = 4000     # train_operands = "auto": batches of at least this many points store binary16 operands
    TRAIN_SAT_CHECK_EVERY = 25      # ... and every this many binary16 steps the kernel's clamp counter is read (one synchronisation)

    def _pinned_is_approximate(self):
        return self._prune_tol >= 0 or self._precision in ("bf16x3a", "bf16x3a2", "f16x3a2")

    def kernel_knobs(self, device=None, lat_rows=None, n_points=None):
        """(prune_tol, precision code with its tier thresholds) of the inference kernels for this call.

        numerics = "fixed": the pinned values - for an approximate setting (pruning budget and / or precision tiers) after a
        one-off guard per weight version (``numerics.clamp_pinned``, ``pinned_guard``: measured on the first call's latent,
        tightened with a warning when it deviates by more than the guard; no shipped mode may run outside the 1e-4 bar).

        numerics = "auto" on a ROCm device: the setting calibrated for the current weights (``numerics.calibrate_numerics``,
        cached per weight version, ``calibration`` holds the report).  ``n_points`` (points of this evaluation) and
        ``lat_rows`` [B, lat_dim] make it per-call: below ``AUTO_MIN_POINTS`` points every member runs the three-pass
        product (no knob to trust); a larger evaluation of a latent that has not been seen yet first measures the
        calibrated knobs on a sample of THAT latent (``numerics.sample_error``: 512 lattice tiles against the three-pass
        kernel with all members, ~0.3 ms, one synchronisation) and re-calibrates on the union of the latents when the
        sample exceeds 1.5 x the target - later latents do not inherit the first one's knobs unverified.  Under stream
        capture nothing can be measured: cached knobs of a verified latent are used, anything else runs the exact setting."""
        if device is None or torch.device(device).type != "cuda":
            return float(self._prune_tol), self._precision_code()
        capturing = torch.cuda.is_current_stream_capturing()
        key = self._weights_key(device)
        if self.numerics != "auto":
            guard = self.pinned_guard
            if not guard or not self._pinned_is_approximate() or lat_rows is None:
                return float(self._prune_tol), self._precision_code()
            pin = (self._precision, self._light_tol, self._mid_tol, self._prune_tol, self._refine_band, float(guard))
            c = self._pinned_check
            fresh = c is None or c[0] != (key, pin)
            large = n_points is None or n_points >= self.AUTO_MIN_POINTS
            if not capturing and (fresh or large):
                # measured once per weight version on 8 192 lattice tiles of the first latent, then on 1 024 tiles of every
                # NEW latent of a large evaluation, starting from what the earlier latents left (tightening is monotone)
                lat = lat_rows.detach().reshape(-1, self.lat_dim)[:1]
                seen = {} if fresh else c[2]
                dig = self._latent_digest(lat)
                if dig not in seen:
                    from .numerics import clamp_pinned
                    start = (dict(precision=self._precision, light_tol=self._light_tol, mid_tol=self._mid_tol, prune_tol=self._prune_tol)
                             if fresh else {k: c[1][k] for k in ("precision", "light_tol", "mid_tol", "prune_tol")})
                    rep = clamp_pinned(self, lat.float(), refine_band=self._refine_band, guard=float(guard),
                                       n_tiles=8192 if fresh else 1024, **start)
                    if not fresh:                      # keep what was asked originally in the report
                        rep["asked"], rep["asked_error"] = c[1]["asked"], c[1]["asked_error"]
                        rep["clamped"] = rep["clamped"] or c[1]["clamped"]
                    seen[dig] = rep["error"]
                    c = ((key, pin), rep, seen)
                    object.__setattr__(self, "_pinned_check", c)
            if c is None or c[0] != (key, pin) or not c[1]["clamped"]:
                return float(self._prune_tol), self._precision_code()
            r = c[1]
            return float(r["prune_tol"]), self.precision_code(r["precision"], r["light_tol"], r["mid_tol"], self._refine_band)
        exact = (self._EXACT_KNOBS[0], self.precision_code(self._EXACT_KNOBS[1]))
        if n_points is not None and n_points < self.AUTO_MIN_POINTS:
            return exact
        have = self._calibration is not None and self._calibration[0] == key
        hist = self.__dict__.setdefault("_auto_hist", {"uses": 0, "pending": None})
        if have:
            hist["uses"] += 1
        if not have:
            if capturing:
                return exact
            # Weight churn (a training loop that extracts one validation mesh per epoch, training.py:312-323): a calibration
            # costs ~0.17 s and pays for itself after ~10 volumes.  If the PREVIOUS calibration served fewer than
            # CHURN_USES large evaluations before the weights changed, the new weights run the exact three-pass setting
            # until a second large evaluation sees them unchanged; a first calibration in a process, or weights that
            # replace a well-used calibration, are calibrated at once.
            if self._calibration is not None and hist["uses"] < self.CHURN_USES and hist["pending"] != key:
                hist["pending"] = key
                return exact
            hist["uses"], hist["pending"] = 1, None
            from .numerics import calibrate_numerics
            lat = None if lat_rows is None else lat_rows.detach().reshape(-1, self.lat_dim)[:2]
            object.__setattr__(self, "_calibration", (key, dict(calibrate_numerics(self, lat, device=device), serial=next(_CAL_SERIAL))))
            object.__setattr__(self, "_verified_latents", {})
            if lat is not None:
                self._verified_latents[self._latent_digest(lat)] = self._calibration[1]["error"]
        elif lat_rows is not None and n_points is not None:
            lat = lat_rows.detach().reshape(-1, self.lat_dim)[:8]          # every row of the usual batches (a digest + <= 8 samples)
            if capturing:
                return exact                  # (a digest needs a device -> host copy; captured evaluations stay exact)
            dig = self._latent_digest(lat)
            lat = lat.float()
            if dig not in self._verified_latents:
                self._verify_latent(lat, dig, device)
        c = self._calibration[1]
        return float(c["prune_tol"]), self.precision_code(c["precision"], c["light_tol"], c["mid_tol"], c.get("refine_band"))

    def inference_numerics(self, device, lat_rows, n_points):
        """((prune_tol, precision code), member bounds [40,4] or None) this rank would run an inference call with, and a
        serial number of the calibration behind them (0: none; it changes whenever the calibration is replaced) - what
        ``reconstruction.shared_identity_numerics`` sends from one rank to the others of a sharded evaluation."""
        knobs = self.kernel_knobs(device, lat_rows, n_points)
        c = self._calibration
        have = self.numerics == "auto" and c is not None and c[0] == self._weights_key(device)
        return knobs, (c[1].get("bounds") if have else None), (int(c[1].get("serial", 0)) if have else 0)

    def _latent_digest(self, lat):
        """Content digest of the latent rows (a device -> host copy of a few KiB = one synchronisation), memoised per
        (address, version counter, size) WHILE the tensor is held alive by the memo itself: a tensor we keep a reference to
        cannot have its memory recycled for another latent, and an unchanged version counter means unchanged content -
        the repeated evaluations of one latent (a benchmark loop, the chunks of one extraction) synchronise once."""
        import hashlib
        memo = self.__dict__.setdefault("_digest_memo", {})
        k = (lat.data_ptr(), lat._version, lat.numel())
        hit = memo.get(k)
        if hit is not None:
            return hit[1]
        dig = hashlib.sha1(lat.detach().contiguous().cpu().numpy().tobytes()).hexdigest()
        if len(memo) >= 8:
            memo.pop(next(iter(memo)))
        memo[k] = (lat, dig)
        return dig

    def _verify_latent(self, lat, dig, device):
        """The calibrated knobs measured on a sample of every row of ``lat`` [R <= 8, lat_dim]; above 1.5 x target: re-calibrate
        on the union of the calibration latents (at most 4 are kept) and the worst rows.  A replaced calibration voids what
        was verified under the old knobs."""
        from .numerics import calibrate_numerics, sample_error
        c = self._calibration[1]
        errs = [sample_error(self, lat[r:r + 1], precision=c["precision"], light_tol=c["light_tol"], mid_tol=c["mid_tol"],
                             prune_tol=c["prune_tol"], refine_band=c.get("refine_band"), bounds=c.get("bounds"))
                for r in range(lat.shape[0])]
        err = max(errs)
        if err > 1.5 * c["target"]:
            worst = lat[torch.tensor(errs).argsort(descending=True)[:2].to(lat.device)]
            union = torch.cat([c["latents"].to(lat), worst])[-4:]
            new = calibrate_numerics(self, union, device=device)
            new["recalibrated_for"] = {"digest": dig, "error_before": err}
            new["serial"] = next(_CAL_SERIAL)
            object.__setattr__(self, "_calibration", (self._calibration[0], new))
            self._verified_latents.clear()               # measured with the old knobs
            err = new["error"]
        if len(self._verified_latents) >= 256:
            self._verified_latents.pop(next(iter(self._verified_latents)))
        self._verified_latents[dig] = err

    @property
    def calibration(self):
        """report of the last calibration (numerics = "auto"), or None"""
        return None if self._calibration is None else self._calibration[1]

    def _forward_hip(self, xyz, lat_rows):
        lib = _lib.load()
        B, N, _ = xyz.shape
        packed, state, anchors = self.prepare_latent(lat_rows, inference=True, n_points=B * N)
        xyz = xyz.contiguous().float()
        out = torch.empty(B, N, 1, dtype=torch.float32, device=xyz.device)
        stream = torch.cuda.current_stream(xyz.device).cuda_stream
        _lib.check(lib.nphm_identity_eval_points(
            packed.data_ptr(), state.data_ptr(), xyz.data_ptr(), B, N, 0 if self.training else N,
            *state.nphm_knobs, out.data_ptr(), None, stream),
            "nphm_identity_eval_points")
        return out, anchors

    def _forward_hip_autograd(self, xyz, lat_rows):
        """Differentiable (first order) w.r.t. xyz and the latent rows; values from the fused kernel."""
        B = xyz.shape[0]
        g, A = self.lat_dim_glob, self.num_kps + 1
        # parameters enter as constants here when they are (treated as) frozen: no .grad is produced for them
        if self.assume_frozen_parameters or not any(p.requires_grad for p in self.mlp_pos.parameters()):
            anchors = self._anchors_of_rows(lat_rows)          # fused head, shared inside an anchor_scope
        else:
            anchors = self.mlp_pos(lat_rows[:, :g]).view(B, self.num_kps, 3)
            anchors = anchors + self.anchors.reshape(1, self.num_kps, 3).to(anchors)
        sdf = _IdentityFieldFn.apply(self, xyz, lat_rows, anchors)
        return sdf, anchors

    _TRAIN_MAG_REF = 0.04     # RMS of the kept member values at the seeded initialisation (0.044)

    def _autograd_tol(self, packed, state, xyz, stream):
        """Pruning budget of the first-order autograd tier (latent fitting).  A pinned ``prune_tol`` (numerics = "fixed") is
        taken as it is; with numerics = "auto" it is scaled like the training tier's (``_train_tol``) to the size of the
        member values, measured ONCE per weight version on the first call's own points (the members the plain rule keeps, one
        extra forward launch and one read-back; fitting loops freeze the weights): on a trained-like checkpoint the plain 1e-7
        leaves 1.7e-5 of the values and 3.7e-4 of the spatial gradients, the scaled budget (1.5e-8) 1e-6 / 3e-5.  Inside a
        graph capture without a measurement: a tenth of the budget."""
        tol = float(self._prune_tol)
        if self.numerics != "auto" or tol <= 0:
            return tol
        key = self._weights_key(xyz.device)
        c = getattr(self, "_fit_mag", None)
        if c is None or c[0] != key:
            if torch.cuda.is_current_stream_capturing():
                return tol / 10.0
            lib = _lib.load()
            A = self.num_kps + 1
            x = xyz[:1, : min(xyz.shape[1], 4096)].contiguous()
            what, tiles, n_used, plist = _member_point_lists_device(state, x, tol, A, stream)
            fmem = torch.empty(1, x.shape[1], A, dtype=torch.float32, device=x.device)
            _lib.check(lib.nphm_identity_member_forward(
                packed.data_ptr(), self._packed_bwd(x.device).data_ptr(), state.data_ptr(), x.data_ptr(), x.shape[1],
                tiles.data_ptr(), tiles.shape[0], n_used.data_ptr(), plist.data_ptr(), fmem.data_ptr(), stream),
                "nphm_identity_member_forward")
            kept = what > 0
            rms = float(torch.sqrt(torch.where(kept, fmem, torch.zeros_like(fmem)).square().sum() / kept.sum().clamp(min=1)))
            c = (key, rms)
            object.__setattr__(self, "_fit_mag", c)
        return tol / min(max(1.0, c[1] / self._TRAIN_MAG_REF), 100.0)

    def _train_tol(self):
        """Pruning budget of the training tier.  ``train_prune_tol`` if the caller pinned one; otherwise ``prune_tol`` scaled
        to the size of the member values: the rule drops members by their BLEND WEIGHT, the error it leaves is weight x
        member value, and training takes the member values from RMS 0.04 (initialisation: 1e-7 leaves 5e-6 of a gradient)
        to ~0.3 with far-field values of several units (a trained-like checkpoint: 1e-7 leaves 6e-3 of some bias gradients,
        1e-8 leaves 3e-5).  The size is the RMS of the kept member values of the previous step (read here - the step
        synchronises once anyway to size its tile lists); the first step, which has none, takes a tenth of the budget.
        Consequence: the kept set follows the weights from step to step - pin ``train_prune_tol`` for bitwise-repeatable
        calls."""
        if self.train_prune_tol is not None:
            return float(self.train_prune_tol)
        tol = float(self.prune_tol)
        if tol <= 0:
            return tol
        mag = getattr(self, "_train_mag", None)
        if mag is None:
            return tol / 10.0
        rms = float(mag[0]) / max(float(mag[1]), 1.0) ** 0.5
        return tol / min(max(1.0, rms / self._TRAIN_MAG_REF), 100.0)

    def _train_members(self, xyz, lat_rows):
        """(anchors, member values S [B,N,40], member gradients G = dS/dxyz [B,N,40,3]) on the training kernels."""
        B = xyz.shape[0]
        g, A = self.lat_dim_glob, self.num_kps + 1
        e = self.ensembled_deep_sdf
        anchors = self.mlp_pos(lat_rows[:, :g]).view(B, self.num_kps, 3)
        anchors = anchors + self.anchors.reshape(1, self.num_kps, 3).to(anchors)
        cond = torch.cat([lat_rows[:, None, :g].expand(B, A, g), lat_rows[:, g:].reshape(B, A, self.lat_dim_loc)], dim=-1)
        d_in, n1 = self.input_dim, e.lin1.out_features
        W0m, W2m = e.lin0.member_weight(), e.lin2.member_weight()
        b0f = torch.einsum("baf,aof->bao", cond, W0m[:, :, d_in:]) + e.lin0.member_bias()[None]
        b2f = torch.einsum("baf,aof->bao", cond, W2m[:, :, n1 + d_in:]) / _SQRT2 + e.lin2.member_bias()[None]
        S, G = _MemberFieldFn.apply(self, xyz, anchors, lat_rows, b0f, b2f, e.lin0.weight, e.lin1.weight, e.lin2.weight,
                                    e.lin3.weight, e.lin4.weight, e.lin1.bias, e.lin3.bias, e.lin4.bias)
        return anchors, S, G

    def _forward_hip_train(self, xyz, lat_rows):
        """Training tier: twice differentiable w.r.t. xyz, differentiable w.r.t. the latent rows and every
        parameter.  Member MLPs (EnsembledDeepSDF.py:101-126) and their first / second-order backward on the
        HIP kernels; anchors, the latent columns of lin0 / the skip layer and the Gaussian blend
        (EnsembledDeepSDF.py:129-150) stay ordinary autograd (a few [B,N,40] elementwise ops)."""
        anchors, S, G = self._train_members(xyz, lat_rows)
        f = _AttachGradientFn.apply(xyz, anchors, S, G)
        pred = sample_point_feature(xyz[..., :3], anchors, f.unsqueeze(-1), background=True, var=0.1 ** 2)
        return pred, anchors

    def _train_tier_serves(self, xyz, lat_rep, eval_ok=False):
        """The HIP training tier's conditions: train mode (or a caller that states the eval-mode overwrite itself),
        trainable parameters, ROCm fp32 tensors, the NPHM architecture, one latent per batch row."""
        if not ((self.training or eval_ok) and self.backend != "composite" and self.train_backend == "hip" and xyz.is_cuda
                and xyz.dtype == torch.float32 and torch.is_grad_enabled() and self.hip_supported()):
            return False
        if self.assume_frozen_parameters or not any(p.requires_grad for p in self.parameters()):
            return False
        N = xyz.shape[1]
        return lat_rep.shape[1] == 1 or (lat_rep.shape[1] == N and bool((lat_rep == lat_rep[:, :1]).all()))

    def value_and_gradient(self, xyz: torch.Tensor, lat_rep: torch.Tensor, last_points=None):
        """(sdf [B,N,1], d sdf / d xyz [B,N,3], anchors) in ONE differentiable evaluation - what ``compute_loss`` obtains
        from ``decoder(...)`` followed by ``gradient(pred, x)`` (loss_functions.py:36-49) - or None when the HIP
        training tier does not serve the call (the caller then does exactly that).  The spatial gradient is an output
        of the fused blend kernel, so no graph-recording backward pass is needed: ``loss.backward()`` on terms of both
        outputs differentiates through the member kernels' second-order backward.

        Eval mode (the validation step, training.py:250-268): the reference overwrites every member's value of the LAST
        point of each decoder call with 1 (EnsembledDeepSDF.py:260-261).  A caller that evaluates several of the
        reference's calls as one batch names those points in ``last_points`` (indices along N); without it an
        eval-mode module is not served here (None)."""
        if xyz.dim() < 3:
            xyz = xyz.unsqueeze(0)
        if not self._train_tier_serves(xyz, lat_rep, eval_ok=last_points is not None):
            return None
        anchors, S, G = self._train_members(xyz, lat_rep[:, 0, :])
        if not self.training:
            # constants at the overwritten points: value 1 for all 40 members (the blend then returns
            # sum(w) / (sum(w) + 1e-6)), no member gradient - written as a blend, autograd sees no in-place update
            keep = torch.ones(xyz.shape[1], dtype=S.dtype, device=S.device)
            keep[torch.as_tensor(last_points, device=S.device, dtype=torch.long)] = 0.0
            S = S * keep[None, :, None] + (1.0 - keep)[None, :, None]
            G = G * keep[None, :, None, None]
        pred, grad = _BlendFn.apply(xyz, anchors, S, G)
        return pred.unsqueeze(-1), grad, anchors

    def predict_anchors(self, lat_rep: torch.Tensor) -> torch.Tensor:
        """Anchors of the identity codes ``lat_rep`` [B, L, lat_dim] (row 0 of every batch entry): the
        second return value of ``forward`` (EnsembledDeepSDF.py:228-229) without evaluating any SDF -
        differentiable w.r.t. the global code.  The fitting loops call it instead of the reference's
        one-point ``decoder(zeros, lat, None)`` whose SDF value they discard (fitting.py:58, :208)."""
        return self._anchors_of_rows(_row0(lat_rep))

    def _anchors_of_rows(self, lat_rows):
        """anchors [B,39,3] of latent rows [B, lat_dim] (differentiable w.r.t. the global code).  Inside ``anchor_scope()``
        the rows of one code tensor - also as an expanded view of it - are evaluated once: a fitting step needs the anchors
        of its identity code for the deformation field's conditioning AND inside the identity field."""
        B = lat_rows.shape[0]
        scope = getattr(self, "_anchor_scope", None)
        key = None
        if scope is not None:
            key = (lat_rows.data_ptr(), lat_rows._version, lat_rows.shape[1], str(lat_rows.device))
            hit = scope.get(key)
            if hit is not None and (hit[0].shape[0] == B or (hit[0].shape[0] == 1 and (B == 1 or lat_rows.stride(0) == 0))):
                a2 = hit[2]                                  # the alias for the later uses (their gradients meet in the head's backward launch)
                return a2 if a2.shape[0] == B else a2.expand(B, -1, -1)
            if B > 1 and lat_rows.stride(0) == 0:
                lat_rows = lat_rows[:1]                      # identical rows (an expanded code): evaluate one
        frozen = self.assume_frozen_parameters
        # the head reads the global part of the whole row and adds the mean anchors itself (no slice / add launches)
        if key is not None:
            a, a2 = frozen_head(self.mlp_pos, lat_rows, frozen, add=self.anchors, twice=True)
            a, a2 = a.view(lat_rows.shape[0], self.num_kps, 3), a2.view(lat_rows.shape[0], self.num_kps, 3)
            scope[key] = (a, lat_rows, a2)
        else:
            a = frozen_head(self.mlp_pos, lat_rows, frozen, add=self.anchors).view(lat_rows.shape[0], self.num_kps, 3)
        return a if a.shape[0] == B else a.expand(B, -1, -1)

    from contextlib import contextmanager as _cm

    @_cm
    def anchor_scope(self):
        """see ``_anchors_of_rows``; the scope holds the tensors, so an address cannot be recycled under it"""
        self._anchor_scope = {}
        try:
            yield self
        finally:
            self._anchor_scope = None

    def _forward_composite(self, xyz, lat_rep):
        B, N, _ = xyz.shape
        A = self.num_kps + 1
        Lr = lat_rep.shape[1]
        g = self.lat_dim_glob
        anchors = self.mlp_pos(lat_rep[:, 0, :g]).view(B, self.num_kps, 3)
        anchors = anchors + self.anchors.reshape(1, self.num_kps, 3).to(anchors)
        origin = torch.cat([anchors, torch.zeros_like(anchors[:, :1])], dim=1)            # [B,A,3]
        local = xyz[:, None, :, :] - origin[:, :, None, :]                               # [B,A,N,3]
        flip = torch.ones(A, 3, dtype=xyz.dtype, device=xyz.device)
        flip[1:2 * self.num_symm_pairs:2, 0] = -1.0
        local = (local * flip[None, :, None, :]).permute(1, 0, 2, 3)                      # [A,B,N,3]
        z_loc = lat_rep[..., g:].reshape(B, Lr, A, self.lat_dim_loc)
        cond = torch.cat([lat_rep[:, :, None, :g].expand(B, Lr, A, g), z_loc], dim=-1)
        cond = cond.permute(2, 0, 1, 3)                                                   # [A,B,Lr,F]
        f = self.ensembled_deep_sdf.evaluate(local, cond)                                 # [A,B,N,1]
        if not self.training:
            # reference quirk (EnsembledDeepSDF.py:260-261): in eval mode every member's value of
            # the LAST point of each batch row is overwritten with 1
            # (channel 0 only, like `sdf_pred[:, :, -1, 0] = 1`; written as a blend so that autograd sees no
            # in-place update)
            keep = torch.ones(N, f.shape[-1], dtype=f.dtype, device=f.device)
            keep[-1, 0] = 0.0
            f = f * keep[None, None] + (1.0 - keep)[None, None]
        pred = sample_point_feature(xyz[..., :3], anchors, f.permute(1, 2, 0, 3), background=True, var=0.1 ** 2)
        return pred, anchors

    # ------------------------------------------------------------------------------------------
    def forward(self, xyz: torch.Tensor, lat_rep: torch.Tensor, anchors_gt: Optional[torch.Tensor] = None):
        """xyz [B,N,3] (or [N,3]); lat_rep [B,N or 1,lat_dim] laid out as
        [z_glob | z_1, z*_1, ..., z_16, z*_16 | z_mid.. | z_bg]; ``anchors_gt`` is ignored (as in the
        reference).  Returns (sdf [B,N,1], anchors [B,n_loc,3])."""
        if xyz.dim() < 3:
            xyz = xyz.unsqueeze(0)
        B, N, _ = xyz.shape
        assert self.lat_dim == lat_rep.shape[-1], "lat dim {}, lat_rep {}".format(self.lat_dim, lat_rep.shape)

        params_train = (not self.assume_frozen_parameters) and any(p.requires_grad for p in self.parameters())
        needs_graph = torch.is_grad_enabled() and (xyz.requires_grad or lat_rep.requires_grad or params_train)
        if self.backend == "composite":
            return self._forward_composite(xyz, lat_rep)
        if not xyz.is_cuda:
            raise _lib.NphmAmdError(
                "FastEnsembleDeepSDFMirrored: tensors are on the CPU; the HIP path needs a ROCm device "
                "(set module.backend = 'composite' explicitly for the PyTorch formulation)")
        if not self.hip_supported() or xyz.dtype != torch.float32:
            return self._forward_composite(xyz, lat_rep)
        if lat_rep.shape[1] != 1:
            # get_logits passes encoding.repeat(1, N, 1): constant along the point axis
            if lat_rep.shape[1] != N or not bool((lat_rep == lat_rep[:, :1]).all()):
                return self._forward_composite(xyz, lat_rep)
        if needs_graph:
            # first-order HIP autograd tier: latent fitting (no parameter requires grad, train mode, no
            # chunk overwrite to differentiate around); everything else builds the composite graph
            if self.training and not params_train:
                return self._forward_hip_autograd(xyz, _row0(lat_rep))
            if self.training and self.train_backend == "hip":
                return self._forward_hip_train(xyz, lat_rep[:, 0, :])
            return self._forward_composite(xyz, lat_rep)
        return self._forward_hip(xyz, lat_rep[:, 0, :])
